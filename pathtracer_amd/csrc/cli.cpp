// pt_render -- headless stand-in for the reference's windowed application
// (main.cpp:114-193): same default scene path, light, camera and resolution,
// but the image goes to a file instead of a window.
//
//   pt_render [--scene assets/TEST.obj] [--mtl-root assets/] [--width 1280] [--height 720]
//             [--spp 64] [--depth 10] [--fov 19.5] [--eye x y z] [--focus x y z] [--scan]
//             [--out image.ppm] [--pfm radiance.pfm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "renderer.h"

namespace {

bool write_ppm(const std::string &path, const std::vector<uint8_t> &rgb, uint32_t w, uint32_t h)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

// PFM stores rows bottom-up, which is the renderer's own row order.
bool write_pfm(const std::string &path, const std::vector<float> &rgba, uint32_t w, uint32_t h)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    bool ok = true;
    for (size_t p = 0; p < static_cast<size_t>(w) * h && ok; ++p) ok = std::fwrite(&rgba[4 * p], 4, 3, f) == 3;
    std::fclose(f);
    return ok;
}

}  // namespace

int main(int argc, char **argv)
{
    std::string scene_path = "assets/TEST.obj", mtl_root = "assets/", out = "image.ppm", pfm;
    uint32_t width = 1280, height = 720, spp = 64, depth = 10;
    float fov = 19.5f;
    pt::Vec3 eye = pt::make3(0.f, 1.5f, 15.f), focus = pt::make3(0.f, -0.25f, 0.f);
    bool scan = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](int k = 1) -> const char * {
            if (i + k >= argc) {
                std::fprintf(stderr, "missing value after %s\n", argv[i]);
                std::exit(2);
            }
            return argv[i + k];
        };
        if (!std::strcmp(argv[i], "--scene")) scene_path = next(), ++i;
        else if (!std::strcmp(argv[i], "--mtl-root")) mtl_root = next(), ++i;
        else if (!std::strcmp(argv[i], "--width")) width = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--height")) height = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--spp")) spp = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--depth")) depth = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--fov")) fov = static_cast<float>(std::atof(next())), ++i;
        else if (!std::strcmp(argv[i], "--out")) out = next(), ++i;
        else if (!std::strcmp(argv[i], "--pfm")) pfm = next(), ++i;
        else if (!std::strcmp(argv[i], "--scan")) scan = true;
        else if (!std::strcmp(argv[i], "--eye")) {
            eye = pt::make3(std::atof(next(1)), std::atof(next(2)), std::atof(next(3)));
            i += 3;
        } else if (!std::strcmp(argv[i], "--focus")) {
            focus = pt::make3(std::atof(next(1)), std::atof(next(2)), std::atof(next(3)));
            i += 3;
        } else {
            std::fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    std::printf("PathTracer\n");
    try {
        pt::Scene scene(scene_path.c_str(), mtl_root.c_str());
        if (!scene.m_error.empty()) std::fprintf(stderr, "%s\n", scene.m_error.c_str());
        if (scene.m_indices.empty()) {
            std::fprintf(stderr, "no triangles loaded from %s\n", scene_path.c_str());
            return 1;
        }
        scene.setDefaultLight();
        pt::Camera camera(eye, focus, width, height);
        pt::Renderer renderer(&scene, &camera, depth, fov, !scan);
        auto t0 = std::chrono::steady_clock::now();
        renderer.draw(spp);
        renderer.synchronize();
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        pt_stats st = renderer.stats();
        std::printf("%zu triangles, %u BVH nodes (depth %u), %ux%u, %u spp, depth %u: %.3f s, %.1f Mpaths/s, %.2f segments/path\n",
                    scene.m_indices.size(), st.n_nodes, st.bvh_depth, width, height, spp, depth, secs,
                    static_cast<double>(st.paths) / secs / 1e6, static_cast<double>(st.segments) / static_cast<double>(st.paths));
        if (!write_ppm(out, renderer.tonemapped(), width, height)) {
            std::fprintf(stderr, "cannot write %s\n", out.c_str());
            return 1;
        }
        if (!pfm.empty() && !write_pfm(pfm, renderer.radiance(), width, height)) {
            std::fprintf(stderr, "cannot write %s\n", pfm.c_str());
            return 1;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
