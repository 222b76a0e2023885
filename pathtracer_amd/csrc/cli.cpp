// pt_render -- headless stand-in for the reference's windowed application
// (main.cpp:114-193): same default scene path, light, camera and resolution,
// but the image goes to a file instead of a window.
//
//   pt_render [--scene assets/TEST.obj] [--mtl-root assets/] [--width 1280] [--height 720]
//             [--spp 64] [--depth 10] [--fov 19.5] [--eye x y z] [--focus x y z]
//             [--orbit dx dy] [--pan dx dy] [--zoom s]      camera moves of main.cpp:75-112, applied in order
//             [--light r g b  px py pz  rx ry rz  sx sy sz]  repeatable; replaces the default light
//             [--no-light]
//             [--resume state.bin] [--save-state state.bin] [--checkpoint-every n]
//             [--preview-every n]   rewrite --out after every n samples per pixel (a progressive preview)
//             [--scan] [--bvh host|device] [--out image.ppm] [--pfm radiance.pfm]
//             [--gpus n]            sample-partition the render over n GPUs of this machine: GPU g renders
//                                   iterations g, g + n, ...; one RCCL reduce into GPU 0 before an image is
//                                   written (not with --resume / --save-state)
// --spp is the total: after --resume only the missing samples are rendered.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

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
    bool scan = false, no_light = false, device_bvh = false;
    std::string resume, save_state;
    uint32_t checkpoint_every = 0, preview_every = 0, gpus = 1;
    bool gpus_given = false;  // --gpus 1 takes the multi-GPU code path with a one-rank communicator
    std::vector<pt::Scene::Light> lights;
    struct Move {
        char kind;
        float a, b;
    };
    std::vector<Move> moves;
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
        else if (!std::strcmp(argv[i], "--bvh")) device_bvh = !std::strcmp(next(), "device"), ++i;
        else if (!std::strcmp(argv[i], "--no-light")) no_light = true;
        else if (!std::strcmp(argv[i], "--resume")) resume = next(), ++i;
        else if (!std::strcmp(argv[i], "--save-state")) save_state = next(), ++i;
        else if (!std::strcmp(argv[i], "--checkpoint-every")) checkpoint_every = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--preview-every")) preview_every = std::atoi(next()), ++i;
        else if (!std::strcmp(argv[i], "--gpus")) gpus = std::max(1, std::atoi(next())), gpus_given = true, ++i;
        else if (!std::strcmp(argv[i], "--light")) {
            float v[12];
            for (int k = 0; k < 12; ++k) v[k] = static_cast<float>(std::atof(next(k + 1)));
            lights.emplace_back(pt::make3(v[0], v[1], v[2]), pt::make3(v[3], v[4], v[5]), pt::make3(v[6], v[7], v[8]),
                                pt::make3(v[9], v[10], v[11]));
            i += 12;
        } else if (!std::strcmp(argv[i], "--orbit") || !std::strcmp(argv[i], "--pan")) {
            moves.push_back(Move{argv[i][2], static_cast<float>(std::atof(next(1))), static_cast<float>(std::atof(next(2)))});
            i += 2;
        } else if (!std::strcmp(argv[i], "--zoom")) {
            moves.push_back(Move{'z', static_cast<float>(std::atof(next())), 0.f});
            ++i;
        }
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
        if (no_light || !lights.empty()) {
            scene.m_lights = lights;
            scene.m_lightCount[0] = static_cast<uint32_t>(lights.size());
        } else {
            scene.setDefaultLight();
        }
        pt::Camera camera(eye, focus, width, height);
        for (const Move &m : moves) {
            if (m.kind == 'o') camera.orbit(m.a, m.b);
            else if (m.kind == 'p') camera.pan(m.a, m.b);
            else camera.zoom(m.a);
        }
        if (gpus_given) {
            // One process, several GPUs: the loop of main.cpp:183-189 over a MultiRenderer.
            if (!resume.empty() || !save_state.empty()) {
                std::fprintf(stderr, "--resume / --save-state do not go with --gpus\n");
                return 2;
            }
            const int visible = pt_device_count();
            if (visible < static_cast<int>(gpus)) {
                std::fprintf(stderr, "--gpus %u, but %d GPU(s) visible\n", gpus, visible);
                return 1;
            }
            pt::MultiRenderer multi(&scene, &camera, gpus, depth, fov, !scan, device_bvh);
            auto t0 = std::chrono::steady_clock::now();
            uint32_t done = 0;
            while (done < spp) {
                const uint32_t n = preview_every ? std::min(spp - done, preview_every - done % preview_every) : spp - done;
                multi.draw(n);
                done += n;
                if (preview_every && done < spp) {
                    multi.reduce();
                    if (!write_ppm(out, multi.root().tonemapped(), width, height)) std::fprintf(stderr, "cannot write %s\n", out.c_str());
                    std::printf("preview at %u spp\n", done);
                }
            }
            multi.reduce();
            multi.synchronize();
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t paths = 0, segments = 0;
            for (size_t g = 0; g < multi.gpus(); ++g) {
                const pt_stats st = multi.renderer(g).stats();
                paths += st.paths;
                segments += st.segments;
            }
            std::printf("%zu triangles, %ux%u, %u spp on %u GPUs, depth %u: %.3f s, %.1f Mpaths/s, %.2f segments/path\n",
                        scene.m_indices.size(), width, height, done, gpus, depth, secs, static_cast<double>(paths) / secs / 1e6,
                        paths ? static_cast<double>(segments) / static_cast<double>(paths) : 0.0);
            if (!write_ppm(out, multi.root().tonemapped(), width, height)) {
                std::fprintf(stderr, "cannot write %s\n", out.c_str());
                return 1;
            }
            if (!pfm.empty() && !write_pfm(pfm, multi.root().radiance(), width, height)) {
                std::fprintf(stderr, "cannot write %s\n", pfm.c_str());
                return 1;
            }
            return 0;
        }
        pt::Renderer renderer(&scene, &camera, depth, fov, !scan, -1, device_bvh);
        uint32_t done = 0;
        if (!resume.empty()) {
            renderer.loadState(resume);
            done = static_cast<uint32_t>(renderer.stats().iterations);
            std::printf("resumed %s at %u spp\n", resume.c_str(), done);
        }
        const uint32_t resumed_at = done;
        auto t0 = std::chrono::steady_clock::now();
        while (done < spp) {
            uint32_t n = spp - done;
            if (checkpoint_every) n = std::min(n, checkpoint_every - done % checkpoint_every);
            if (preview_every) n = std::min(n, preview_every - done % preview_every);
            renderer.draw(n);
            done += n;
            if (checkpoint_every && done % checkpoint_every == 0 && !save_state.empty()) renderer.saveState(save_state);
            if (preview_every && done % preview_every == 0 && done < spp) {
                if (!write_ppm(out, renderer.tonemapped(), width, height)) std::fprintf(stderr, "cannot write %s\n", out.c_str());
                std::printf("preview at %u spp\n", done);
            }
        }
        renderer.synchronize();
        if (!save_state.empty()) renderer.saveState(save_state);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        pt_stats st = renderer.stats();
        const double new_paths = static_cast<double>(done - resumed_at) * width * height;  // what this run traced
        std::printf("%zu triangles, %u BVH nodes (depth %u, built on the %s in %.1f ms), %ux%u, %u spp, depth %u: %.3f s, "
                    "%.1f Mpaths/s, %.2f segments/path\n",
                    scene.m_indices.size(), st.n_nodes, st.bvh_depth, st.bvh_builder ? "device" : "host", st.bvh_build_ms,
                    width, height, done, depth, secs,
                    new_paths / secs / 1e6, st.paths > 0 ? static_cast<double>(st.segments) / static_cast<double>(st.paths) : 0.0);
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
