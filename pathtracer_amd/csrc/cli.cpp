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
//             [--scan] [--bvh host|device] [--out image.ppm | image.png] [--pfm radiance.pfm]
//                                   (--out: a name ending in .png is written as PNG, the format of the reference's
//                                   published output/*.png; anything else as binary PPM)
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

// PNG, 8-bit RGB, rows top-down as the tone-mapped image has them.  The pixel data go into the zlib stream as
// stored (uncompressed) deflate blocks: no compression library is needed, and a file of this kind is read by
// every PNG decoder.  Chunks carry a CRC-32 (polynomial 0xEDB88320), the zlib stream ends with the Adler-32 of the
// raw data.
uint32_t crc32_update(uint32_t crc, const uint8_t *data, size_t n)
{
    static uint32_t table[256];
    static bool made = false;
    if (!made) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        made = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    for (int s = 24; s >= 0; s -= 8) v.push_back(static_cast<uint8_t>(x >> s));
}
bool write_png_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put_be32(head, static_cast<uint32_t>(data.size()));
    head.insert(head.end(), type, type + 4);
    uint32_t crc = crc32_update(0xFFFFFFFFu, head.data() + 4, 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}
bool write_png(const std::string &path, const std::vector<uint8_t> &rgb, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0 || rgb.size() != static_cast<size_t>(w) * h * 3) return false;
    // scanlines, each led by its filter byte (0 = none)
    std::vector<uint8_t> raw;
    raw.reserve((static_cast<size_t>(w) * 3 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + static_cast<size_t>(y) * w * 3, rgb.begin() + static_cast<size_t>(y + 1) * w * 3);
    }
    uint32_t a = 1, b = 0;  // Adler-32
    for (size_t i = 0; i < raw.size();) {
        const size_t n = std::min<size_t>(5552, raw.size() - i);  // the sums cannot overflow 32 bits within 5552 bytes
        for (size_t k = 0; k < n; ++k) {
            a += raw[i + k];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        i += n;
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);  // deflate, 32 K window
    z.push_back(0x01);  // no preset dictionary, check bits: 0x7801 is a multiple of 31
    for (size_t i = 0; i < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - i);
        z.push_back(i + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back(static_cast<uint8_t>(n & 0xFF));
        z.push_back(static_cast<uint8_t>(n >> 8));
        z.push_back(static_cast<uint8_t>(~n & 0xFF));
        z.push_back(static_cast<uint8_t>((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + i, raw.begin() + i + n);
        i += n;
    }
    put_be32(z, (b << 16) | a);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8 bits per channel, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    bool ok = std::fwrite(signature, 1, 8, f) == 8 && write_png_chunk(f, "IHDR", ihdr) && write_png_chunk(f, "IDAT", z) &&
              write_png_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

// The tone-mapped image by the name's extension: .png as PNG, anything else as binary PPM.
bool write_image(const std::string &path, const std::vector<uint8_t> &rgb, uint32_t w, uint32_t h)
{
    const bool png = path.size() >= 4 && (path.compare(path.size() - 4, 4, ".png") == 0 || path.compare(path.size() - 4, 4, ".PNG") == 0);
    return png ? write_png(path, rgb, w, h) : write_ppm(path, rgb, w, h);
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
                    if (!write_image(out, multi.root().tonemapped(), width, height)) std::fprintf(stderr, "cannot write %s\n", out.c_str());
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
            if (!write_image(out, multi.root().tonemapped(), width, height)) {
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
                if (!write_image(out, renderer.tonemapped(), width, height)) std::fprintf(stderr, "cannot write %s\n", out.c_str());
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
        if (!write_image(out, renderer.tonemapped(), width, height)) {
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
