// C ABI of everything that reads a renderer's results back (include/pathtracer_amd.h): the image in its several
// forms, checkpoints, primary-hit buffers, and the entry points that run single kernels for the unit tests.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "renderer_state.h"

using namespace pt_internal;

namespace {

// Staging memory of pt_read_tonemapped_noalloc: the 8-bit RGB image the tone-map kernel writes before it is
// copied to the caller.  A file-scope device array instead of a hipMalloc per call, for callers that must not
// allocate.  Every GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; in host code the name below is only a handle for hipGetSymbolAddress, never a device address.
constexpr size_t kNoAllocMaxPixels = size_t(2560) * 1440;
constexpr size_t kNoAllocScratchBytes = 3 * kNoAllocMaxPixels;  // 10.5 MiB per GPU
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, set by scratch_init

}  // namespace

namespace pt_internal {

hipError_t noalloc_staging_init(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    return hipSuccess;
}

}  // namespace pt_internal

namespace {
struct StateHeader {
    char magic[8];  // "PTAMDST3" ("PTAMDST2": the same header with 14 counters)
    uint32_t width, height, max_depth, rank, world, reserved;
    uint64_t local_iterations, paths, scene_hash;
    float fov_y_degrees;
    float eye[3], forward[3], up[3], right[3];
    uint64_t counters[kCounters];  // the device counters behind pt_stats (segments, non-finite samples, ...), so that
                                   // the statistics of a resumed render cover the whole render like `paths` does
};
constexpr size_t kStateHeaderV2Bytes = sizeof(StateHeader) - 2 * sizeof(uint64_t);  // version 2 ends after counters[13]
}  // namespace

extern "C" {

int pt_save_state(pt_renderer *r, const char *path)
{
    PT_REQUIRE(r && path, "pt_save_state: null argument");
    PT_USABLE(r);
    if (r->d_rgba8) return fail(PT_ERR_STATE, "pt_save_state: not available with emulate_rgba8");
    PT_HIP(hipSetDevice(r->device));
    const size_t n = static_cast<size_t>(r->width) * r->height * 4;
    std::vector<float> sum(n);
    int rc = pt_read_sum(r, sum.data());
    if (rc != PT_OK) return rc;
    StateHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "PTAMDST3", 8);
    h.width = r->width;
    h.height = r->height;
    h.max_depth = r->cfg.max_depth;
    h.rank = r->rank;
    h.world = r->world;
    h.local_iterations = r->local_iterations;
    h.paths = r->paths;
    h.scene_hash = r->scene_hash;
    PT_HIP(hipMemcpy(h.counters, r->d_counters, sizeof(h.counters), hipMemcpyDeviceToHost));  // pt_read_sum has synchronised
    h.fov_y_degrees = r->cfg.fov_y_degrees;
    for (int k = 0; k < 3; ++k) {
        h.eye[k] = r->cam.eye[k];
        h.forward[k] = r->cam.forward[k];
        h.up[k] = r->cam.up[k];
        h.right[k] = r->cam.right[k];
    }
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(PT_ERR_IO, std::string("pt_save_state: cannot write ") + tmp);
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(sum.data(), sizeof(float), n, f) == n;
    const bool closed = std::fclose(f) == 0;
    if (!ok || !closed || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(PT_ERR_IO, std::string("pt_save_state: writing ") + path + " failed");
    }
    return PT_OK;
}

int pt_load_state(pt_renderer *r, const char *path)
{
    PT_REQUIRE(r && path, "pt_load_state: null argument");
    PT_USABLE(r);
    if (r->d_rgba8) return fail(PT_ERR_STATE, "pt_load_state: not available with emulate_rgba8");
    PT_HIP(hipSetDevice(r->device));
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(PT_ERR_IO, std::string("pt_load_state: cannot open ") + path);
    StateHeader h;
    const size_t n = static_cast<size_t>(r->width) * r->height * 4;
    std::vector<float> sum(n);
    std::memset(&h, 0, sizeof(h));
    bool ok = std::fread(&h, kStateHeaderV2Bytes, 1, f) == 1;
    if (ok && std::memcmp(h.magic, "PTAMDST1", 8) == 0) {
        // Files of the first format carry no counters and a scene hash computed byte by byte, which this build
        // cannot check against its own (word-wise) hash: refused by name rather than as a damaged file.
        std::fclose(f);
        return fail(PT_ERR_STATE, std::string("pt_load_state: ") + path +
                                      " is a version-1 state file (PTAMDST1); this build reads versions 2 and 3 -- render again and save");
    }
    // version 3 carries two more counters (primary rays culled at generation, last-segment rays not made); a
    // version-2 file has neither, and its render had none of either
    if (ok && std::memcmp(h.magic, "PTAMDST3", 8) == 0)
        ok = std::fread(reinterpret_cast<char *>(&h) + kStateHeaderV2Bytes, sizeof(h) - kStateHeaderV2Bytes, 1, f) == 1;
    else
        ok = ok && std::memcmp(h.magic, "PTAMDST2", 8) == 0;
    const bool fits = ok && h.width == r->width && h.height == r->height;
    ok = fits && std::fread(sum.data(), sizeof(float), n, f) == n && std::fgetc(f) == EOF;
    std::fclose(f);
    if (!fits) return fail(PT_ERR_STATE, std::string("pt_load_state: ") + path + " is not a state file of this image size");
    if (!ok) return fail(PT_ERR_IO, std::string("pt_load_state: ") + path + " is truncated or has trailing data");
    bool same = h.max_depth == r->cfg.max_depth && h.scene_hash == r->scene_hash && h.rank == r->rank &&
                h.world == r->world && h.fov_y_degrees == r->cfg.fov_y_degrees;
    for (int k = 0; k < 3; ++k)
        same = same && h.eye[k] == r->cam.eye[k] && h.forward[k] == r->cam.forward[k] && h.up[k] == r->cam.up[k] &&
               h.right[k] == r->cam.right[k];
    if (!same)
        return fail(PT_ERR_STATE, "pt_load_state: the file was written for another scene, camera, depth or partition");
    const uint64_t next = h.rank + h.local_iterations * static_cast<uint64_t>(h.world);
    if (next > 0xFFFFFFFEull) return fail(PT_ERR_STATE, "pt_load_state: iteration counter out of range");
    int rc = reset_accumulation(r);
    if (rc != PT_OK) return rc;
    PT_HIP(hipMemcpyAsync(r->d_sum, sum.data(), n * sizeof(float), hipMemcpyHostToDevice, r->stream));
    PT_HIP(hipMemcpyAsync(r->d_counters, h.counters, sizeof(h.counters), hipMemcpyHostToDevice, r->stream));
    PT_HIP(sync_streams(r));
    r->local_iterations = h.local_iterations;
    r->paths = h.paths;
    return PT_OK;
}

int pt_read_sum(pt_renderer *r, float *rgba)
{
    PT_REQUIRE(r && rgba, "pt_read_sum: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));  // batches of both streams add to the sum
    PT_HIP(hipMemcpyAsync(rgba, r->d_sum, static_cast<size_t>(r->width) * r->height * sizeof(float4), hipMemcpyDeviceToHost,
                          r->stream));
    PT_HIP(sync_streams(r));
    return PT_OK;
}

int pt_read_tonemapped_noalloc(pt_renderer *r, uint8_t *rgb)
{
    PT_REQUIRE(r && rgb, "pt_read_tonemapped_noalloc: null argument");
    PT_USABLE(r);
    const size_t n = static_cast<size_t>(r->width) * r->height;
    if (n > kNoAllocMaxPixels)
        return fail(PT_ERR_INVALID, "pt_read_tonemapped_noalloc: " + std::to_string(r->width) + " x " + std::to_string(r->height) +
                                        " pixels exceed the staging array (" + std::to_string(kNoAllocMaxPixels) +
                                        " pixels, e.g. 2560 x 1440); use pt_read_tonemapped");
    PT_HIP(hipSetDevice(r->device));
    unsigned char *staging = (r->device >= 0 && r->device < kMaxDevices) ? g_scratch_ptr[r->device] : nullptr;
    if (!staging) return fail(PT_ERR_STATE, "pt_read_tonemapped_noalloc: no staging array on this device");
    PT_HIP(sync_streams(r));
    // the kernel writes all 3 * n bytes the copy reads: nothing of an earlier call's contents is seen
    pt::launch_tonemap(r->stream, shape_for(r, n), r->d_sum, staging, r->width, r->height);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(rgb, staging, 3 * n, hipMemcpyDeviceToHost, r->stream));
    PT_HIP(sync_streams(r));
    return PT_OK;
}

int pt_read_rgba8(pt_renderer *r, uint8_t *rgba)
{
    PT_REQUIRE(r && rgba, "pt_read_rgba8: null argument");
    PT_USABLE(r);
    if (!r->d_rgba8) return fail(PT_ERR_STATE, "pt_read_rgba8: the renderer was created without emulate_rgba8");
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));
    PT_HIP(hipMemcpyAsync(rgba, r->d_rgba8, static_cast<size_t>(r->width) * r->height * 4, hipMemcpyDeviceToHost, r->stream));
    PT_HIP(sync_streams(r));
    return PT_OK;
}

int pt_read_radiance(pt_renderer *r, float *rgba)
{
    int rc = pt_read_sum(r, rgba);
    if (rc != PT_OK) return rc;
    const size_t n = static_cast<size_t>(r->width) * r->height;
    for (size_t p = 0; p < n; ++p) {
        const float cnt = rgba[4 * p + 3];
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
        rgba[4 * p + 0] *= inv;
        rgba[4 * p + 1] *= inv;
        rgba[4 * p + 2] *= inv;
    }
    return PT_OK;
}

int pt_read_tonemapped(pt_renderer *r, uint8_t *rgb)
{
    PT_REQUIRE(r && rgb, "pt_read_tonemapped: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    const size_t n = static_cast<size_t>(r->width) * r->height;
    uint8_t *d = nullptr;
    PT_HIP(sync_streams(r));
    PT_HIP(hipMalloc(&d, 3 * n));
    pt::launch_tonemap(r->stream, shape_for(r, n), r->d_sum, d, r->width, r->height);
    hipError_t e = hipMemcpyAsync(rgb, d, 3 * n, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = sync_streams(r);
    (void)hipFree(d);
    PT_HIP(e);
    return PT_OK;
}

}  // extern "C"

namespace {
// Primary rays of one iteration through the extend kernel, then any of: hit codes, distances, ray directions,
// surface normals, per pixel.
int read_primary(pt_renderer *r, uint32_t iteration, int32_t *ids, float *t, float *dirs, float *normals)
{
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));
    const size_t n = static_cast<size_t>(r->width) * r->height;
    int32_t *d_ids = nullptr;
    float *d_t = nullptr, *d_dirs = nullptr, *d_normals = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        if (d_ids) (void)hipFree(d_ids);
        if (d_t) (void)hipFree(d_t);
        if (d_dirs) (void)hipFree(d_dirs);
        if (d_normals) (void)hipFree(d_normals);
    };
#define PT_TRY(expr)                \
    do {                            \
        e = (expr);                 \
        if (e != hipSuccess) {      \
            cleanup();              \
            PT_HIP(e);              \
        }                           \
    } while (0)
    if (ids) PT_TRY(hipMalloc(&d_ids, n * sizeof(int32_t)));
    if (t) PT_TRY(hipMalloc(&d_t, n * sizeof(float)));
    if (dirs) PT_TRY(hipMalloc(&d_dirs, 3 * n * sizeof(float)));
    if (normals) PT_TRY(hipMalloc(&d_normals, 3 * n * sizeof(float)));
    const pt::LaunchShape g = shape_for(r, n);
    pt::launch_set_batch(r->stream, r->d_params, r->cam, iteration, 0u);
    pt::launch_raygen(r->stream, g, r->view, r->d_params, 1, 1, r->q[0], r->d_counts, region_size_for(r, n));
    pt::launch_extend(r->stream, g, r->view, r->stack_entries, r->cfg.use_bvh != 0, regions_for(r, n, 0, -1), r->q[0],
                      pt::ExtendOutput{true, r->d_hits, nullptr, false, false, r->exact_readbacks, nullptr, r->d_stack_overflow[0]},
                      pt::TraversalCounters{nullptr, nullptr, nullptr}, r->n_nodes, r->scene_in_lds);
    pt::launch_unpack_primary(r->stream, g, static_cast<uint32_t>(n), r->q[0], reinterpret_cast<const float2 *>(r->d_hits), d_ids, d_t,
                              d_dirs, &r->cam);
    if (normals)
        pt::launch_primary_normals(r->stream, g, static_cast<uint32_t>(n), r->view, r->q[0],
                                   reinterpret_cast<const float2 *>(r->d_hits), d_normals, r->cam);
    PT_TRY(hipGetLastError());
    if (normals) PT_TRY(hipMemcpyAsync(normals, d_normals, 3 * n * sizeof(float), hipMemcpyDeviceToHost, r->stream));
    if (ids) PT_TRY(hipMemcpyAsync(ids, d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
    if (t) PT_TRY(hipMemcpyAsync(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost, r->stream));
    if (dirs) PT_TRY(hipMemcpyAsync(dirs, d_dirs, 3 * n * sizeof(float), hipMemcpyDeviceToHost, r->stream));
    PT_TRY(sync_streams(r));
    cleanup();
    return PT_OK;
}
}  // namespace

extern "C" {

int pt_read_hit_ids(pt_renderer *r, uint32_t iteration, int32_t *ids, float *t, float *dirs)
{
    PT_REQUIRE(r && (ids || t || dirs), "pt_read_hit_ids: null argument");
    return read_primary(r, iteration, ids, t, dirs, nullptr);
}

int pt_read_hit_normals(pt_renderer *r, uint32_t iteration, float *normals)
{
    PT_REQUIRE(r && normals, "pt_read_hit_normals: null argument");
    return read_primary(r, iteration, nullptr, nullptr, nullptr, normals);
}

int pt_trace_rays(pt_renderer *r, uint64_t n, const float *origins, const float *dirs, float *t, int32_t *ids)
{
    PT_REQUIRE(r && (n == 0 || (origins && dirs)), "pt_trace_rays: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));
    if (n == 0) return PT_OK;
    const size_t chunk_max = r->capacity;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int32_t *d_ids = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        if (d_o) (void)hipFree(d_o);
        if (d_d) (void)hipFree(d_d);
        if (d_t) (void)hipFree(d_t);
        if (d_ids) (void)hipFree(d_ids);
    };
    const size_t m = static_cast<size_t>(std::min<uint64_t>(n, chunk_max));
    PT_TRY(hipMalloc(&d_o, 3 * m * sizeof(float)));
    PT_TRY(hipMalloc(&d_d, 3 * m * sizeof(float)));
    PT_TRY(hipMalloc(&d_t, m * sizeof(float)));
    PT_TRY(hipMalloc(&d_ids, m * sizeof(int32_t)));
    for (uint64_t off = 0; off < n; off += m) {
        const size_t c = static_cast<size_t>(std::min<uint64_t>(m, n - off));
        PT_TRY(hipMemcpyAsync(d_o, origins + 3 * off, 3 * c * sizeof(float), hipMemcpyHostToDevice, r->stream));
        PT_TRY(hipMemcpyAsync(d_d, dirs + 3 * off, 3 * c * sizeof(float), hipMemcpyHostToDevice, r->stream));
        const pt::LaunchShape g = shape_for(r, c);
        pt::launch_pack_rays(r->stream, g, static_cast<uint32_t>(c), d_o, d_d, r->q[0], r->d_counts, region_size_for(r, c));
        pt::launch_extend(r->stream, g, r->view, r->stack_entries, r->cfg.use_bvh != 0, regions_for(r, c, 0, -1), r->q[0],
                          pt::ExtendOutput{false, r->d_hits, nullptr, false, false, r->exact_readbacks, nullptr, r->d_stack_overflow[0]},
                          pt::TraversalCounters{nullptr, nullptr, nullptr}, r->n_nodes, r->scene_in_lds);
        pt::launch_unpack_primary(r->stream, g, static_cast<uint32_t>(c), r->q[0], reinterpret_cast<const float2 *>(r->d_hits), d_ids,
                                  d_t, nullptr, nullptr);
        PT_TRY(hipGetLastError());
        if (t) PT_TRY(hipMemcpyAsync(t + off, d_t, c * sizeof(float), hipMemcpyDeviceToHost, r->stream));
        if (ids) PT_TRY(hipMemcpyAsync(ids + off, d_ids, c * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
        PT_TRY(sync_streams(r));
    }
    cleanup();
    return PT_OK;
}

int pt_sample_surface(pt_renderer *r, uint64_t n, const float *materials, const float *normals, const float *wo,
                      const float *xi, const uint32_t *seeds, float *wi_out, float *pdf_out, float *f_out,
                      uint32_t *flags_out, uint32_t *seeds_out)
{
    PT_REQUIRE(r && materials && normals && wo && xi && seeds && wi_out && pdf_out && f_out && flags_out && seeds_out,
               "pt_sample_surface: null argument");
    PT_REQUIRE(n < (1ull << 31), "pt_sample_surface: too many records");
    PT_HIP(hipSetDevice(r->device));
    if (n == 0) return PT_OK;
    // One device block: inputs then outputs, all 4-byte elements.
    const size_t in_words = n * (8 + 3 + 3 + 2 + 2), out_words = n * (3 + 1 + 3 + 1 + 2);
    float *d = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        if (d) (void)hipFree(d);
    };
    PT_TRY(hipMalloc(&d, (in_words + out_words) * 4));
    float *d_mat = d, *d_nrm = d_mat + 8 * n, *d_wo = d_nrm + 3 * n, *d_xi = d_wo + 3 * n;
    uint32_t *d_seed = reinterpret_cast<uint32_t *>(d_xi + 2 * n);
    float *d_wi = reinterpret_cast<float *>(d_seed + 2 * n), *d_pdf = d_wi + 3 * n, *d_f = d_pdf + n;
    uint32_t *d_flags = reinterpret_cast<uint32_t *>(d_f + 3 * n), *d_seed_out = d_flags + n;
    PT_TRY(hipMemcpyAsync(d_mat, materials, 8 * n * 4, hipMemcpyHostToDevice, r->stream));
    PT_TRY(hipMemcpyAsync(d_nrm, normals, 3 * n * 4, hipMemcpyHostToDevice, r->stream));
    PT_TRY(hipMemcpyAsync(d_wo, wo, 3 * n * 4, hipMemcpyHostToDevice, r->stream));
    PT_TRY(hipMemcpyAsync(d_xi, xi, 2 * n * 4, hipMemcpyHostToDevice, r->stream));
    PT_TRY(hipMemcpyAsync(d_seed, seeds, 2 * n * 4, hipMemcpyHostToDevice, r->stream));
    pt::launch_sample_surface(r->stream, shape_for(r, n), static_cast<uint32_t>(n), d_mat, d_nrm, d_wo, d_xi, d_seed, d_wi,
                              d_pdf, d_f, d_flags, d_seed_out);
    PT_TRY(hipGetLastError());
    PT_TRY(hipMemcpyAsync(wi_out, d_wi, 3 * n * 4, hipMemcpyDeviceToHost, r->stream));
    PT_TRY(hipMemcpyAsync(pdf_out, d_pdf, n * 4, hipMemcpyDeviceToHost, r->stream));
    PT_TRY(hipMemcpyAsync(f_out, d_f, 3 * n * 4, hipMemcpyDeviceToHost, r->stream));
    PT_TRY(hipMemcpyAsync(flags_out, d_flags, n * 4, hipMemcpyDeviceToHost, r->stream));
    PT_TRY(hipMemcpyAsync(seeds_out, d_seed_out, 2 * n * 4, hipMemcpyDeviceToHost, r->stream));
    PT_TRY(sync_streams(r));
    cleanup();
    return PT_OK;
}
#undef PT_TRY

int pt_read_bvh(pt_renderer *r, float *nodes_host, float *triangles_host)
{
    PT_REQUIRE(r, "pt_read_bvh: null argument");
    PT_HIP(sync_streams(r));
    if (nodes_host && r->n_nodes) PT_HIP(hipMemcpy(nodes_host, r->d_nodes, static_cast<size_t>(r->n_nodes) * 64, hipMemcpyDeviceToHost));
    if (triangles_host && r->view.n_triangles)
        PT_HIP(hipMemcpy(triangles_host, r->d_tris, static_cast<size_t>(r->view.n_triangles) * 64, hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"
