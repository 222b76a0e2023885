// C ABI of the integrator (include/pathtracer_amd.h): host-side scene/camera
// objects, device upload, and the per-batch launch sequence
//   raygen -> [extend -> shade] x max_depth -> accumulate
// on one HIP stream of one GPU.
#include "../../include/pathtracer_amd.h"
#include "tuning.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "bvh.h"
#include "bvh_device.h"
#include "renderer_state.h"

using namespace pt_internal;

namespace {

thread_local std::string g_last_error;

// Inverse of a column-major f32 4x4 evaluated in float64 and rounded to f32
// (the shader inverts per ray, pathtracer.frag.glsl:102; the float64 inverse
// reproduces its light mask).
bool invert_mat4(const float m[16], float inv[16])
{
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = m[c * 4 + r];
            a[r][c + 4] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col)
            for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
        const double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0)
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) inv[c * 4 + r] = static_cast<float>(a[r][c + 4]);
    return true;
}

}  // namespace

namespace pt_internal {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

hipError_t sync_streams(pt_renderer *r)
{
    hipError_t e = hipSuccess;
    if (r->stream2) e = hipStreamSynchronize(r->stream2);
    const hipError_t e1 = r->stream ? hipStreamSynchronize(r->stream) : hipSuccess;
    return e != hipSuccess ? e : e1;
}

}  // namespace pt_internal

namespace {

// Work just enqueued on the main stream touches what the second stream's next batch uses (the sum buffer,
// the counters): that batch will wait for it.
void mark_main_stream_work(pt_renderer *r)
{
    if (!r->two_lanes) return;
    (void)hipEventRecord(r->ev_ops, r->stream);
    r->ops_pending = true;
}

void free_paths(pt_renderer *r)
{
    for (int k = 0; k < 2; ++k) {
        if (r->q2[k].p0) (void)hipFree(r->q2[k].p0);
        if (r->q2[k].p1) (void)hipFree(r->q2[k].p1);
        if (r->q2[k].p2) (void)hipFree(r->q2[k].p2);
        if (r->q2[k].lt) (void)hipFree(r->q2[k].lt);
        r->q2[k] = pt::PathQueue{nullptr, nullptr, nullptr, nullptr};
    }
    if (r->d_hits2) (void)hipFree(r->d_hits2);
    if (r->d_samples2) (void)hipFree(r->d_samples2);
    if (r->d_counts2) (void)hipFree(r->d_counts2);
    r->d_hits2 = r->d_samples2 = nullptr;
    r->d_counts2 = nullptr;
    r->ev_accum_set[0] = r->ev_accum_set[1] = false;
    for (int k = 0; k < 2; ++k) {
        if (r->q[k].p0) (void)hipFree(r->q[k].p0);
        if (r->q[k].p1) (void)hipFree(r->q[k].p1);
        if (r->q[k].p2) (void)hipFree(r->q[k].p2);
        if (r->q[k].lt) (void)hipFree(r->q[k].lt);
        r->q[k] = pt::PathQueue{nullptr, nullptr, nullptr, nullptr};
    }
    if (r->d_hits) (void)hipFree(r->d_hits);
    if (r->d_samples) (void)hipFree(r->d_samples);
    if (r->d_sum) (void)hipFree(r->d_sum);
    if (r->d_rgba8) (void)hipFree(r->d_rgba8);
    r->d_rgba8 = nullptr;
    if (r->d_counts) (void)hipFree(r->d_counts);
    r->d_counts = nullptr;
    r->d_hits = nullptr;
    r->d_samples = nullptr;
    r->d_sum = nullptr;
}

uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    while (b) {
        uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

bool bounces_in_one_launch(const pt_renderer *r, size_t n_paths);

// Samples per pixel that share a batch at this image size; 0 when the image has more pixels than path slots
// can be numbered (32 bits).  (r->bounces_any_size is known: alloc_paths sets it first.)
uint32_t batch_iterations_for(const pt_renderer *r, uint32_t width, uint32_t height)
{
    const size_t n_pixels = static_cast<size_t>(width) * height;
    uint32_t k = r->cfg.iterations_per_batch;
    if (k == 0) {
        // Enough paths in flight to keep 256 CUs busy after compaction has thinned the queue, and launches
        // long enough that their ends weigh little: about 64 M slots (8.6 GB of queues per set of buffers) when
        // all bounces of a batch go in one launch, twice that when every bounce is a launch of its own, each
        // ending in a tail of thinning waves (measured with both: Veggie +1-3 % at 1080p, +5 % at 3840x2160; Box,
        // one launch, -2 %).  At most 128 samples per pixel (Box at 512 x 512: +3 % over 32).
        const size_t target = bounces_in_one_launch(r, ~size_t(0)) ? (size_t(64) << 20) : (size_t(128) << 20);
        k = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(128, target / std::max<size_t>(1, n_pixels))));
    }
    while (k > 1 && n_pixels * k >= (size_t(1) << 31)) --k;
    return n_pixels * k >= (size_t(1) << 31) ? 0u : k;
}

// Frees the path buffers and allocates them for width x height.  On failure the renderer holds no (or only
// part of the) buffers: the caller either restores a size that worked or marks the renderer unusable.
// Slots a region should have, about: with 64 M slots per batch this makes one region per resident wave (12 to 16
// thousand slots; the two-ended hit list packs its counts into 16 bits each, so at most 32768).  The kernels'
// per-region work -- opening a region, draining its last rays, the part-filled batches at both ends of its hit
// list -- is then paid once per wave and bounce, and every wave still gets the same mix of the image because
// consecutive slots are scattered over it.  Measured against regions of 2048 slots: Veggie +19 %, the
// Transmission scenes +7-11 %, Box +3 %, the 10 M-triangle scene +3 %.
size_t region_target_slots()
{
    static const size_t v = [] {
        const char *e = pt::tuning_env("PT_REGION_TARGET");  // tuning experiments only
        return e ? static_cast<size_t>(std::max(64, std::atoi(e))) : size_t(16384);
    }();
    return v;
}

bool bounces_in_one_launch(const pt_renderer *r, size_t n_paths);

int alloc_paths(pt_renderer *r, uint32_t width, uint32_t height)
{
    r->bounces_any_size = r->cfg.use_bvh && pt::bounces_suit_any_batch(r->view, r->stack_entries, r->n_nodes, r->scene_in_lds,
                                                                     r->d_stack_overflow[0] != nullptr);
    const uint32_t k = batch_iterations_for(r, width, height);
    if (k == 0) return fail(PT_ERR_INVALID, "resolution too large");  // nothing has been freed yet
    free_paths(r);
    r->width = width;
    r->height = height;
    const size_t n_pixels = static_cast<size_t>(width) * height;
    r->batch_iterations = k;
    r->capacity = n_pixels * k;
    for (int b = 0; b < 2; ++b) {
        PT_HIP(hipMalloc(&r->q[b].p0, r->capacity * sizeof(float4)));
        PT_HIP(hipMalloc(&r->q[b].p1, r->capacity * sizeof(float4)));
        PT_HIP(hipMalloc(&r->q[b].p2, r->capacity * sizeof(float4)));
        PT_HIP(hipMalloc(&r->q[b].lt, r->capacity * sizeof(float)));
    }
    {
        // Region size.  The kernel that walks the regions runs `waves` persistent waves; every wave should get
        // the same whole number m of regions (a wave with one region more than the others sets the kernel's run
        // time) of about region_target_slots() slots.  size = capacity / (waves * m), rounded up to a multiple of 64.
        r->extend_waves = r->cfg.use_bvh
                              ? pt::extend_resident_waves(r->n_cus, r->view, r->stack_entries, r->n_nodes, r->scene_in_lds,
                                                          r->d_stack_overflow[0] != nullptr)
                              : r->n_cus * 32u;
        r->bounces_waves = r->cfg.use_bvh ? pt::bounces_resident_waves(r->n_cus, r->view, r->stack_entries, r->n_nodes, r->scene_in_lds,
                                                                       r->d_stack_overflow[0] != nullptr)
                                          : r->extend_waves;
        const uint32_t waves = bounces_in_one_launch(r, r->capacity) ? r->bounces_waves : r->extend_waves;
        const size_t target = region_target_slots();
        size_t m = std::max<size_t>(1, (r->capacity + size_t(waves) * (target / 2)) / (size_t(waves) * target));
        size_t size = (r->capacity + waves * m - 1) / (waves * m);
        size = std::max<size_t>(256, (size + 63) & ~size_t(63));
        while (size > 32768) size = ((size / 2) + 63) & ~size_t(63);
        r->region_size = r->region_size_env ? r->region_size_env : static_cast<uint32_t>(size);
    }
    r->region_pitch = static_cast<uint32_t>((r->capacity + r->region_size - 1) / r->region_size);
    // rows 0..depth: live rays per region and bounce; rows depth+1..2*depth: surface hits per region and bounce;
    // row 2*depth+1: primary rays per region that ray generation did not enqueue
    PT_HIP(hipMalloc(&r->d_counts, static_cast<size_t>(2 * r->cfg.max_depth + 2) * r->region_pitch * sizeof(uint32_t)));
    // whole regions: the hit list of a region is filled from both of its ends
    PT_HIP(hipMalloc(&r->d_hits, static_cast<size_t>(r->region_pitch) * r->region_size * sizeof(float4)));
    PT_HIP(hipMalloc(&r->d_samples, r->capacity * sizeof(float4)));
    PT_HIP(hipMemsetAsync(r->d_samples, 0, r->capacity * sizeof(float4), r->stream));  // tag 0 = never written
    r->sample_tag = 0;
    if (r->two_lanes) {
        // the second set of batch buffers; without the memory for it the renderer works with one
        auto second = [&]() -> hipError_t {
            hipError_t e = hipSuccess;
            for (int b = 0; b < 2 && e == hipSuccess; ++b) {
                e = hipMalloc(&r->q2[b].p0, r->capacity * sizeof(float4));
                if (e == hipSuccess) e = hipMalloc(&r->q2[b].p1, r->capacity * sizeof(float4));
                if (e == hipSuccess) e = hipMalloc(&r->q2[b].p2, r->capacity * sizeof(float4));
                if (e == hipSuccess) e = hipMalloc(&r->q2[b].lt, r->capacity * sizeof(float));
            }
            if (e == hipSuccess)
                e = hipMalloc(&r->d_counts2, static_cast<size_t>(2 * r->cfg.max_depth + 2) * r->region_pitch * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(&r->d_hits2, static_cast<size_t>(r->region_pitch) * r->region_size * sizeof(float4));
            if (e == hipSuccess) e = hipMalloc(&r->d_samples2, r->capacity * sizeof(float4));
            if (e == hipSuccess) e = hipMemsetAsync(r->d_samples2, 0, r->capacity * sizeof(float4), r->stream2);
            return e;
        };
        const hipError_t e = second();
        r->sample_tag2 = 0;
        r->lanes_in_use = 2;
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();
            for (int b = 0; b < 2; ++b) {
                (void)hipFree(r->q2[b].p0), (void)hipFree(r->q2[b].p1), (void)hipFree(r->q2[b].p2), (void)hipFree(r->q2[b].lt);
                r->q2[b] = pt::PathQueue{nullptr, nullptr, nullptr, nullptr};
            }
            (void)hipFree(r->d_counts2), (void)hipFree(r->d_hits2), (void)hipFree(r->d_samples2);
            r->d_counts2 = nullptr;
            r->d_hits2 = r->d_samples2 = nullptr;
            r->lanes_in_use = 1;
        } else {
            PT_HIP(e);
        }
    } else {
        r->lanes_in_use = 1;
    }
    PT_HIP(hipMalloc(&r->d_sum, n_pixels * sizeof(float4)));
    if (r->cfg.emulate_rgba8) PT_HIP(hipMalloc(&r->d_rgba8, n_pixels * 4));
    PT_HIP(hipMemsetAsync(r->d_sum, 0, n_pixels * sizeof(float4), r->stream));
    r->cam.width = width;
    r->cam.height = height;
    r->cam.width_f = static_cast<float>(width);
    r->cam.height_f = static_cast<float>(height);
    r->cam.aspect = static_cast<float>(width) / static_cast<float>(height);
    {
        // Slot -> pixel mapping of the queues (kernels.hip, pixel_of_slot): units of 64 pixels, scattered.
        const uint32_t n_px = width * height;
        const bool tiles = (width % 8u == 0u) && (height % 8u == 0u);
        r->cam.map_mode = pt::tuning_env("PT_NO_SCATTER") ? 0u : (tiles ? 2u : 1u);  // (environment: experiments only)
        r->cam.map_units = n_px / 64u;
        r->cam.tiles_x = tiles ? width / 8u : 0u;
        uint32_t stride = static_cast<uint32_t>(r->cam.map_units * 0.6180339887) | 1u;
        while (r->cam.map_units > 1u && gcd_u32(stride, r->cam.map_units) != 1u) stride += 2u;
        r->cam.map_stride = r->cam.map_units > 1u ? stride % r->cam.map_units : 1u;
        if (r->cam.map_stride == 0u || r->cam.map_units < 2u) r->cam.map_mode = 0u;
        r->cam.div_pixels = pt::make_fast_div(n_px);
        r->cam.div_tiles_x = pt::make_fast_div(std::max(1u, r->cam.tiles_x));
        r->cam.div_width = pt::make_fast_div(width);
        r->cam.div_map_units = pt::make_fast_div(std::max(1u, r->cam.map_units));
        if (static_cast<unsigned long long>(r->cam.map_units) * r->cam.map_stride >= (1ull << 31))
            r->cam.div_map_units.magic = 0xFFFFFFFFu;  // the product does not fit: pixel_of_slot divides in 64 bits
    }
    return PT_OK;
}


// A small batch (the reference's one sample per draw) spends its time starting and draining 2 * depth - 1
// kernels over thinning queues: all its bounces go in one launch instead (k_bounces) -- and so do the large
// batches of a scene that kernel suits at any size -- unless one launch per kernel (profile bit 4), the
// counters or the exact-samples pass are asked for.  Same results either way.
constexpr size_t kFusedMaxPaths = size_t(1) << 23;  // measured on Box: one launch wins up to about 8 M paths (+15 % at 2 M, +2 % at 8 M), loses 2 % at 16 M
bool bounces_in_one_launch(const pt_renderer *r, size_t n_paths)
{
    static const size_t fused_max_paths = [] {
        const char *e = pt::tuning_env("PT_FUSED_MAX");  // tuning experiments only
        return e ? static_cast<size_t>(std::atoll(e)) : kFusedMaxPaths;
    }();
    const bool exact_samples = r->cfg.emulate_rgba8 || (r->cfg.profile & 4u);
    static const bool never_any_size = pt::tuning_env("PT_FUSED_SMALL_ONLY") != nullptr;  // tuning experiments only
    return r->cfg.use_bvh && !exact_samples && !(r->cfg.profile & (2u | 16u)) &&
           (n_paths <= fused_max_paths || (r->bounces_any_size && !never_any_size));
}

}  // namespace

namespace pt_internal {

// Slots per region for a launch over n_items queue slots.  A full batch uses the size the buffers were laid out
// for; a smaller one (the reference's one sample per draw) gets smaller regions, so that it still spreads over
// all resident waves: the same rule as in alloc_paths, within what the count rows and the hit list can hold.
uint32_t region_size_for(const pt_renderer *r, size_t n_items)
{
    if (n_items >= r->capacity || r->region_size_env) return r->region_size;
    // as many equal regions as the kernel that will walk them keeps waves resident
    const bool fused = bounces_in_one_launch(r, n_items);
    const size_t waves = std::max<uint32_t>(1, fused ? r->bounces_waves : r->extend_waves);
    const size_t target = region_target_slots();
    size_t m = std::max<size_t>(1, (n_items + waves * (target / 2)) / (waves * target));
    size_t size = (n_items + waves * m - 1) / (waves * m);
    static const size_t min_region_env = [] {
        const char *e = pt::tuning_env("PT_MIN_REGION");  // tuning experiments only
        return e ? static_cast<size_t>(std::atoi(e)) : size_t(0);
    }();
    // (in one launch a wave walks its regions through every bounce: shorter regions, more waves at work)
    const size_t min_region = min_region_env ? min_region_env : (fused ? size_t(128) : size_t(256));
    size = std::max<size_t>(min_region, (size + 63) & ~size_t(63));
    // no more regions than a row of counts has entries
    const size_t at_least = (n_items + r->region_pitch - 1) / std::max<uint32_t>(1, r->region_pitch);
    size = std::max(size, (at_least + 63) & ~size_t(63));
    size = std::min<size_t>(size, r->region_size);
    // the hit list holds region_pitch regions of the full size: whole regions of the smaller size must fit in it
    if (((n_items + size - 1) / size) * size > static_cast<size_t>(r->region_pitch) * r->region_size) return r->region_size;
    return static_cast<uint32_t>(size);
}

// Region bookkeeping for a launch over n_items queue slots: wave w visits the
// regions (k * stride) mod n_regions, k = w, w + n_waves, ...; a stride near
// the golden ratio of n_regions, coprime to it, scatters neighbouring image
// areas over different waves.
pt::Regions regions_for(const pt_renderer *r, size_t n_items, uint32_t row_in, int row_out, uint32_t *counts)
{
    if (!counts) counts = r->d_counts;
    pt::Regions g;
    g.size = region_size_for(r, n_items);
    g.n_regions = static_cast<uint32_t>(std::max<size_t>(1, (n_items + g.size - 1) / g.size));
    uint32_t stride = static_cast<uint32_t>(g.n_regions * 0.6180339887) | 1u;
    while (gcd_u32(stride, g.n_regions) != 1u) stride += 2u;
    g.stride = g.n_regions == 1 ? 1u : stride % g.n_regions;
    if (g.stride == 0) g.stride = 1;
    g.counts_in = counts + static_cast<size_t>(row_in) * r->region_pitch;
    g.counts_out = row_out < 0 ? nullptr : counts + static_cast<size_t>(row_out) * r->region_pitch;
    return g;
}

pt::LaunchShape shape_for(const pt_renderer *r, size_t work_items)
{
    const size_t blocks_needed = (work_items + 255) / 256;
    const size_t cap = static_cast<size_t>(r->n_cus) * 8;
    const size_t regions = (work_items + region_size_for(r, work_items) - 1) / region_size_for(r, work_items);
    return pt::LaunchShape{static_cast<uint32_t>(std::max<size_t>(1, std::min(blocks_needed, cap))), r->n_cus,
                           static_cast<uint32_t>(std::max<size_t>(1, regions))};
}

}  // namespace pt_internal

namespace {

// The window of pixels outside of which no primary ray can meet the scene's bounds or a light
// (DeviceCamera::cull_*; kernels.hip, raygen_regions).  A pixel's rays go through image-plane points
// forward + H * sx + V * sy with H = right * tan * aspect, V = up * tan and (sx, sy) inside the pixel's square
// (frag:306-320); a world point P is seen at the (sx, sy) that solves forward + H sx + V sy = lambda (P - eye),
// lambda > 0.  All points in front of the eye: the convex hull of their images holds the image of everything
// between them, and its bounding rectangle -- widened by a pixel and by what the traversal's padded slabs let
// through -- is the window.  Any point beside or behind the eye, or a frame that cannot be inverted: the whole image.
void update_cull_window(pt_renderer *r)
{
    pt::DeviceCamera &c = r->cam;
    c.cull_x0 = c.cull_y0 = 0;
    c.cull_x1 = r->width ? r->width - 1 : 0;
    c.cull_y1 = r->height ? r->height - 1 : 0;
    if (r->cull_points.empty() || r->width == 0 || r->height == 0) return;
    const double tan_h = c.tan_half_fov, aspect = c.aspect;
    double H[3], V[3], F[3], E[3];
    for (int k = 0; k < 3; ++k) {
        H[k] = double(c.right[k]) * tan_h * aspect;
        V[k] = double(c.up[k]) * tan_h;
        F[k] = c.forward[k];
        E[k] = c.eye[k];
    }
    auto det3 = [](const double a[3], const double b[3], const double d[3]) {
        return a[0] * (b[1] * d[2] - b[2] * d[1]) - b[0] * (a[1] * d[2] - a[2] * d[1]) + d[0] * (a[1] * b[2] - a[2] * b[1]);
    };
    double sx_lo = 1e300, sx_hi = -1e300, sy_lo = 1e300, sy_hi = -1e300, lambda_max = 0.0;
    bool any_in_front = false, any_behind = false;
    for (const std::array<double, 3> &P : r->cull_points) {
        // [H V -v] (sx, sy, lambda)^T = -F
        const double nv[3] = {-(P[0] - E[0]), -(P[1] - E[1]), -(P[2] - E[2])}, nF[3] = {-F[0], -F[1], -F[2]};
        const double det = det3(H, V, nv);
        if (!(std::fabs(det) > 1e-30) || !std::isfinite(det)) return;
        const double sx = det3(nF, V, nv) / det, sy = det3(H, nF, nv) / det, lambda = det3(H, V, nF) / det;
        if (!std::isfinite(lambda) || !std::isfinite(sx) || !std::isfinite(sy)) return;
        // lambda <= 0: the point lies in or behind the plane through the eye parallel to the image plane, where
        // no primary ray goes (every ray direction forward + H sx + V sy has the same positive component across it)
        if (!(lambda > 1e-9)) {
            any_behind = true;
            continue;
        }
        any_in_front = true;
        sx_lo = std::min(sx_lo, sx), sx_hi = std::max(sx_hi, sx);
        sy_lo = std::min(sy_lo, sy), sy_hi = std::max(sy_hi, sy);
        lambda_max = std::max(lambda_max, lambda);
    }
    auto empty_window = [&]() { c.cull_x0 = 1, c.cull_x1 = 0, c.cull_y0 = 1, c.cull_y1 = 0; };
    if (!any_in_front) {
        // everything lies behind that plane, and so does all that is between the points: nothing is in view.
        // (Points within the padding of the plane count as in front: lambda > 1e-9 is a test with a wide margin,
        // since 1 / lambda is the point's distance in units of |forward|.)
        empty_window();
        return;
    }
    if (any_behind) return;  // the bounds straddle the plane: their image is unbounded, the whole frame stays
    // The traversal widens every slab by 2^-17 * max(|origin|, scene radius) (kernels.hip, extend_body): a ray may
    // pass that far outside the box and still be tested against what is inside.  Seen from the eye at the nearest
    // point's distance 1 / lambda_max that is an offset of pad * lambda_max on the image plane at distance 1, i.e.
    // pad * lambda_max / tan in units of (sx, sy); taken sixteen times over, plus one whole pixel.
    double reach = r->view.scene_radius;
    for (int k = 0; k < 3; ++k) reach = std::max(reach, std::fabs(E[k]));
    const double pad_s = 16.0 * (reach * 7.62939453125e-6) * lambda_max / std::max(1e-12, tan_h * std::min(1.0, aspect));
    const double W = r->width, Hh = r->height;
    // pixel column x covers fragcoord + jitter in [x + 0.5, x + 1.5] (jitter in [0, 1], frag:310), sx = 2 (fragcoord + jitter) / W - 1
    const double fx_lo = (sx_lo - pad_s + 1.0) * 0.5 * W - 1.0, fx_hi = (sx_hi + pad_s + 1.0) * 0.5 * W + 1.0;
    const double fy_lo = (sy_lo - pad_s + 1.0) * 0.5 * Hh - 1.0, fy_hi = (sy_hi + pad_s + 1.0) * 0.5 * Hh + 1.0;
    auto lower = [](double f, double n) { return static_cast<uint32_t>(std::min(n, std::max(0.0, std::floor(f - 1.5)))); };  // at or before the first x with x + 1.5 >= f
    auto upper = [](double f, double n) { return std::min(n - 1.0, std::ceil(f - 0.5)); };                                 // at or after the last x with x + 0.5 <= f
    const double ux = upper(fx_hi, W), uy = upper(fy_hi, Hh);
    c.cull_x0 = lower(fx_lo, W);
    c.cull_y0 = lower(fy_lo, Hh);
    if (ux < 0.0 || uy < 0.0 || c.cull_x0 >= r->width || c.cull_y0 >= r->height) {
        empty_window();  // everything lies outside the frame
        return;
    }
    c.cull_x1 = static_cast<uint32_t>(ux);
    c.cull_y1 = static_cast<uint32_t>(uy);
}

}  // namespace

namespace pt_internal {

int reset_accumulation(pt_renderer *r)
{
    update_cull_window(r);  // every change of camera, size or scene lights comes through here
    if (r->stream2) PT_HIP(hipStreamSynchronize(r->stream2));  // its last accumulate still adds to the sum
    PT_HIP(hipMemsetAsync(r->d_sum, 0, static_cast<size_t>(r->width) * r->height * sizeof(float4), r->stream));
    PT_HIP(hipMemsetAsync(r->d_counters, 0, kCounters * sizeof(unsigned long long), r->stream));
    if (r->d_rgba8) PT_HIP(hipMemsetAsync(r->d_rgba8, 0, static_cast<size_t>(r->width) * r->height * 4, r->stream));
    mark_main_stream_work(r);
    r->local_iterations = 0;
    r->paths = 0;
    for (int k = 0; k < 5; ++k) {
        r->ms[k] = 0;
        r->launches[k] = 0;
    }
    return PT_OK;
}

}  // namespace pt_internal

namespace {

// Brackets one launch with events when profiling is on.
struct Timed {
    pt_renderer *r;
    int slot;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(pt_renderer *r_, int slot_, hipStream_t stream_) : r(r_), slot(slot_), stream(stream_)
    {
        if (!(r->cfg.profile & 1u)) return;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, stream);
    }
    ~Timed()
    {
        if (!(r->cfg.profile & 1u)) return;
        (void)hipEventRecord(b, stream);
        r->timed.push_back(pt_renderer::TimedLaunch{a, b, slot});
    }
};

void drain_events(pt_renderer *r)
{
    for (const pt_renderer::TimedLaunch &t : r->timed) {
        float ms = 0.f;
        if (hipEventSynchronize(t.end) == hipSuccess && hipEventElapsedTime(&ms, t.begin, t.end) == hipSuccess) {
            r->ms[t.kernel] += ms;
            r->launches[t.kernel] += 1;
        }
        (void)hipEventDestroy(t.begin);
        (void)hipEventDestroy(t.end);
    }
    r->timed.clear();
}

// Whether the batches' ray generation leaves out the pixels outside the camera's window.  (Not with the
// exact-samples pass: a ray that misses the scene may still be the reference's hit at t = inf; the scan kernel
// has no window either.)
bool primary_rays_culled(const pt_renderer *r)
{
    const bool exact_samples = r->cfg.emulate_rgba8 || (r->cfg.profile & 4u);
    return !exact_samples && r->cfg.use_bvh != 0;
}

// The launches of one batch up to (not including) its accumulate kernel: raygen, then extend / shade per bounce.
// Everything that differs from batch to batch is read from the lane's BatchParams block, so for a given lane and
// batch size these launches are always the same.
void enqueue_bounces(pt_renderer *r, uint32_t lane, uint32_t kb, hipStream_t st)
{
    const uint32_t depth = r->cfg.max_depth;
    const size_t n_paths = static_cast<size_t>(r->width) * r->height * kb;
    pt::TraversalCounters tc{nullptr, nullptr, nullptr};
    if (r->cfg.profile & 2u) tc = pt::TraversalCounters{r->d_counters + 2, r->d_counters + 3, r->d_counters + 4};
    const pt::LaunchShape g = shape_for(r, n_paths);
    pt::PathQueue *q = lane ? r->q2 : r->q;
    float4 *d_hits = lane ? r->d_hits2 : r->d_hits, *d_samples = lane ? r->d_samples2 : r->d_samples;
    uint32_t *d_counts = lane ? r->d_counts2 : r->d_counts;
    const pt::BatchParams *bp = r->d_params + lane;
    // The reference's non-finite samples one for one (the 8-bit emulation needs them, profile bit 2 asks for
    // them): rays that escape are checked for planes they are exactly parallel to (a "hit" at t = inf in the
    // reference, a NaN sample), and the BSDF is evaluated at the last surface too.
    const bool exact_samples = r->cfg.emulate_rgba8 || (r->cfg.profile & 4u);
    if (bounces_in_one_launch(r, n_paths)) {
        const pt::Regions rg = regions_for(r, n_paths, 0, -1, d_counts);
        pt::BounceLoop loop{{q[0], q[1]}, d_hits, d_samples, d_counts, r->region_pitch, rg.n_regions, rg.stride, rg.size, depth,
                            r->d_counters + 15};
        Timed t(r, 4, st);
        pt::launch_bounces(st, g, r->view, r->stack_entries, loop, bp, r->n_nodes, r->scene_in_lds, r->d_stack_overflow[lane], r->world, kb);
        return;
    }
    {
        // (with the exact-samples pass a ray that misses the scene may still be a hit at t = inf: nothing is culled)
        Timed t(r, 0, st);
        pt::launch_raygen_regions(st, g, r->view, bp, r->world, kb, q[0], regions_for(r, n_paths, 0, 0, d_counts),
                                  d_counts + static_cast<size_t>(2 * depth + 1) * r->region_pitch, primary_rays_culled(r));
    }
    for (uint32_t b = 0; b < depth; ++b) {
        const pt::PathQueue &in = q[b & 1], &out = q[(b + 1) & 1];
        // The last surface hit needs no new direction -- except that the reference still evaluates the BSDF
        // there, and a NaN it produces becomes the path's sample: the 8-bit emulation runs that last shade too,
        // and so does a renderer asked to count non-finite samples exactly (profile bit 2).
        const bool final_shade = (b + 1 == depth) && exact_samples;
        const bool last = (b + 1 == depth) && !final_shade;
        const int hit_row = static_cast<int>(depth + 1 + b);
        {
            Timed t(r, 1, st);
            pt::ExtendOutput eo{true, d_hits, d_samples, true, last, exact_samples, bp, r->d_stack_overflow[lane]};
            pt::launch_extend(st, g, r->view, r->stack_entries, r->cfg.use_bvh != 0, regions_for(r, n_paths, b, hit_row, d_counts), in,
                              eo, tc, r->n_nodes, r->scene_in_lds);
        }
        if (!last) {
            // The rays this bounce makes for the last allowed segment can only matter if they reach the light (a
            // surface hit there is not shaded): with one light, those that do not point at it are not made.  Not
            // with the exact-samples pass, which does shade the last surface.
            const bool before_last = (b + 2 == depth) && !exact_samples && r->view.n_lights == 1u;
            Timed t(r, 2, st);
            pt::launch_shade(st, g, r->view,
                             regions_for(r, n_paths, static_cast<uint32_t>(hit_row), static_cast<int>(b) + 1, d_counts), in, d_hits,
                             out, d_samples, bp, final_shade, before_last ? r->d_counters + 15 : nullptr);
        }
    }
}

// One batch of `kb` iterations starting at partition index `first_k`.
int render_batch(pt_renderer *r, uint64_t first_k, uint32_t kb)
{
    const uint32_t depth = r->cfg.max_depth;
    const size_t n_pixels = static_cast<size_t>(r->width) * r->height;
    const size_t n_paths = n_pixels * kb;
    const uint32_t it_first = static_cast<uint32_t>(r->rank + first_k * r->world);
    // the set of batch buffers this batch runs in, and its stream
    const uint32_t lane = r->lanes_in_use == 2u ? (r->next_lane++ & 1u) : 0u;
    hipStream_t st = lane ? r->stream2 : r->stream;
    float4 *d_samples = lane ? r->d_samples2 : r->d_samples;
    uint32_t *d_counts = lane ? r->d_counts2 : r->d_counts;
    uint32_t &tag = lane ? r->sample_tag2 : r->sample_tag;
    if (lane && r->ops_pending) {
        PT_HIP(hipStreamWaitEvent(st, r->ev_ops, 0));
        r->ops_pending = false;
    }
    if (++tag == 0u) tag = 1u;  // 0 is the tag of the cleared buffer
    pt::launch_set_batch(st, r->d_params + lane, r->cam, it_first, tag);

    // (Replaying these launches as a HIP graph was measured on ROCm 7.2 with one-sample draws at 512^2 ... 1080p:
    // frame time and draw-to-synchronize latency unchanged to the microsecond, so they are issued one by one.)
    enqueue_bounces(r, lane, kb, st);
    {
        // sum += samples in batch order: wait for the other set's accumulate, which belongs to the batch before
        if (r->ev_accum_set[lane ^ 1u]) PT_HIP(hipStreamWaitEvent(st, r->ev_accum[lane ^ 1u], 0));
        static const bool accumulate_reads_all = pt::tuning_env("PT_ACCUMULATE_ALL") != nullptr;  // tuning experiments only
        {
            Timed t(r, 3, st);
            pt::launch_accumulate(st, shape_for(r, n_pixels), d_samples, r->d_sum, static_cast<uint32_t>(n_pixels), kb, d_counts,
                                  depth, regions_for(r, n_paths, 0, -1, d_counts).n_regions, r->region_pitch, r->d_counters,
                                  r->d_counters + 12, r->d_counters + 1, r->d_counters + 14, r->d_rgba8, r->d_params + lane, r->world,
                                  primary_rays_culled(r) && !accumulate_reads_all);
        }
        if (r->two_lanes) {
            PT_HIP(hipEventRecord(r->ev_accum[lane], st));
            r->ev_accum_set[lane] = true;
        }
    }
    PT_HIP(hipGetLastError());
    if ((r->cfg.profile & 1u) && r->timed.size() > 8192) {
        PT_HIP(sync_streams(r));
        drain_events(r);
    }
    r->local_iterations += kb;
    r->paths += n_paths;
    return PT_OK;
}

// 64-bit FNV-1a over 8-byte words in four interleaved lanes (a byte-wise hash of a
// gigabyte of scene data would take longer than building its tree), the tail byte-wise.
uint64_t fnv1a(uint64_t h, const void *data, size_t bytes)
{
    constexpr uint64_t kPrime = 1099511628211ull;
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t lane[4] = {h, h ^ 0x9E3779B97F4A7C15ull, h ^ 0xC2B2AE3D27D4EB4Full, h ^ 0x165667B19E3779F9ull};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) lane[k] = (lane[k] ^ w[k]) * kPrime;
    }
    for (; i < bytes; ++i) lane[0] = (lane[0] ^ p[i]) * kPrime;
    uint64_t out = lane[0];
    for (int k = 1; k < 4; ++k) out = (out ^ lane[k]) * kPrime;
    return out;
}

int upload_scene(pt_renderer *r, const pt_scene_desc *sc)
{
    const uint32_t T = sc->n_triangles;
    const bool timing = pt::tuning_env("PT_TIMING") != nullptr;  // stage times of this function on stderr
    double t_stage = now_ms();
    auto stage = [&](const char *what) {
        if (timing) std::fprintf(stderr, "upload_scene: %-28s %8.1f ms\n", what, now_ms() - t_stage);
        t_stage = now_ms();
    };
    {
        uint64_t h = 1469598103934665603ull;
        h = fnv1a(h, sc->vertices, static_cast<size_t>(sc->n_vertices) * 12);
        h = fnv1a(h, sc->indices, static_cast<size_t>(T) * 16);
        h = fnv1a(h, sc->vertex_data, static_cast<size_t>(T) * 3 * 24);
        h = fnv1a(h, sc->lights, static_cast<size_t>(sc->n_lights) * 80);
        h = fnv1a(h, sc->materials, static_cast<size_t>(sc->n_materials) * 32);
        h = fnv1a(h, sc->material_map, static_cast<size_t>(T) * 4);
        r->scene_hash = h;
    }
    stage("scene hash");
    std::vector<uint32_t> ids(static_cast<size_t>(T) * 3);
    for (uint32_t t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            const float f = sc->indices[4 * static_cast<size_t>(t) + k];
            if (!(f >= 0.f) || f >= static_cast<float>(sc->n_vertices) || f != std::floor(f))
                return fail(PT_ERR_INVALID, "triangle " + std::to_string(t) + " references a vertex out of range");
            ids[3 * static_cast<size_t>(t) + k] = static_cast<uint32_t>(f);
        }

    stage("vertex indices");
    auto upload = [&](void **dst, const void *src, size_t bytes) -> int {
        PT_HIP(hipMalloc(dst, std::max<size_t>(bytes, 64)));
        if (bytes) PT_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return PT_OK;
    };
    int rc;
    if ((rc = upload(&r->d_materials, sc->materials, static_cast<size_t>(sc->n_materials) * 32))) return rc;

    const double t0 = now_ms();
    pt::Bvh bvh;
    // bvh_builder 2: the device builder for scenes large enough to repay its launches, and the
    // host builder for input the device builder declines.
    constexpr uint32_t kAutoDeviceTriangles = 1u << 16;
    const bool automatic = r->cfg.bvh_builder == 2u;
    bool device_build = r->cfg.use_bvh && T > 0 && (r->cfg.bvh_builder == 1u || (automatic && T >= kAutoDeviceTriangles));
    float bounds_lo[3], bounds_hi[3];
    double device_upload_ms = 0.0;  // device builder: the scene arrays go up before the tree is made
    void *d_vertices = nullptr, *d_ids = nullptr, *d_map = nullptr, *d_indices = nullptr, *d_vertex_data = nullptr;
    auto drop = [&]() {
        (void)hipFree(d_vertices);
        (void)hipFree(d_ids);
        (void)hipFree(d_map);
        (void)hipFree(d_indices);
        (void)hipFree(d_vertex_data);
        d_vertices = d_ids = d_map = d_indices = d_vertex_data = nullptr;
    };
    pt::DeviceBvh built;
    if (device_build) {
        // The triangles go up as they are; the tree is made where the renderer will read it.
        if ((rc = upload(&d_vertices, sc->vertices, static_cast<size_t>(sc->n_vertices) * 12)) ||
            (rc = upload(&d_ids, ids.data(), ids.size() * 4)) ||
            (rc = upload(&d_map, sc->material_map, static_cast<size_t>(T) * 4)) ||
            (rc = upload(&d_indices, sc->indices, static_cast<size_t>(T) * 16)) ||
            (rc = upload(&d_vertex_data, sc->vertex_data, static_cast<size_t>(T) * 72))) {
            drop();
            return rc;
        }
        if (hipMalloc(&r->d_shade, std::max<size_t>(static_cast<size_t>(T) * sizeof(pt::ShadeTriangle), 64)) != hipSuccess) {
            drop();
            return fail(PT_ERR_DEVICE, "out of device memory for the shading records");
        }
        device_upload_ms = now_ms() - t0;
        pt::build_shade_records(static_cast<const float *>(d_vertices), static_cast<const uint32_t *>(d_ids),
                                static_cast<const float *>(d_indices), static_cast<const float *>(d_vertex_data),
                                static_cast<const uint32_t *>(d_map), T, r->d_shade, r->stream);
        pt::DeviceBvhParams params;
        params.leaf_max = pt::bvh_leaf_max();
        params.traversal_cost = pt::bvh_traversal_cost();
        std::string why;
        if (!pt::build_bvh_device(static_cast<const float *>(d_vertices), static_cast<const uint32_t *>(d_ids), T, params,
                                  r->stream, &built, &why)) {
            drop();
            if (!automatic) return fail(PT_ERR_STATE, why);
            (void)hipStreamSynchronize(r->stream);
            (void)hipFree(r->d_shade);
            r->d_shade = nullptr;
            device_build = false;
            device_upload_ms = 0.0;
        }
    }
    if (device_build) {
        pt::mark_triangle_classes(built.d_triangles, T, static_cast<const uint32_t *>(d_map),
                                  static_cast<const float *>(r->d_materials), sc->n_materials, r->stream);
        const hipError_t e = hipStreamSynchronize(r->stream);
        drop();
        r->d_nodes = built.d_nodes;
        r->d_tris = built.d_triangles;
        if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("device BVH build: ") + hipGetErrorString(e));
        r->n_nodes = built.n_nodes;
        r->bvh_depth = built.depth;
        r->sah_cost = built.sah_cost;
        for (int a = 0; a < 3; ++a) bounds_lo[a] = built.bounds_lo[a], bounds_hi[a] = built.bounds_hi[a];
    }
    if (!device_build) {
        // automatic choice: a scene small enough for the host builder gets the best of several bin counts
        if (automatic && r->cfg.use_bvh && T < kAutoDeviceTriangles) {
            pt::build_bvh_best_of(sc->vertices, ids.data(), T, 64, &bvh);
            pt::order_leaves_by_area(&bvh);  // what the traversal kernel keeps of the triangles in LDS is a prefix
        } else
            pt::build_bvh(sc->vertices, ids.data(), T, 64, &bvh);
        if (r->cfg.use_bvh) {
            if (bvh.depth > 64) return fail(PT_ERR_STATE, "BVH deeper than the largest traversal stack");
        } else {
            // Scan mode: triangle records in the reference's order, no tree.
            std::vector<pt::BvhTriangle> ordered(T);
            for (const pt::BvhTriangle &tri : bvh.triangles) ordered[tri.original] = tri;
            bvh.triangles.swap(ordered);
        }
        r->n_nodes = static_cast<uint32_t>(bvh.nodes.size());
        r->bvh_depth = bvh.depth;
        r->sah_cost = bvh.sah_cost;
        for (int a = 0; a < 3; ++a) bounds_lo[a] = bvh.bounds_lo[a], bounds_hi[a] = bvh.bounds_hi[a];
    }
    r->bvh_build_ms = now_ms() - t0 - device_upload_ms;
    stage("BVH");
    r->bvh_builder = device_build ? 1u : 0u;
    r->stack_entries = std::max<uint32_t>(4, r->bvh_depth + 1u);  // traversal stack: the end marker + one entry per inner level

    std::vector<pt::ShadeTriangle> shade(device_build ? 0 : T);
    for (uint32_t t = 0; t < static_cast<uint32_t>(shade.size()); ++t) {
        const float *v[3];
        for (int k = 0; k < 3; ++k) v[k] = sc->vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + k]);
        // The reference reads the normals of record int(indices.w) (frag:437,447-449).
        const float wf = sc->indices[4 * static_cast<size_t>(t) + 3];
        const size_t data = (wf >= 0.f && wf < static_cast<float>(T)) ? static_cast<size_t>(wf) : t;
        const float *n1 = sc->vertex_data + 6 * (3 * data + 0);
        const float *n2 = sc->vertex_data + 6 * (3 * data + 1);
        const float *n3 = sc->vertex_data + 6 * (3 * data + 2);
        float *s = shade[t].s;
        for (int k = 0; k < 3; ++k) {
            s[4 * k + 0] = v[k][0];
            s[4 * k + 1] = v[k][1];
            s[4 * k + 2] = v[k][2];
            s[4 * k + 3] = n1[k];
        }
        s[12] = n2[0];
        s[13] = n2[1];
        s[14] = n2[2];
        std::memcpy(&s[15], &sc->material_map[t], 4);
        s[16] = n3[0];
        s[17] = n3[1];
        s[18] = n3[2];
        s[19] = 0.f;
    }

    std::vector<pt::DeviceLight> lights(std::max<uint32_t>(1, sc->n_lights));
    std::memset(lights.data(), 0, lights.size() * sizeof(pt::DeviceLight));
    for (uint32_t l = 0; l < sc->n_lights; ++l) {
        const float *rec = sc->lights + 20 * static_cast<size_t>(l);
        std::memcpy(lights[l].radiance, rec, 16);
        if (!invert_mat4(rec + 4, lights[l].inv))
            for (float &x : lights[l].inv) x = NAN;
        // transform * (0,0,1,0) = third column
        lights[l].normal[0] = rec[4 + 8];
        lights[l].normal[1] = rec[4 + 9];
        lights[l].normal[2] = rec[4 + 10];
    }

    stage("shading records, lights");
    const double t1 = now_ms();
    if (!device_build) {
        if ((rc = upload(&r->d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(pt::BvhNode)))) return rc;
        // Sorting hint for the shade kernel: bit 30 of a record's triangle index marks surfaces that are not
        // purely diffuse (scene.h material model: metallic > 0 or roughness < 1 reach the microfacet lobes).
        for (pt::BvhTriangle &tri : bvh.triangles) {
            const uint32_t m = sc->material_map[tri.original];
            bool diffuse = false;
            if (m < sc->n_materials) {
                const float *mat = sc->materials + 8 * static_cast<size_t>(m);
                diffuse = mat[4] <= 0.f && mat[3] >= 1.f;  // metallic, roughness
            }
            if (!diffuse) tri.original |= pt::kHitClassBit;
        }
        if ((rc = upload(&r->d_tris, bvh.triangles.data(), bvh.triangles.size() * sizeof(pt::BvhTriangle)))) return rc;
    }
    if (!device_build && (rc = upload(&r->d_shade, shade.data(), shade.size() * sizeof(pt::ShadeTriangle)))) return rc;
    if ((rc = upload(&r->d_lights, lights.data(), lights.size() * sizeof(pt::DeviceLight)))) return rc;
    r->upload_ms = now_ms() - t1 + device_upload_ms;
    stage("uploads");

    r->view.nodes = static_cast<const float4 *>(r->d_nodes);
    r->view.triangles = static_cast<const float4 *>(r->d_tris);
    r->view.shade = static_cast<const float4 *>(r->d_shade);
    r->view.materials = static_cast<const float4 *>(r->d_materials);
    r->view.lights = static_cast<const pt::DeviceLight *>(r->d_lights);
    r->view.n_triangles = T;
    r->view.n_materials = sc->n_materials;
    r->view.n_lights = sc->n_lights;
    float radius = 0.f;
    for (int a = 0; a < 3; ++a) radius = std::max(radius, std::max(std::fabs(bounds_lo[a]), std::fabs(bounds_hi[a])));
    r->view.scene_radius = radius;
    r->cull_points.clear();
    for (int c = 0; c < 8; ++c)
        r->cull_points.push_back({double((c & 1) ? bounds_hi[0] : bounds_lo[0]), double((c & 2) ? bounds_hi[1] : bounds_lo[1]),
                                  double((c & 4) ? bounds_hi[2] : bounds_lo[2])});
    for (uint32_t l = 0; l < sc->n_lights; ++l) {
        // the light is the unit square |x|, |y| <= 0.5 of its own z = 0 plane (frag:363-382); column-major transform
        const float *m = sc->lights + 20 * static_cast<size_t>(l) + 4;
        for (int c = 0; c < 4; ++c) {
            const double lx = (c & 1) ? 0.5 : -0.5, ly = (c & 2) ? 0.5 : -0.5;
            const double w = m[3] * lx + m[7] * ly + m[15];
            r->cull_points.push_back({(m[0] * lx + m[4] * ly + m[12]) / w, (m[1] * lx + m[5] * ly + m[13]) / w,
                                      (m[2] * lx + m[6] * ly + m[14]) / w});
        }
    }
    return PT_OK;
}

// tan(FOVY * 0.5f) with FOVY = deg * PI / 180.f in f32 (frag:306,315).
float tan_half_fov(float degrees)
{
    const float fovy = degrees * 3.14159265358979323f / 180.f;
    return std::tan(fovy * 0.5f);
}

}  // namespace

namespace pt_internal {

bool sum_buffer(pt_renderer *r, SumBuffer *out)
{
    if (!r || !r->usable || !r->d_sum) {
        fail(PT_ERR_STATE, "the renderer has no image buffers");
        return false;
    }
    if (hipSetDevice(r->device) != hipSuccess || sync_streams(r) != hipSuccess) {
        fail(PT_ERR_DEVICE, "synchronising the renderer's streams failed");
        return false;
    }
    out->data = reinterpret_cast<float *>(r->d_sum);
    out->count = static_cast<size_t>(r->width) * r->height * 4;
    out->device = r->device;
    out->stream = r->stream;
    return true;
}

void main_stream_work_enqueued(pt_renderer *r) { mark_main_stream_work(r); }

int set_error(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

}  // namespace pt_internal

extern "C" {

const char *pt_last_error(void) { return g_last_error.c_str(); }
const char *pt_version(void) { return "pathtracer_amd 0.1 (gfx950)"; }

int pt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}
/* --------------------------------------------------------------- Renderer */

void pt_config_default(pt_config *cfg)
{
    if (!cfg) return;
    cfg->width = 1280;
    cfg->height = 720;
    cfg->max_depth = 10;
    cfg->fov_y_degrees = 19.5f;
    cfg->device = -1;
    cfg->use_bvh = 1;
    cfg->iterations_per_batch = 0;
    cfg->profile = 0;
    cfg->emulate_rgba8 = 0;
    cfg->bvh_builder = 2;
    cfg->batch_streams = 2;
}

int pt_create(const pt_scene_desc *scene, const pt_config *cfg_in, pt_renderer **out)
{
    PT_REQUIRE(scene && out, "pt_create: null argument");
    pt_config cfg;
    if (cfg_in)
        cfg = *cfg_in;
    else
        pt_config_default(&cfg);
    PT_REQUIRE(cfg.width >= 1 && cfg.height >= 1, "pt_create: empty resolution");
    PT_REQUIRE(cfg.max_depth >= 1 && cfg.max_depth <= 1024, "pt_create: max_depth out of range");
    PT_REQUIRE(scene->n_triangles == 0 || (scene->vertices && scene->indices && scene->vertex_data && scene->material_map),
               "pt_create: scene arrays missing");
    PT_REQUIRE(scene->n_triangles < (1u << 28), "pt_create: too many triangles");
    PT_REQUIRE(scene->n_lights == 0 || scene->lights, "pt_create: lights missing");
    PT_REQUIRE(scene->n_materials == 0 || scene->materials, "pt_create: materials missing");

    int ndev = 0;
    PT_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(PT_ERR_DEVICE, "no HIP device visible");
    int device = cfg.device;
    if (device < 0) PT_HIP(hipGetDevice(&device));
    if (device >= ndev) return fail(PT_ERR_INVALID, "pt_create: device ordinal out of range");
    PT_HIP(hipSetDevice(device));
    PT_HIP(noalloc_staging_init(device));  // the device is current: address of the no-alloc staging array on this GPU

    pt_renderer *r = new pt_renderer();
    r->cfg = cfg;
    r->device = device;
    if (const char *e = pt::tuning_env("PT_REGION_SIZE")) {  // tuning experiments only
        int v = std::atoi(e);
        if (v >= 64 && v <= 32768 && v % 64 == 0) r->region_size_env = static_cast<uint32_t>(v);
    }
    if (const char *e = pt::tuning_env("PT_SCENE_LDS")) r->scene_in_lds = std::atoi(e) != 0;  // tuning experiments only
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        r->n_cus = static_cast<uint32_t>(prop.multiProcessorCount);
    int rc = PT_OK;
    auto bail = [&](int code) {
        std::string keep = g_last_error;
        pt_destroy(r);
        g_last_error = keep;
        return code;
    };
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(PT_ERR_DEVICE, "hipStreamCreate failed"));
    r->two_lanes = cfg.batch_streams != 1u && pt::tuning_env("PT_ONE_LANE") == nullptr;  // (environment: experiments only)
    if (r->two_lanes) {
        bool ok = hipStreamCreateWithFlags(&r->stream2, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k) ok = hipEventCreateWithFlags(&r->ev_accum[k], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r->ev_ops, hipEventDisableTiming) == hipSuccess;
        if (!ok) return bail(fail(PT_ERR_DEVICE, "hipStreamCreate / hipEventCreate failed"));
    }
    if ((rc = upload_scene(r, scene)) != PT_OK) return bail(rc);
    if (hipMalloc(&r->d_counters, kCounters * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&r->d_params, 2 * sizeof(pt::BatchParams)) != hipSuccess)
        return bail(fail(PT_ERR_DEVICE, "hipMalloc failed"));
    if (cfg.profile & 8u) r->view.deep_stack_lds_entries = 4;  // tests: make deep stacks overflow early
    if (cfg.use_bvh && (cfg.profile & 32u) &&
        pt::extend_reads_tree_from_memory(r->view, r->stack_entries, r->n_nodes, r->scene_in_lds)) {
        // A tree that is read from memory, collapsed into four-wide nodes, one cache line each: half the
        // dependent node fetches per ray.  Opt-in (profile bit 5): measured on the 9.88 M-triangle scene it is
        // 8-12 % slower than the binary tree -- the same bytes per ray through the CU's vector-memory path, and
        // the ordering of four children costs more instructions than the shorter chain of fetches returns.
        pt::WideBvh wide;
        std::string why;
        const double t0 = now_ms();
        if (!pt::build_wide_nodes(r->d_nodes, r->n_nodes, r->bvh_depth, r->stream, &wide, &why)) return bail(fail(PT_ERR_DEVICE, why));
        r->d_wide_nodes = wide.d_nodes;
        r->view.wide_nodes = static_cast<const float4 *>(wide.d_nodes);
        r->view.n_wide_nodes = wide.n_nodes;
        r->view.wide_stack_entries = 3u * wide.depth + 1u;
        r->bvh_build_ms += now_ms() - t0;
    }
    if (cfg.use_bvh) {
        const size_t bytes = pt::extend_stack_overflow_bytes(r->n_cus, r->view, r->stack_entries, r->n_nodes, r->scene_in_lds);
        for (int k = 0; k < (r->two_lanes ? 2 : 1) && bytes; ++k)
            if (hipMalloc(&r->d_stack_overflow[k], bytes) != hipSuccess) return bail(fail(PT_ERR_DEVICE, "hipMalloc failed"));
    }
    if ((rc = alloc_paths(r, cfg.width, cfg.height)) != PT_OK) return bail(rc);
    r->cam.tan_half_fov = tan_half_fov(cfg.fov_y_degrees);

    // The pose main.cpp:126 hard-codes, until pt_set_camera says otherwise.
    pt::Camera def(pt::make3(0.f, 1.5f, 15.f), pt::make3(0.f, -0.25f, 0.f), cfg.width, cfg.height);
    def.update();
    for (int k = 0; k < 3; ++k) {
        r->cam.eye[k] = def.m_eye[k];
        r->cam.forward[k] = def.m_forward[k];
        r->cam.up[k] = def.m_up[k];
        r->cam.right[k] = def.m_right[k];
    }
    if ((rc = reset_accumulation(r)) != PT_OK) return bail(rc);
    if (sync_streams(r) != hipSuccess) return bail(fail(PT_ERR_DEVICE, "initial synchronise failed"));
    *out = r;
    return PT_OK;
}

void pt_destroy(pt_renderer *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)sync_streams(r);
    drain_events(r);
    free_paths(r);
    for (hipEvent_t e : {r->ev_accum[0], r->ev_accum[1], r->ev_ops})
        if (e) (void)hipEventDestroy(e);
    if (r->stream2) (void)hipStreamDestroy(r->stream2);
    void *bufs[] = {r->d_nodes, r->d_wide_nodes, r->d_tris, r->d_shade, r->d_materials, r->d_lights, r->d_counters, r->d_params,
                    r->d_stack_overflow[0], r->d_stack_overflow[1]};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

int pt_set_camera(pt_renderer *r, const float eye[3], const float forward[3], const float up[3], const float right[3])
{
    PT_REQUIRE(r && eye && forward && up && right, "pt_set_camera: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    for (int k = 0; k < 3; ++k) {
        r->cam.eye[k] = eye[k];
        r->cam.forward[k] = forward[k];
        r->cam.up[k] = up[k];
        r->cam.right[k] = right[k];
    }
    return reset_accumulation(r);
}

int pt_resize(pt_renderer *r, uint32_t width, uint32_t height)
{
    PT_REQUIRE(r && width >= 1 && height >= 1, "pt_resize: bad argument");
    PT_HIP(hipSetDevice(r->device));
    if (batch_iterations_for(r, width, height) == 0) return fail(PT_ERR_INVALID, "pt_resize: resolution too large");
    PT_HIP(sync_streams(r));
    const uint32_t old_w = r->cfg.width, old_h = r->cfg.height;
    const bool was_usable = r->usable;
    int rc = alloc_paths(r, width, height);
    if (rc != PT_OK) {
        // The old buffers are gone (both sets together may not fit beside each other, so the new ones are
        // not allocated first).  Go back to the size that worked; if even that fails the renderer refuses
        // everything but pt_resize / pt_destroy until a resize succeeds.
        const std::string why = g_last_error;
        (void)hipGetLastError();
        r->usable = was_usable && alloc_paths(r, old_w, old_h) == PT_OK && reset_accumulation(r) == PT_OK;
        if (!r->usable) {
            free_paths(r);
            r->width = r->height = 0;
            r->capacity = 0;
        }
        return fail(rc, "pt_resize: " + why + (r->usable ? " (the previous size was restored, accumulation reset)"
                                                         : " (the renderer has no image buffers now)"));
    }
    r->usable = true;
    r->cfg.width = width;
    r->cfg.height = height;
    return reset_accumulation(r);
}

int pt_reset(pt_renderer *r)
{
    PT_REQUIRE(r, "pt_reset: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    return reset_accumulation(r);
}

int pt_set_partition(pt_renderer *r, uint32_t rank, uint32_t world)
{
    PT_REQUIRE(r && world >= 1 && rank < world, "pt_set_partition: need rank < world");
    PT_USABLE(r);
    if (r->cfg.emulate_rgba8 && world != 1)
        return fail(PT_ERR_STATE, "pt_set_partition: the 8-bit running average cannot be partitioned");
    PT_HIP(hipSetDevice(r->device));
    r->rank = rank;
    r->world = world;
    return reset_accumulation(r);
}

int pt_render(pt_renderer *r, uint32_t n_iterations)
{
    PT_REQUIRE(r, "pt_render: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    uint32_t left = n_iterations;
    while (left > 0) {
        const uint32_t kb = std::min(left, r->batch_iterations);
        const uint64_t last = r->rank + (r->local_iterations + kb - 1) * static_cast<uint64_t>(r->world);
        if (last > 0xFFFFFFFEull) return fail(PT_ERR_STATE, "iteration counter would overflow 32 bits");
        int rc = render_batch(r, r->local_iterations, kb);
        if (rc != PT_OK) return rc;
        left -= kb;
    }
    return PT_OK;
}

int pt_synchronize(pt_renderer *r)
{
    PT_REQUIRE(r, "pt_synchronize: null argument");
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));
    return PT_OK;
}

int pt_set_exact_readbacks(pt_renderer *r, uint32_t enabled)
{
    PT_REQUIRE(r, "pt_set_exact_readbacks: null argument");
    r->exact_readbacks = enabled != 0;
    return PT_OK;
}

int pt_sum_device_ptr(pt_renderer *r, void **dev_ptr)
{
    PT_REQUIRE(r && dev_ptr, "pt_sum_device_ptr: null argument");
    PT_USABLE(r);
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));  // the caller reads the buffer on a stream of its own
    *dev_ptr = r->d_sum;
    return PT_OK;
}

int pt_get_stats(pt_renderer *r, pt_stats *s)
{
    PT_REQUIRE(r && s, "pt_get_stats: null argument");
    PT_HIP(hipSetDevice(r->device));
    PT_HIP(sync_streams(r));
    drain_events(r);
    unsigned long long c[kCounters] = {};
    PT_HIP(hipMemcpy(c, r->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    std::memset(s, 0, sizeof(*s));
    s->iterations = r->local_iterations;
    s->paths = r->paths;
    s->segments = c[0];
    s->nan_samples = c[1];
    s->nodes_visited = c[2];
    s->tris_tested = c[3];
    s->n_triangles = r->view.n_triangles;
    s->n_nodes = r->n_nodes;
    s->bvh_depth = r->bvh_depth;
    s->stack_entries = r->stack_entries;
    s->bvh_build_ms = r->bvh_build_ms;
    s->upload_ms = r->upload_ms;
    s->ms_raygen = r->ms[0];
    s->ms_extend = r->ms[1];
    s->ms_shade = r->ms[2];
    s->ms_accumulate = r->ms[3];
    s->launches_raygen = r->launches[0];
    s->launches_extend = r->launches[1];
    s->launches_shade = r->launches[2];
    s->launches_accumulate = r->launches[3];
    s->ms_bounces = r->ms[4];
    s->launches_bounces = r->launches[4];
    s->sah_cost = r->sah_cost;
    for (int k = 0; k < 8; ++k) s->extend_diag[k] = c[4 + k];
    s->hits_shaded = c[12];
    s->shade_batches = c[13];
    s->bvh_builder = r->bvh_builder;
    s->iterations_per_batch = r->batch_iterations;
    s->primary_rays_culled = c[14];
    s->unlit_last_rays_skipped = c[15];
    s->segments += c[15];  // (they are segments of their paths all the same)
    return PT_OK;
}

}  // extern "C"
