// The renderer's host-side state and the helpers its translation units share (pt_api.cpp: lifetime, scene upload and
// the per-batch launch sequence; pt_readback.cpp: read-backs, checkpoints and the unit-test entry points;
// pt_objects.cpp: the scene, camera and light objects).  Internal: nothing here is part of the C ABI.
#pragma once
#include "../../include/pathtracer_amd.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "camera.h"
#include "device_types.h"
#include "kernels.h"
#include "renderer_internal.h"
#include "scene.h"

static inline int fail(int code, const std::string &msg) { return pt_internal::set_error(code, msg); }

#define PT_HIP(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" +      \
                                           __FILE__ + ":" + std::to_string(__LINE__) + ")");            \
    } while (0)

#define PT_REQUIRE(cond, msg) \
    do {                      \
        if (!(cond)) return fail(PT_ERR_INVALID, msg); \
    } while (0)

// Entry points that touch the path buffers refuse a renderer whose last pt_resize left it without any.
#define PT_USABLE(r)                                                                                              \
    do {                                                                                                          \
        if (!(r)->usable)                                                                                         \
            return fail(PT_ERR_STATE, "the renderer has no image buffers (a pt_resize failed); call pt_resize again"); \
    } while (0)

struct pt_scene {
    pt::Scene scene;
};

struct pt_camera {
    pt::Camera cam;
    explicit pt_camera(const pt::Camera &c) : cam(c) {}
};

constexpr size_t kCounters = 16;  // device counters behind pt_stats (pt_renderer::d_counters)

struct pt_renderer {
    pt_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n_cus = 256;

    // scene on the device
    pt::SceneView view{};
    void *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_materials = nullptr, *d_lights = nullptr;
    void *d_wide_nodes = nullptr;  // the tree as four-wide nodes, for trees that are read from memory (SceneView::wide_nodes)
    uint32_t n_nodes = 0, bvh_depth = 0, stack_entries = 16;
    double sah_cost = 0, bvh_build_ms = 0, upload_ms = 0;

    // camera
    pt::DeviceCamera cam{};

    // path state
    bool usable = true;  // false after a pt_resize that could neither get the new buffers nor restore the old size
    uint32_t width = 0, height = 0, batch_iterations = 1;
    size_t capacity = 0;
    pt::PathQueue q[2] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    float4 *d_hits = nullptr;                 // hit list of the render loop; (t, code) results of the readbacks
    float4 *d_samples = nullptr;
    uint32_t sample_tag = 0;                  // batch counter; tags the samples written in the current batch
    float4 *d_sum = nullptr;
    uint32_t *d_counts = nullptr;             // (max_depth + 1) rows of region_pitch live counts, one row per bounce
    uint32_t region_pitch = 0;                // regions the queues hold at full capacity
    uint32_t region_size = 1024;              // slots per region, chosen from the batch size and the resident waves
    uint32_t region_size_env = 0;             // set through the environment (experiments)
    uint32_t extend_waves = 0;                // waves of the traversal kernel resident at once
    uint32_t bounces_waves = 0;               // the same for the kernel that runs all bounces of a batch in one launch
    bool bounces_any_size = false;            // that kernel serves this scene best at every batch size (kernels.h)
    uint64_t scene_hash = 0;                  // FNV-1a over the scene arrays (checkpoint compatibility)
    uint8_t *d_rgba8 = nullptr;               // emulated 8-bit accumulation texture (cfg.emulate_rgba8)
    uint32_t bvh_builder = 0;                 // the builder that made the tree (0 host, 1 device)

    // Second set of batch buffers with a stream of its own.  Consecutive batches alternate between the
    // two sets, so that the kernels of two batches are in flight together: the traversal kernel is bound by
    // VALU issue and the shade / raygen / accumulate kernels by HBM, and a kernel's thinning tail is filled
    // by the other batch's work.  Only the accumulate kernels are ordered against each other (events), in
    // batch order, so the image is the one a single stream produces.
    bool two_lanes = false;
    hipStream_t stream2 = nullptr;
    pt::PathQueue q2[2] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    float4 *d_hits2 = nullptr, *d_samples2 = nullptr;
    uint32_t *d_counts2 = nullptr;
    uint32_t sample_tag2 = 0;
    hipEvent_t ev_accum[2] = {nullptr, nullptr};  // recorded after a set's accumulate kernel
    bool ev_accum_set[2] = {false, false};
    hipEvent_t ev_ops = nullptr;                  // recorded on `stream` after work the second stream must wait for
    bool ops_pending = false;
    uint32_t next_lane = 0;
    uint32_t lanes_in_use = 1;                    // 2 unless the second set did not fit in memory
    bool scene_in_lds = true;                 // extend keeps nodes (and triangles) in LDS when they fit
    unsigned long long *d_counters = nullptr; // kCounters words: [0] segments [1] nan samples [2] nodes visited [3] tris tested [4..11] extend diagnostics [12] hits shaded [13] shade batches [14] primary rays culled at generation [15] rays of the last segment not made because they miss the light

    // pt_read_hit_ids / pt_trace_rays report hits at t = +inf on exactly
    // parallel triangles like the reference does (see k_extend_bvh).
    bool exact_readbacks = true;

    // progress
    uint32_t rank = 0, world = 1;
    uint64_t local_iterations = 0;  // samples per pixel rendered by this renderer since reset
    uint64_t paths = 0;

    // profile
    double ms[5] = {0, 0, 0, 0, 0};
    uint64_t launches[5] = {0, 0, 0, 0, 0};
    struct TimedLaunch {
        hipEvent_t begin, end;
        int kernel;  // 0 raygen, 1 extend, 2 shade, 3 accumulate, 4 all bounces in one launch
    };
    std::vector<TimedLaunch> timed;

    // per-batch scalars on the device, one block per set of batch buffers (kernels.h, BatchParams)
    pt::BatchParams *d_params = nullptr;

    // World-space points whose projections bound everything a primary ray can meet: the corners of the mesh's
    // bounding box and of every light's rectangle (update_cull_window).
    std::vector<std::array<double, 3>> cull_points;

    // the deep ends of the traversal kernel's stacks for a tree that is read from memory (kernels.h,
    // ExtendOutput::stack_overflow): one buffer per stream, null when the stacks fit in LDS
    int *d_stack_overflow[2] = {nullptr, nullptr};
};

namespace pt_internal {

double now_ms();
// Waits for everything the renderer has enqueued, on both of its streams.
hipError_t sync_streams(pt_renderer *r);
// Slots per region, region bookkeeping and grid for a launch over n_items queue slots (pt_api.cpp).
uint32_t region_size_for(const pt_renderer *r, size_t n_items);
pt::Regions regions_for(const pt_renderer *r, size_t n_items, uint32_t row_in, int row_out, uint32_t *counts = nullptr);
pt::LaunchShape shape_for(const pt_renderer *r, size_t work_items);
// Clears the sum, the counters and the progress of the renderer (every change of camera, size or lights).
int reset_accumulation(pt_renderer *r);
// Looks up the address of the staging array of pt_read_tonemapped_noalloc on `device`, which must be the current
// device; once per device (pt_readback.cpp).
hipError_t noalloc_staging_init(int device);

}  // namespace pt_internal
