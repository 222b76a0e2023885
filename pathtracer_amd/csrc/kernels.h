// Launch wrappers of the gfx950 wavefront-integrator kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.h"

namespace pt {

// Everything the kernels read about the scene; all pointers are device memory.
struct SceneView {
    const float4 *nodes;       // 4 lanes per BvhNode
    const float4 *triangles;   // 4 lanes per BvhTriangle, leaf order (or scan order)
    const float4 *shade;       // 5 lanes per ShadeTriangle, original order
    const float4 *materials;   // 2 lanes per material
    const DeviceLight *lights;
    uint32_t n_triangles;
    uint32_t n_materials;
    uint32_t n_lights;
    float scene_radius;        // largest |coordinate| of the mesh bounds
    // Entries of a deep traversal stack (a tree read from memory) kept in LDS; 0 = the default (16).  Tests lower
    // it so that the overflow buffer is actually used (pt_config.profile bit 3).
    uint32_t deep_stack_lds_entries = 0;
    // The tree collapsed into four-wide nodes (bvh_device.h, WideBvh), made for trees that are read from memory;
    // null otherwise.  wide_stack_entries: stack entries a ray can need in it (3 per level + the end marker).
    const float4 *wide_nodes = nullptr;
    uint32_t n_wide_nodes = 0;
    uint32_t wide_stack_entries = 0;
};

// One path queue: three 16-byte lanes per path.
//   p0 = (origin.xyz, dir.x)  p1 = (dir.y, dir.z, seed.x, seed.y)  p2 = (throughput.rgb, sample slot)
// lt: with exactly one light in the scene, the distance at which the ray meets it (frag:363-382), computed by
// the kernel that creates the ray; a quiet NaN when it does not (and always with another number of lights).  The traversal then starts with that distance as its
// limit instead of testing the light when the ray finishes.
struct PathQueue {
    float4 *p0;
    float4 *p1;
    float4 *p2;
    float *lt;
};

// The path queues are cut into regions of `size` consecutive slots.  A
// region belongs to exactly one wave for the duration of a kernel, which
// compacts the survivors of region r into region r of the other queue and
// writes the region's new live count: no atomics and no ordering between
// waves are needed, and the queue order is the same in every run.
// counts_in/out are n_regions entries for one bounce.  Wave w handles the
// regions perm(w), perm(w + n_waves), ... with perm(k) = k * stride mod
// n_regions (stride coprime to n_regions), which spreads image areas with much
// and little work evenly over the waves.
struct Regions {
    const uint32_t *counts_in;
    uint32_t *counts_out;  // may be null (extend does not change counts)
    uint32_t n_regions;
    uint32_t stride;
    uint32_t size;         // slots per region (a multiple of 64, at most 32768)
};

struct TraversalCounters {
    unsigned long long *nodes_visited;  // may be null
    unsigned long long *tris_tested;
    // Scheduling diagnostics of k_extend_bvh, 8 words, may be null:
    // [0] phase-A loop iterations (per wave) [1] lanes active in them
    // [2] phase-B triangle-loop iterations   [3] lane-triangle tests in them
    // [4] rounds (A + B)                     [5] lanes holding a ray per round
    // [6] sum over waves of their busy time  [7] longest busy time of one wave (100 MHz ticks)
    unsigned long long *diag;
};

// Grid used by every kernel: persistent-style, grid-stride over the live count.
struct LaunchShape {
    uint32_t blocks;      // grid for the grid-stride kernels
    uint32_t n_cus;       // compute units of the device
    uint32_t wave_items;  // work available to the region kernels, in waves' worth (regions)
};

// What changes from one batch to the next (and with the camera), kept in a small block of device memory that
// the kernels of a batch read, so that their arguments stay the same from batch to batch.  launch_set_batch writes
// the block in stream order.
struct BatchParams {
    DeviceCamera cam;
    uint32_t it_first;    // iterationCount of the batch's first sample (renderer.cpp:132)
    uint32_t sample_tag;  // written into the w lane of every sample of this batch (see launch_accumulate)
    uint32_t pad[2];
};
void launch_set_batch(hipStream_t s, BatchParams *params, const DeviceCamera &cam, uint32_t it_first, uint32_t sample_tag);

// Readbacks: every pixel's ray of one iteration at the slot the slot -> pixel mapping gives it (lanes p0, p1 and
// the light distance; no path state).  counts_out: n_regions live counts.
void launch_raygen(hipStream_t s, LaunchShape g, const SceneView &scene, const BatchParams *params, uint32_t it_stride,
                   uint32_t n_iterations, PathQueue out, uint32_t *counts_out, uint32_t region_size);

// The render loop's primary rays, region by region, all three lanes.  cull: rays that meet neither the scene's
// bounds (the traversal's own first step) nor a light are not written -- they could only add zero -- and the
// rest is packed to the front of its region; regions.counts_out receives the rays written per region,
// culled_out the rays left out.
void launch_raygen_regions(hipStream_t s, LaunchShape g, const SceneView &scene, const BatchParams *params, uint32_t it_stride,
                           uint32_t n_iterations, PathQueue out, Regions regions, uint32_t *culled_out, bool cull);

// Where extend puts its results.
//  render = false: hit_list is written as a float2 array, results[slot] = (t, hit code), for every ray
//                  (readbacks); exact_parallel also reports the reference's hits at t = +inf on triangles
//                  whose plane the ray is exactly parallel to.
//  render = true : rays that reach a light write samples[path slot]; rays that escape (or hit a surface on
//                  the last allowed segment) add nothing; other surface hits are appended per region to
//                  hit_list as (t, triangle, slot in region, -) and regions.counts_out receives their number.
struct ExtendOutput {
    bool light_distance_in_queue;  // q.lt is valid (rays made by raygen / shade in a one-light scene)
    float4 *hit_list;
    float4 *samples;
    bool render;
    bool last_segment;
    bool exact_parallel;
    const BatchParams *params = nullptr;  // the batch's block (render = true): sample tag, camera
    // Where the traversal kernel keeps the deep ends of its stacks when the tree is read from memory
    // (extend_stack_overflow_bytes); null: whole stacks in LDS.  One buffer per launch in flight.
    int *stack_overflow = nullptr;
};

// stack_entries: entries of the per-lane traversal stack in LDS (>= BVH depth).
void launch_extend(hipStream_t s, LaunchShape g, const SceneView &scene, uint32_t stack_entries, bool use_bvh,
                   Regions regions, PathQueue q, const ExtendOutput &out, TraversalCounters tc, uint32_t n_nodes,
                   bool scene_in_lds);

// regions.counts_in = hits per region, regions.counts_out = surviving paths per region.
// Waves of the traversal kernel that are resident at once for this scene (the occupancy the LDS-resident part
// of the scene leaves): the host sizes the regions so that each of them gets the same number of regions.
uint32_t extend_resident_waves(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes,
                               bool scene_in_lds, bool overflow_buffer);

// Whether the traversal kernel would read this scene's tree from memory (it does not fit in LDS): such a scene
// gets four-wide nodes (SceneView::wide_nodes) before anything else is planned for it.
bool extend_reads_tree_from_memory(const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds);

// Bytes of ExtendOutput::stack_overflow this scene's launches want (0: the stacks fit in LDS as they are).
size_t extend_stack_overflow_bytes(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes,
                                   bool scene_in_lds);

// A whole batch in one launch (k_bounces): what launch_raygen and then launch_extend / launch_shade for bounces
// 0 .. depth-1 of the render loop would do (no counters, no exact-samples pass), with the same results.
struct BounceLoop {
    PathQueue q[2];    // bounce b reads q[b & 1] and writes q[(b + 1) & 1]
    float4 *hit_list;
    float4 *samples;
    uint32_t *counts;  // rows of `pitch` entries: 0..depth rays per region and bounce, depth+1..2*depth hits per region and
                       // bounce, 2*depth+1 primary rays per region that ray generation did not enqueue
    uint32_t pitch, n_regions, stride, size, depth;
    // counter of the rays of the last allowed segment that were not made because they do not point at the scene's one
    // light (launch_shade, unlit_skipped)
    unsigned long long *unlit_skipped;
};
// Waves of k_bounces resident at once for this scene (as extend_resident_waves for the traversal kernel).
uint32_t bounces_resident_waves(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds,
                                bool overflow_buffer);
// Whether the one launch also serves large batches best for this scene (the whole scene in LDS; measured: Box and
// the Anisotropic scenes, in small blocks, +3-6 %; the Transmission scenes, one large block per CU, +1.5 %; Veggie,
// one large block with most triangles in memory, -9 %; trees in memory -5 %).
bool bounces_suit_any_batch(const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds, bool overflow_buffer);
void launch_bounces(hipStream_t s, LaunchShape g, const SceneView &scene, uint32_t stack_entries, const BounceLoop &loop,
                    const BatchParams *params, uint32_t n_nodes, bool scene_in_lds, int *stack_overflow, uint32_t it_stride,
                    uint32_t n_iterations);

void launch_shade(hipStream_t s, LaunchShape g, const SceneView &scene, Regions regions, PathQueue in,
                  const float4 *hit_list, PathQueue out, float4 *samples /* receives the sample of a path whose ray went non-finite */,
                  const BatchParams *params,
                  bool final_bounce /* the hits of the last allowed segment: nothing survives, only non-finite samples are written */,
                  unsigned long long *unlit_skipped = nullptr /* not null: the rays made here travel the last allowed segment; in a
                  one-light scene those that do not point at the light are not written (they could add nothing) but counted here */);

// The sample buffer is never cleared: a path that ends on a light writes (radiance, tag) with the tag of its
// batch, and a slot whose w lane does not carry this batch's tag counts as a zero sample (a path that escaped).
void launch_accumulate(hipStream_t s, LaunchShape g, const float4 *samples, float4 *sum, uint32_t n_pixels,
                       uint32_t n_iterations, const uint32_t *counts, uint32_t n_rows, uint32_t n_regions,
                       uint32_t pitch, unsigned long long *segments_total,
                       unsigned long long *shade_totals /* [0] surface hits shaded, [1] 64-lane batches formed */,
                       unsigned long long *nan_counter, unsigned long long *culled_total /* primary rays not enqueued (counts row 2*n_rows+1) */,
                       uint8_t *rgba8 /* may be null */, const BatchParams *params, uint32_t it_stride,
                       bool culled /* the batch's ray generation left out the pixels outside the camera's window: their samples are not read */);

// hits -> int32 hit codes / t / directions in pixel order for one iteration.
// cam: the camera whose raygen made the rays (its slot -> pixel mapping orders the output by pixel); null: rays
// were packed by the caller, output in slot order.
void launch_unpack_primary(hipStream_t s, LaunchShape g, uint32_t n, PathQueue q, const float2 *results, int32_t *ids,
                           float *t, float *dirs, const DeviceCamera *cam);

// Intersection::normal of the primary hits whose (t, code) results lie in `hits` (3 floats per pixel).
void launch_primary_normals(hipStream_t s, LaunchShape g, uint32_t n, const SceneView &scene, PathQueue q, const float2 *hits,
                            float *normals, const DeviceCamera &cam);

void launch_pack_rays(hipStream_t s, LaunchShape g, uint32_t n, const float *origins, const float *dirs, PathQueue q,
                      uint32_t *counts_out, uint32_t region_size);

void launch_tonemap(hipStream_t s, LaunchShape g, const float4 *sum, uint8_t *rgb, uint32_t width, uint32_t height);

void launch_sample_surface(hipStream_t s, LaunchShape g, uint32_t n, const float *materials, const float *normals,
                           const float *wo, const float *xi, const uint32_t *seeds, float *wi_out, float *pdf_out,
                           float *f_out, uint32_t *flags_out, uint32_t *seeds_out);

}  // namespace pt

