/*
 * pathtracer_amd.h -- C ABI of the MI355X-native path-tracing integrator.
 *
 * The reference (kbau121/PathTracer) has no FFI layer; the boundary this
 * library stands behind is `class Renderer` (src/renderer.h:12-39) fed by the
 * flat arrays of `class Scene` (src/scene.h:53-62) and the axes of `class
 * Camera` (src/camera.h:12-19).  Every entry point below names the reference
 * interface it replaces.  All functions return PT_OK (0) or a negative error
 * code and never throw; pt_last_error() describes the most recent failure on
 * the calling thread.  A renderer is driven by one host thread at a time.
 *
 * Image convention: pixel index = y * width + x with row 0 the BOTTOM row, as
 * gl_FragCoord numbers them (pathtracer.frag.glsl:310,866).
 */
#ifndef PATHTRACER_AMD_H
#define PATHTRACER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID -1   /* bad argument                         */
#define PT_ERR_IO -2        /* file could not be read               */
#define PT_ERR_DEVICE -3    /* HIP call failed / no usable GPU      */
#define PT_ERR_STATE -4     /* call not valid in the current state  */

/* Hit codes written by pt_read_hit_ids: a triangle index >= 0 (position in
 * Scene::m_indices), PT_HIT_MISS, or PT_HIT_LIGHT(k) for light k. */
#define PT_HIT_MISS (-1)
#define PT_HIT_LIGHT(k) (-(2 + (k)))

typedef struct pt_scene pt_scene;
typedef struct pt_camera pt_camera;
typedef struct pt_renderer pt_renderer;

/* The six device-bound arrays of Scene exactly as the reference lays them out
 * and uploads them (scene.h:53-62, renderer.cpp:32-83), so a host Scene can be
 * handed over without conversion.  Pointers are host memory, read during the
 * call only and never retained. */
typedef struct {
    const float *vertices;        /* n_vertices x 3                       m_vertices    */
    uint32_t n_vertices;
    const float *indices;         /* n_triangles x 4 floats (i0,i1,i2,w)  m_indices     */
    uint32_t n_triangles;
    const float *vertex_data;     /* 3*n_triangles x 6 (normal, uv0)      m_vertexData  */
    const float *lights;          /* n_lights x 20 (radiance4, mat4)      m_lights      */
    uint32_t n_lights;            /*                                      m_lightCount[0] */
    const float *materials;       /* n_materials x 8                      m_materials   */
    uint32_t n_materials;
    const uint32_t *material_map; /* n_triangles                          m_materialMap */
} pt_scene_desc;

/* Values the reference hard-codes, as parameters with the reference's values
 * as defaults (pt_config_default). */
typedef struct {
    uint32_t width, height;        /* Camera::m_resolution, main.cpp:126: 1280 x 720   */
    uint32_t max_depth;            /* pathtracer.frag.glsl:874: 10                      */
    float fov_y_degrees;           /* pathtracer.frag.glsl:306: 19.5                    */
    int32_t device;                /* HIP device ordinal; -1 = current device           */
    uint32_t use_bvh;              /* 1 = SAH BVH traversal; 0 = linear scan in the
                                      reference's triangle order (frag:391-416)         */
    uint32_t iterations_per_batch; /* samples per pixel in flight at once; 0 = auto (about 64 M or 128 M paths,
                                      see pt_stats.iterations_per_batch) */
    uint32_t profile;              /* bit 0: time every launch with HIP events; bit 1:
                                      count BVH nodes / triangle tests (both slower);
                                      bit 2: produce the reference's non-finite samples one
                                      for one, so that pt_stats.nan_samples equals what the
                                      shader would make: the BSDF is evaluated at the surface
                                      a path's last allowed segment ends on (frag:884-895),
                                      and an escaping ray exactly parallel to some triangle's
                                      plane counts as the hit at t = inf it is there
                                      (frag:404-411).  Such samples add zero to the image
                                      either way; the check costs a scan over all triangles
                                      per escaping ray;
                                      bit 3 (for tests): a tree too large for on-chip memory
                                      is traversed with only 4 entries of each ray's stack
                                      on chip, so that the overflow buffer, which otherwise
                                      holds entries 17 and up, is used by most rays;
                                      bit 4: one launch per kernel and bounce even where all
                                      bounces of a batch would go in one launch (per-kernel
                                      timings with bit 0; the image is the same);
                                      bit 5: a tree too large for on-chip memory is traversed
                                      as four-wide nodes (two levels of the binary tree per
                                      128-byte record): half the dependent node fetches per
                                      ray, same hits; measured slower than the binary tree on
                                      ten million triangles, kept for measurement             */
    uint32_t emulate_rgba8;        /* 1: beside the float sum, keep the reference's own
                                      accumulator: an 8-bit RGBA texel per pixel updated as
                                      (accum*it + sample)/(it+1) and re-quantised every
                                      iteration (frag:900-904, renderer.cpp:21,25), NaN
                                      samples and all.  Single renderer only (the running
                                      average is not a sum and cannot be partitioned).   */
    uint32_t bvh_builder;          /* 0: the tree is built on the host (threads) and uploaded;
                                      1: it is built on the GPU from the uploaded triangles;
                                      2 (default): on the GPU from 65536 triangles up, on
                                      the host below that and for input the device builder
                                      declines.  Both make the same tree; the device build
                                      takes milliseconds where the host takes seconds on ten
                                      million triangles.  With 1, pt_create fails on input
                                      that needs the host builder's object-median fall-back
                                      (see pt_last_error).                                  */
    uint32_t batch_streams;        /* 0 or 2 (default): consecutive batches alternate between two
                                      sets of path buffers on two HIP streams, so that kernels of
                                      two batches run side by side (the traversal kernel is bound
                                      by VALU issue, the others by HBM; +18 % paths/s); the
                                      accumulate kernels are ordered by events, the image is the
                                      same.  1: one set, one stream, kernels strictly in sequence
                                      (half the queue memory; per-kernel timings in isolation).  */
} pt_config;

typedef struct {
    uint64_t iterations;      /* samples per pixel rendered by THIS renderer since reset */
    uint64_t paths;           /* pixel-samples traced                                    */
    uint64_t segments;        /* rays cast (extend work items)                           */
    uint64_t nan_samples;     /* non-finite samples replaced by 0                        */
    uint64_t nodes_visited;   /* BVH inner nodes fetched   (profile bit 1)               */
    uint64_t tris_tested;     /* triangle tests            (profile bit 1)               */
    uint32_t n_triangles, n_nodes, bvh_depth, stack_entries;
    double bvh_build_ms, upload_ms;
    /* Sums of HIP-event durations per kernel and launch counts (profile bit 0). */
    double ms_raygen, ms_extend, ms_shade, ms_accumulate;
    uint64_t launches_raygen, launches_extend, launches_shade, launches_accumulate;
    double sah_cost;
    /* Scheduling diagnostics of the extend kernel (profile bit 1), per wave:
     * inner-node loop iterations, lanes active in them, triangle-loop
     * iterations, lane-triangle tests, rounds, lanes holding a ray per round,
     * sum over waves of their busy time and the longest single wave (100 MHz ticks). */
    uint64_t extend_diag[8];
    /* Lane occupancy of the shade kernel after compaction: surface hits shaded and
     * the 64-lane batches they were processed in (fill = hits / (64 * batches)). */
    uint64_t hits_shaded, shade_batches;
    uint32_t bvh_builder;     /* the builder that made the tree: 0 host, 1 device */
    uint32_t iterations_per_batch; /* samples per pixel that share a batch (pt_config.iterations_per_batch, or what
                                    * the renderer chose for 0 = auto at this image size and for this scene) */
    /* Batches whose bounces all ran in one launch (small batches, and every batch of a scene that lives on
     * chip with all its triangles): duration and count of those launches (profile bit 0); ms_extend / ms_shade then
     * cover only the batches that were launched kernel by kernel. */
    double ms_bounces;
    uint64_t launches_bounces;
    /* Primary rays that ray generation found to meet neither the scene's bounds nor a light and therefore
     * never enqueued (they can only add zero, frag:876).  They are counted in `segments` -- every path has a
     * first segment -- but no traversal kernel ever reads them. */
    uint64_t primary_rays_culled;
    /* Rays of the paths' last allowed segment that were never made because they do not point at the scene's
     * one light: on that segment a surface hit is not shaded and an escape is black (frag:874-897), so only
     * reaching the light can add anything.  Counted in `segments` like the culled primary rays.  Zero with any
     * other number of lights and with profile bit 2 / emulate_rgba8 (which shade the last surface too). */
    uint64_t unlit_last_rays_skipped;
} pt_stats;

const char *pt_last_error(void);
const char *pt_version(void);
/* Number of visible HIP devices; negative error code when the runtime fails. */
int pt_device_count(void);

/* ------------------------------------------------------------------ Scene
 * Replaces Scene::Scene(objFilename, mtlRoot), Scene::Scene() and the public
 * vectors (scene.h:64-218). */
int pt_scene_load_obj(const char *obj_path, const char *mtl_root, pt_scene **out);
int pt_scene_create_default(pt_scene **out);                      /* scene.h:64-76 */
int pt_scene_from_desc(const pt_scene_desc *desc, pt_scene **out);
void pt_scene_destroy(pt_scene *scene);
/* Fills desc with pointers into the scene's own arrays (valid until the scene
 * is modified or destroyed). */
int pt_scene_get_desc(const pt_scene *scene, pt_scene_desc *desc);
/* Scene::Light(radiance, position, rotation, scale) (scene.h:32-34): writes the
 * 20 floats of one light record. */
int pt_light_make(const float radiance[3], const float position[3], const float rotation[3],
                  const float scale[3], float out_light[20]);
/* Replaces m_lights / m_lightCount (main.cpp:122-123 assigns them directly). */
int pt_scene_set_lights(pt_scene *scene, const float *lights, uint32_t n_lights);
/* The light main.cpp:122 installs. */
int pt_scene_set_default_light(pt_scene *scene);
/* Deterministic stress scene: shapes after the first room_triangles triangles
 * instanced on an n^3 lattice (see Scene::instanceLattice). */
int pt_scene_instance_lattice(const pt_scene *base, uint32_t room_triangles, uint32_t n, pt_scene **out);

/* ----------------------------------------------------------------- Camera
 * Replaces class Camera (camera.h:22-33, camera.cpp). */
int pt_camera_create(const float eye[3], const float focus[3], uint32_t width, uint32_t height,
                     pt_camera **out);
void pt_camera_destroy(pt_camera *cam);
int pt_camera_look_at(pt_camera *cam, const float focus[3], const float *eye_or_null); /* camera.cpp:12-22 */
int pt_camera_move(pt_camera *cam, const float offset[3]);                             /* camera.cpp:25-29 */
int pt_camera_pan(pt_camera *cam, float dx, float dy);                                 /* camera.cpp:32-35 */
int pt_camera_orbit(pt_camera *cam, float dx, float dy);                               /* camera.cpp:38-76 */
int pt_camera_zoom(pt_camera *cam, float scale);                                       /* camera.cpp:79-86 */
int pt_camera_update(pt_camera *cam);                                                  /* camera.cpp:89-95 */
/* out = eye, focus, forward, up, right (15 floats); res = width, height. */
int pt_camera_get(const pt_camera *cam, float out[15], uint32_t res[2]);

/* --------------------------------------------------------------- Renderer */
void pt_config_default(pt_config *cfg);
/* Renderer::Renderer (renderer.cpp:11-112): copies the scene to the device
 * (building the BVH and the SoA records on the way).  The scene is immutable
 * afterwards; nothing in desc is retained. */
int pt_create(const pt_scene_desc *scene, const pt_config *cfg, pt_renderer **out);
void pt_destroy(pt_renderer *r);                                   /* renderer.cpp:114-122 */
/* Renderer::updateCamera (renderer.cpp:189-202) with the axes Camera::update
 * produced; resets the accumulation. */
int pt_set_camera(pt_renderer *r, const float eye[3], const float forward[3], const float up[3],
                  const float right[3]);
int pt_resize(pt_renderer *r, uint32_t width, uint32_t height);    /* renderer.cpp:160-187 */
int pt_reset(pt_renderer *r);                                      /* renderer.cpp:152-158 */
/* n calls of Renderer::draw (renderer.cpp:124-150): n more samples per pixel.
 * Sample k uses iterationCount = first + k * stride of the partition set by
 * pt_set_partition (default first 0, stride 1).  Asynchronous. */
int pt_render(pt_renderer *r, uint32_t n_iterations);
/* Sample-space partition for one-process-per-GPU runs: this renderer renders
 * iterations rank, rank + world, rank + 2*world, ...  Resets the accumulation. */
int pt_set_partition(pt_renderer *r, uint32_t rank, uint32_t world);
int pt_synchronize(pt_renderer *r);

/* Checkpoint / resume of a progressive render.  The reference's render state is its accumulation texture and
 * iteration counter (renderer.h:21-29, renderer.cpp:132-133); here: the float sum buffer, the number of
 * iterations this renderer has done, and the partition they belong to.  The file also records the image size,
 * maximum depth, camera basis and a checksum of the scene arrays; pt_load_state refuses a file that does not
 * match the renderer it is loaded into (PT_ERR_STATE).  After a load, pt_render continues with the next
 * iteration index, and the final sums equal those of an uninterrupted render bit for bit. */
int pt_save_state(pt_renderer *r, const char *path);
int pt_load_state(pt_renderer *r, const char *path);

/* Readbacks (the reference shows the image in a window; none of these has a
 * counterpart there).  All synchronise first. */
/* Sum of samples: width*height*4 floats, rgb = radiance sum, a = sample count. */
int pt_read_sum(pt_renderer *r, float *rgba_host);
/* Mean radiance (sum / count): the float equivalent of the accumulation texture. */
int pt_read_radiance(pt_renderer *r, float *rgba_host);
/* post.frag:11-15 (Reinhard then gamma 1/2.2) on the mean, as 8-bit RGB,
 * row 0 = TOP row (image-file order). */
int pt_read_tonemapped(pt_renderer *r, uint8_t *rgb_host);
/* pt_read_tonemapped for callers that cannot allocate device memory (tight preview loops, harnesses that
 * forbid hipMalloc / hipFree between frames; not for use inside a HIP graph capture: like every readback it
 * synchronises the renderer's streams and copies to pageable host memory): the same kernel, the same bytes, but the 8-bit image is
 * staged in a device array that is part of the library (one copy per GPU, 10.5 MiB, held until the process
 * ends) instead of a buffer allocated and freed per call.  No counterpart in the reference.  Images of more
 * than 2560 x 1440 = 3 686 400 pixels are refused with PT_ERR_INVALID.  All calls on one GPU share that
 * GPU's array: calls that overlap in time on the same GPU (two renderers driven from two threads) race. */
int pt_read_tonemapped_noalloc(pt_renderer *r, uint8_t *rgb_host);
/* The emulated 8-bit accumulation texture (pt_config.emulate_rgba8): width*height*4 bytes, row 0 = bottom
 * row like the texture the reference's shader reads back (frag:900), before post.frag. */
int pt_read_rgba8(pt_renderer *r, uint8_t *rgba_host);
/* Primary-ray hit codes of iteration `iteration` for every pixel. t_host and
 * dirs_host (3 floats per pixel) may be NULL. Does not touch the accumulation. */
int pt_read_hit_ids(pt_renderer *r, uint32_t iteration, int32_t *ids_host, float *t_host, float *dirs_host);
/* Intersection::normal of the same primary hits (frag:434-460), 3 floats per pixel: the corner normals
 * weighted by the barycentric coordinates of the hit point, not renormalised (frag:450); on a light the
 * light's transform * (0,0,1,0); zero on a miss. */
int pt_read_hit_normals(pt_renderer *r, uint32_t iteration, float *normals_host);
/* Closest hit of n caller-supplied rays (origins/dirs: 3 floats each, host
 * memory) through the same extend kernel the renderer uses. */
int pt_trace_rays(pt_renderer *r, uint64_t n, const float *origins, const float *dirs, float *t_host,
                  int32_t *ids_host);
/* pt_read_hit_ids and pt_trace_rays normally add one pass to the BVH traversal:
 * a ray that misses everything but is EXACTLY parallel (dot(n,d) == 0 in f32)
 * to some triangle's plane is reported as hitting that triangle at t = +inf,
 * because the reference's scan accepts it (pathtracer.frag.glsl:404-411 compare
 * false on inf).  The render loop does not run that pass: such a hit can only
 * produce a non-finite, dropped sample, i.e. what a miss contributes.
 * enabled = 0 makes the readbacks use the render loop's kernel verbatim. */
int pt_set_exact_readbacks(pt_renderer *r, uint32_t enabled);
/* One sampleSurface evaluation per record on the GPU (pathtracer.frag.glsl:
 * 841-858) for testing the BSDF stage in isolation.  Per record: material 8
 * floats, normal 3, wo 3, xi 2, seed 2 (uint32).  Outputs: wi 3, pdf 1, f 3,
 * flags 1 (bit 0 = the lobe returned without a direction), seed_out 2. */
int pt_sample_surface(pt_renderer *r, uint64_t n, const float *materials, const float *normals,
                      const float *wo, const float *xi, const uint32_t *seeds, float *wi_out, float *pdf_out,
                      float *f_out, uint32_t *flags_out, uint32_t *seeds_out);

/* ------------------------------------------------------------ Several GPUs
 * The reference drives one GPU (one GL context, renderer.cpp:124-150).  Here the samples of a render are
 * partitioned over GPUs (pt_set_partition: the scene is replicated, renderer g of G renders iterations g,
 * g + G, ...; exact because a sample depends only on pixel and iteration index, frag:866) and the per-GPU
 * sums are combined by ONE reduce over RCCL (xGMI inside a node).  No reference counterpart.
 *
 * A communicator is made either per process (one process per GPU: rank 0 calls pt_comm_unique_id, hands the
 * 128 bytes to the others by any means -- a file, MPI, a torch.distributed broadcast --, and every rank
 * calls pt_comm_create, collectively) or for all GPUs of one process at once (pt_comm_create_all). */
#define PT_COMM_ID_BYTES 128
typedef struct pt_comm pt_comm;
int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]);                        /* ncclGetUniqueId  */
/* device: HIP ordinal the rank's renderer lives on; -1 = current device.       ncclCommInitRank */
int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, pt_comm **out);
/* n communicators, rank k on devices[k], written to out[0..n-1].               ncclCommInitAll  */
int pt_comm_create_all(const int32_t *devices, uint32_t n, pt_comm **out);
void pt_comm_destroy(pt_comm *comm);
/* Which shared objects the reduce runs on, as text "hip=<path>\nrccl=<path>\nrccl_hip=<path>": the HIP runtime
 * this library is bound to, the RCCL it loaded (the copy beside that runtime when there is one), and the HIP
 * runtime that RCCL calls into.  A process may hold two HIP runtimes (a Python wheel's bundled copy and the
 * system's); the reduce is only meaningful when `hip` and `rccl_hip` name the same file.  Diagnostic; no
 * reference counterpart. */
int pt_comm_runtime_paths(char *out, size_t bytes);
/* ncclReduce(sum, float) of the renderers' sum buffers (width*height*4 floats, the alpha lane carrying the
 * sample counts) into rank `root`'s, in place, enqueued on each renderer's own stream after its pending
 * batches: asynchronous, pt_synchronize or any readback waits for it.  n = 1 in a process that drives one
 * GPU; a process that drives several passes all its renderers and communicators in one call (one group).
 * Afterwards the root's pt_read_radiance is the mean over all samples of all ranks.  The other ranks' sum
 * buffers are cleared (their samples now live in the root's buffer; iteration counters and statistics go
 * on), so the render may continue and be reduced again, e.g. for previews.  All image sizes must agree. */
int pt_reduce_sum(pt_renderer *const *renderers, pt_comm *const *comms, uint32_t n, uint32_t root);

/* Device pointer of the sum buffer (width*height*4 floats) so that a caller
 * can reduce it across GPUs without a host round trip; valid until
 * pt_resize / pt_destroy. */
int pt_sum_device_ptr(pt_renderer *r, void **dev_ptr);
/* The alpha lane of the sum buffer carries the per-pixel sample count, so a sum
 * reduced across ranks still normalises correctly in pt_read_radiance. */
int pt_get_stats(pt_renderer *r, pt_stats *stats);

/* Host-only: builds the BVH the renderer would build for this scene (no GPU
 * needed) and checks its invariants -- every triangle in exactly one leaf,
 * every child box encloses what is below it, depth as reported.  Returns
 * PT_ERR_STATE with a message when one is violated.  No reference counterpart
 * (the reference scans all triangles, pathtracer.frag.glsl:391-416). */
typedef struct {
    uint32_t n_triangles, n_nodes, n_leaves, depth, max_leaf_size, stack_entries;
    double sah_cost, build_ms;
} pt_bvh_info;
int pt_bvh_inspect(const pt_scene_desc *scene, pt_bvh_info *info);
/* The renderer's tree as it lies in device memory, for comparing the two builders:
 * nodes_host receives n_nodes records of 16 words (two child boxes as 12 floats, then
 * the two child codes as int32 and two zero words), triangles_host n_triangles
 * records of 16 words in leaf order (v0 xyz, d0, v1 xyz, n.x, v2 xyz, n.y, n.z, the
 * original triangle index with bit 30 set on surfaces that are not purely diffuse,
 * two zero words).  Either pointer may be null.  Sizes: pt_get_stats.  No reference
 * counterpart. */
int pt_read_bvh(pt_renderer *r, float *nodes_host, float *triangles_host);

#ifdef __cplusplus
}
#endif
#endif
