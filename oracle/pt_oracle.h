/*
 * pt_oracle.h -- CPU restatement of kbau121/PathTracer's GLSL integrator.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pathtracer_amd/,
 * include/) may include, link or call this.  It exists so that tests, the
 * smoke check and the benchmark's cpu_baseline leg have an independent
 * statement of what the reference computes.
 *
 * The scene arrays use the reference's own flat layout (scene.h:53-62 as
 * uploaded by renderer.cpp:32-83).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const float *vertices;      /* n_vertices x 3                      (scene.h:53)  */
    uint32_t n_vertices;
    const float *indices;       /* n_triangles x 4 floats i0,i1,i2,w   (scene.h:56)  */
    uint32_t n_triangles;
    const float *vertex_data;   /* 3*n_triangles x 6: normal, uv0      (scene.h:18-25) */
    const float *lights;        /* n_lights x 20: radiance4 + mat4 col-major (scene.h:27-38) */
    uint32_t n_lights;
    const float *materials;     /* n_materials x 8                     (scene.h:40-51) */
    uint32_t n_materials;
    const uint32_t *material_map; /* n_triangles                       (scene.h:62)  */
} oracle_scene;

typedef struct {
    float eye[3], forward[3], up[3], right[3]; /* camera.h:12-17 after update() */
    uint32_t width, height;
    float tan_half_fov;                        /* frag:306,315 constant        */
} oracle_camera;

typedef struct {
    uint64_t paths;
    uint64_t segments;      /* intersect() calls                      */
    uint64_t light_paths;   /* paths that ended on the light          */
    uint64_t escaped_paths; /* intersect() returned false             */
    uint64_t zeroed_paths;  /* pdf <= 0 or early "return vec3(0)"     */
    uint64_t nan_samples;   /* non-finite pass colour, replaced by 0  */
} oracle_stats;

/* frag:297-304.  seed is the uvec2 state, advanced in place. */
float oracle_rng(uint32_t seed[2]);

/* frag:866 + frag:306-320.  x,y are integer pixel coordinates, row 0 = bottom
 * image row (gl_FragCoord convention).  seed_out receives the RNG state after
 * the two jitter draws. */
void oracle_raycast(const oracle_camera *cam, uint32_t x, uint32_t y, uint32_t iteration,
                    float origin[3], float dir[3], uint32_t seed_out[2]);

/* frag:384-431 (geometry loop + light loop), without the attribute fetch.
 * Returns the hit code: triangle index >= 0, -1 miss, -(2+k) light k.
 * *t_out is the geometry t (the reference overwrites t with 0 on a light hit,
 * frag:427; here the geometry/light parametric distance is kept for tests). */
int32_t oracle_intersect(const oracle_scene *sc, const float o[3], const float d[3], float *t_out);

/* Brute-force intersect() for n arbitrary rays. */
void oracle_trace(const oracle_scene *sc, uint64_t n, const float *origins, const float *dirs,
                  float *t_out, int32_t *id_out);

/* flags[i] = 1 when some triangle yields st = NaN (0/0) for ray i. */
void oracle_poisoned(const oracle_scene *sc, uint64_t n, const float *origins, const float *dirs, uint8_t *flags);

/* Primary rays of one iteration: hit codes, t and ray directions per pixel
 * (pixel index = y*width + x, row 0 = bottom).  Any output may be NULL. */
void oracle_primary(const oracle_scene *sc, const oracle_camera *cam, uint32_t iteration,
                    int32_t *ids, float *t, float *dirs);

/* frag:841-858 for one shading point.  material = 8 floats.  seed is advanced
 * by the draws sampleSurface itself makes (1 or 2).  Returns 1 when the BxDF
 * took an early "return vec3(0)" path (inDir/pdf unwritten in the reference),
 * else 0.  f receives the BxDF value, wi the sampled direction. */
int oracle_sample_surface(const float material[8], const float normal[3], const float wo[3],
                          const float xi[2], uint32_t seed[2], float wi[3], float *pdf, float f[3]);

/* frag:864-905 for iterations [it_begin, it_end) with a float accumulation
 * target and non-finite pass colours replaced by 0: sum_rgba[4*pixel+c] +=
 * passCol; alpha lane += 1 per sample.  max_depth replaces the literal 10 of
 * frag:874.  nthreads <= 0 means 1. */
void oracle_render(const oracle_scene *sc, const oracle_camera *cam, uint32_t it_begin,
                   uint32_t it_end, uint32_t max_depth, float *sum_rgba, oracle_stats *stats,
                   int nthreads);

/* The as-shipped accumulator: RGBA8 UNORM running average that is read back
 * every iteration (renderer.cpp:21, frag:900-904).  accum_rgba8 holds W*H*4
 * bytes and is updated in place for iterations [it_begin, it_end). */
void oracle_render_rgba8(const oracle_scene *sc, const oracle_camera *cam, uint32_t it_begin,
                         uint32_t it_end, uint32_t max_depth, uint8_t *accum_rgba8, int nthreads);

/* post.frag:11-15 on n rgb triples (in place allowed). */
void oracle_tonemap(const float *rgb_in, float *rgb_out, uint64_t n);

/* Inverse of a column-major f32 mat4 computed in float64 and rounded to f32;
 * returns 0 if singular.  (frag:102 inverts per ray; the float64 inverse is
 * what reproduced the llvmpipe light mask.) */
int oracle_light_inverse(const float m[16], float inv[16]);

#ifdef __cplusplus
}
#endif
#endif
