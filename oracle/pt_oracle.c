/*
 * pt_oracle.c -- scalar CPU restatement of the reference integrator
 * (src/shaders/pathtracer/pathtracer.frag.glsl, post.frag.glsl).
 *
 * TEST INFRASTRUCTURE ONLY: never linked into, or called from, the product
 * library.  Brute-force O(#triangles) intersection exactly like the shader.
 *
 * Arithmetic: every operation is one IEEE binary32 op, no contraction (build
 * with -ffp-contract=off, never -ffast-math).  For the parts that must agree
 * bit for bit with the shader as Mesa llvmpipe executes it (rng, raycast,
 * triangle test, light test) the evaluation trees are:
 *     dot(a,b)      = a.x*b.x + (a.y*b.y + a.z*b.z)
 *     normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
 * The BxDF code uses sinf/cosf/powf whose precision is driver-defined in
 * GLSL, so beyond the first hit agreement is statistical, not bitwise.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_F 3.14159265358979323f     /* frag:3 */
#define INV_PI_F 0.31830988618379067f /* frag:4 */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline v3 cross3(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize3(v3 v)
{
    float r = 1.0f / sqrtf(dot3(v, v));
    return mul3(v, r);
}
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

/* ------------------------------------------------------------------ rng */

/* frag:297-304 */
float oracle_rng(uint32_t seed[2])
{
    seed[0] += 1u;
    seed[1] += 1u;
    uint32_t qx = 1103515245u * ((seed[0] >> 1) ^ seed[1]);
    uint32_t qy = 1103515245u * ((seed[1] >> 1) ^ seed[0]);
    uint32_t n = 1103515245u * (qx ^ (qy >> 3));
    return (float)n * 2.3283064365386963e-10f; /* 1.f/float(0xffffffffU) == 2^-32 */
}

/* -------------------------------------------------------------- raycast */

/* frag:866, frag:306-320 */
void oracle_raycast(const oracle_camera *cam, uint32_t x, uint32_t y, uint32_t iteration,
                    float origin[3], float dir[3], uint32_t seed_out[2])
{
    uint32_t seed[2];
    seed[0] = (iteration + 1u) * x;
    seed[1] = (iteration + 2u) * y;
    float ox = oracle_rng(seed);
    float oy = oracle_rng(seed);

    float sx = (((float)x + 0.5f) + ox) / (float)cam->width;
    float sy = (((float)y + 0.5f) + oy) / (float)cam->height;
    sx = sx * 2.f - 1.f;
    sy = sy * 2.f - 1.f;

    float aspect = (float)cam->width / (float)cam->height;
    v3 eye = ld3(cam->eye);
    v3 ref = add3(eye, ld3(cam->forward));
    v3 V = mul3(ld3(cam->up), cam->tan_half_fov);
    v3 H = mul3(mul3(ld3(cam->right), cam->tan_half_fov), aspect);
    v3 p = add3(add3(ref, mul3(H, sx)), mul3(V, sy));
    st3(origin, eye);
    st3(dir, normalize3(sub3(p, eye)));
    if (seed_out) { seed_out[0] = seed[0]; seed_out[1] = seed[1]; }
}

/* ------------------------------------------------------------ intersect */

int oracle_light_inverse(const float m[16], float inv[16])
{
    /* Gauss-Jordan on the 4x4 in float64; m and inv are column-major. */
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = (double)m[c * 4 + r];
            a[r][c + 4] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return 0;
        if (piv != col)
            for (int c = 0; c < 8; ++c) { double t = a[piv][c]; a[piv][c] = a[col][c]; a[col][c] = t; }
        double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            double f = a[r][col];
            if (f != 0.0)
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) inv[c * 4 + r] = (float)a[r][c + 4];
    return 1;
}

/* frag:363-382.  inv is column-major; rows are evaluated
 * ((m0*x + m1*y) + m2*z) + m3*w as the llvmpipe lowering does. */
static int rectangle_intersect(v3 o, v3 d, const float inv[16], float *t)
{
    float lox = ((inv[0] * o.x + inv[4] * o.y) + inv[8] * o.z) + inv[12] * 1.f;
    float loy = ((inv[1] * o.x + inv[5] * o.y) + inv[9] * o.z) + inv[13] * 1.f;
    float loz = ((inv[2] * o.x + inv[6] * o.y) + inv[10] * o.z) + inv[14] * 1.f;
    float ldx = ((inv[0] * d.x + inv[4] * d.y) + inv[8] * d.z) + inv[12] * 0.f;
    float ldy = ((inv[1] * d.x + inv[5] * d.y) + inv[9] * d.z) + inv[13] * 0.f;
    float ldz = ((inv[2] * d.x + inv[6] * d.y) + inv[10] * d.z) + inv[14] * 0.f;
    float dt = -ldz;
    if (dt < 0.f) return 0;
    float tl = loz / dt;
    *t = tl;
    if (tl < 0.f) return 0;
    float px = lox + ldx * tl;
    float py = loy + ldy * tl;
    return (fabsf(px) <= 0.5f && fabsf(py) <= 0.5f);
}

typedef struct {
    float t;
    int type;  /* -1 none, 0 geometry, 1 light */
    int index;
} hit_t;

/* frag:384-431.  light_inv: n_lights x 16 precomputed inverses. */
static hit_t intersect_scan(const oracle_scene *sc, const float *light_inv, v3 o, v3 d, float *t_param)
{
    hit_t h;
    h.t = INFINITY;
    h.type = -1;
    h.index = -1;
    for (uint32_t i = 0; i < sc->n_triangles; ++i) {
        const float *tri = sc->indices + 4 * (size_t)i;
        v3 v0 = ld3(sc->vertices + 3 * (size_t)(int)tri[0]);
        v3 v1 = ld3(sc->vertices + 3 * (size_t)(int)tri[1]);
        v3 v2 = ld3(sc->vertices + 3 * (size_t)(int)tri[2]);
        v3 e0 = sub3(v1, v0);
        v3 e1 = sub3(v2, v0);
        v3 n = normalize3(cross3(e0, e1));
        float st = (dot3(n, v0) - dot3(n, o)) / dot3(n, d);
        if (st < 0.f || st > h.t) continue;
        v3 Q = V3(o.x + st * d.x, o.y + st * d.y, o.z + st * d.z);
        if (dot3(cross3(sub3(v1, v0), sub3(Q, v0)), n) < 0.f) continue;
        if (dot3(cross3(sub3(v2, v1), sub3(Q, v1)), n) < 0.f) continue;
        if (dot3(cross3(sub3(v0, v2), sub3(Q, v2)), n) < 0.f) continue;
        h.t = st;
        h.index = (int)i;
        h.type = 0;
    }
    float tp = h.t;
    for (uint32_t l = 0; l < sc->n_lights; ++l) {
        float tl;
        if (!rectangle_intersect(o, d, light_inv + 16 * (size_t)l, &tl)) continue;
        if (tl < 0.f || tl > h.t) continue;
        tp = tl;
        h.t = 0.f; /* frag:427 */
        h.index = (int)l;
        h.type = 1;
    }
    if (t_param) *t_param = tp;
    return h;
}

static float *make_light_inverses(const oracle_scene *sc)
{
    float *inv = (float *)malloc(sizeof(float) * 16 * (sc->n_lights ? sc->n_lights : 1));
    for (uint32_t l = 0; l < sc->n_lights; ++l)
        if (!oracle_light_inverse(sc->lights + 20 * (size_t)l + 4, inv + 16 * (size_t)l))
            for (int k = 0; k < 16; ++k) inv[16 * (size_t)l + k] = NAN;
    return inv;
}

static int32_t hit_code(hit_t h)
{
    if (h.type == 0) return h.index;
    if (h.type == 1) return -(2 + h.index);
    return -1;
}

int32_t oracle_intersect(const oracle_scene *sc, const float o[3], const float d[3], float *t_out)
{
    float *inv = make_light_inverses(sc);
    hit_t h = intersect_scan(sc, inv, ld3(o), ld3(d), t_out);
    free(inv);
    return hit_code(h);
}

/* The same scan for a group of rays at once: the triangles are visited in the same ascending order for
 * every ray, each triangle's n and dot(n, v0) are evaluated once per group instead of once per ray (the
 * same operations, hence the same bits), and the triangle data is streamed once per group.  This is what
 * makes a scan over ten million triangles for a few thousand rays a matter of seconds. */
#define TRACE_GROUP 32
static void intersect_scan_group(const oracle_scene *sc, const float *light_inv, int count, const v3 *o, const v3 *d,
                                 hit_t *h, float *t_param)
{
    for (int r = 0; r < count; ++r) {
        h[r].t = INFINITY;
        h[r].type = -1;
        h[r].index = -1;
    }
    for (uint32_t i = 0; i < sc->n_triangles; ++i) {
        const float *tri = sc->indices + 4 * (size_t)i;
        v3 v0 = ld3(sc->vertices + 3 * (size_t)(int)tri[0]);
        v3 v1 = ld3(sc->vertices + 3 * (size_t)(int)tri[1]);
        v3 v2 = ld3(sc->vertices + 3 * (size_t)(int)tri[2]);
        v3 e0 = sub3(v1, v0);
        v3 e1 = sub3(v2, v0);
        v3 n = normalize3(cross3(e0, e1));
        float d0 = dot3(n, v0);
        v3 e12 = sub3(v2, v1), e20 = sub3(v0, v2);
        for (int r = 0; r < count; ++r) {
            float st = (d0 - dot3(n, o[r])) / dot3(n, d[r]);
            if (st < 0.f || st > h[r].t) continue;
            v3 Q = V3(o[r].x + st * d[r].x, o[r].y + st * d[r].y, o[r].z + st * d[r].z);
            if (dot3(cross3(e0, sub3(Q, v0)), n) < 0.f) continue;
            if (dot3(cross3(e12, sub3(Q, v1)), n) < 0.f) continue;
            if (dot3(cross3(e20, sub3(Q, v2)), n) < 0.f) continue;
            h[r].t = st;
            h[r].index = (int)i;
            h[r].type = 0;
        }
    }
    for (int r = 0; r < count; ++r) {
        float tp = h[r].t;
        for (uint32_t l = 0; l < sc->n_lights; ++l) {
            float tl;
            if (!rectangle_intersect(o[r], d[r], light_inv + 16 * (size_t)l, &tl)) continue;
            if (tl < 0.f || tl > h[r].t) continue;
            tp = tl;
            h[r].t = 0.f; /* frag:427 */
            h[r].index = (int)l;
            h[r].type = 1;
        }
        t_param[r] = tp;
    }
}

void oracle_trace(const oracle_scene *sc, uint64_t n, const float *origins, const float *dirs,
                  float *t_out, int32_t *id_out)
{
    float *inv = make_light_inverses(sc);
    const int64_t groups = (int64_t)((n + TRACE_GROUP - 1) / TRACE_GROUP);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t g = 0; g < groups; ++g) {
        v3 o[TRACE_GROUP], d[TRACE_GROUP];
        hit_t h[TRACE_GROUP];
        float tp[TRACE_GROUP];
        const int64_t first = g * TRACE_GROUP;
        const int count = (int)((int64_t)n - first < TRACE_GROUP ? (int64_t)n - first : TRACE_GROUP);
        for (int r = 0; r < count; ++r) {
            o[r] = ld3(origins + 3 * (first + r));
            d[r] = ld3(dirs + 3 * (first + r));
        }
        intersect_scan_group(sc, inv, count, o, d, h, tp);
        for (int r = 0; r < count; ++r) {
            if (t_out) t_out[first + r] = tp[r];
            if (id_out) id_out[first + r] = hit_code(h[r]);
        }
    }
    free(inv);
}

/* flags[i] = 1 when some triangle gives st = NaN for ray i (the ray lies
 * exactly in that triangle's plane: 0/0 at frag:404).  From that triangle on
 * the shader's "st > t" is always false, so its result depends on the scan
 * order rather than on distance; tests exclude such rays from tree-vs-scan
 * comparisons. */
void oracle_poisoned(const oracle_scene *sc, uint64_t n, const float *origins, const float *dirs, uint8_t *flags)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t r = 0; r < (int64_t)n; ++r) {
        v3 o = ld3(origins + 3 * r), d = ld3(dirs + 3 * r);
        uint8_t bad = 0;
        for (uint32_t i = 0; i < sc->n_triangles && !bad; ++i) {
            const float *tri = sc->indices + 4 * (size_t)i;
            v3 v0 = ld3(sc->vertices + 3 * (size_t)(int)tri[0]);
            v3 v1 = ld3(sc->vertices + 3 * (size_t)(int)tri[1]);
            v3 v2 = ld3(sc->vertices + 3 * (size_t)(int)tri[2]);
            v3 nn = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
            float st = (dot3(nn, v0) - dot3(nn, o)) / dot3(nn, d);
            if (st != st) bad = 1;
        }
        flags[r] = bad;
    }
}

void oracle_primary(const oracle_scene *sc, const oracle_camera *cam, uint32_t iteration,
                    int32_t *ids, float *t, float *dirs)
{
    float *inv = make_light_inverses(sc);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t y = 0; y < (int64_t)cam->height; ++y) {
        for (uint32_t x = 0; x < cam->width; ++x) {
            float o[3], d[3];
            oracle_raycast(cam, x, (uint32_t)y, iteration, o, d, NULL);
            size_t p = (size_t)y * cam->width + x;
            if (dirs) { dirs[3 * p] = d[0]; dirs[3 * p + 1] = d[1]; dirs[3 * p + 2] = d[2]; }
            if (ids || t) {
                float tp;
                hit_t h = intersect_scan(sc, inv, ld3(o), ld3(d), &tp);
                if (ids) ids[p] = hit_code(h);
                if (t) t[p] = tp;
            }
        }
    }
    free(inv);
}

/* --------------------------------------------------------------- shading */

typedef struct {
    v3 albedo;
    float roughness, metallic, ior, anisotropy, transmission;
} material_t;

/* frag:110-126; an out-of-range index reads zeros like a robust texelFetch. */
static material_t get_material(const oracle_scene *sc, uint32_t idx)
{
    material_t m;
    memset(&m, 0, sizeof m);
    if (idx < sc->n_materials) {
        const float *p = sc->materials + 8 * (size_t)idx;
        m.albedo = ld3(p);
        m.roughness = p[3];
        m.metallic = p[4];
        m.ior = p[5];
        m.anisotropy = p[6];
        m.transmission = p[7];
    }
    return m;
}

/* frag:128-147 */
static v3 barycentric(v3 p, v3 v0, v3 v1, v3 v2)
{
    v3 e0 = sub3(v1, v0), e1 = sub3(v2, v0), eP = sub3(p, v0);
    float e00 = dot3(e0, e0), e01 = dot3(e0, e1), e11 = dot3(e1, e1);
    float e0P = dot3(e0, eP), e1P = dot3(e1, eP);
    float inv = 1.f / (e00 * e11 - e01 * e01);
    float v = (e11 * e0P - e01 * e1P) * inv;
    float w = (e00 * e1P - e01 * e0P) * inv;
    float u = 1.f - v - w;
    return V3(u, v, w);
}

/* frag:154-165 */
static void coordinate_system(v3 v1, v3 *v2, v3 *v3o)
{
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float l = sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = V3(-v1.z / l, 0.f / l, v1.x / l);
    } else {
        float l = sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = V3(0.f / l, v1.z / l, -v1.y / l);
    }
    *v3o = cross3(v1, *v2);
}

typedef struct { v3 t, b, n; } frame_t; /* mat3(tangent, bitangent, normal) columns */

/* frag:176-181 */
static frame_t frame_from_normal(v3 n)
{
    frame_t f;
    f.n = n;
    coordinate_system(n, &f.t, &f.b);
    return f;
}
/* frag:169-174 */
static frame_t frame_from_normal_tangent(v3 n, v3 tangent)
{
    frame_t f;
    f.b = normalize3(cross3(n, tangent));
    f.t = cross3(f.b, n);
    f.n = n;
    return f;
}
static v3 to_world(frame_t f, v3 l)
{
    return add3(add3(mul3(f.t, l.x), mul3(f.b, l.y)), mul3(f.n, l.z));
}
static v3 to_local(frame_t f, v3 w) { return V3(dot3(f.t, w), dot3(f.b, w), dot3(f.n, w)); }

/* frag:197-249 */
static float cos2_theta(v3 v) { return v.z * v.z; }
static float sin2_theta(v3 v) { return fmaxf(0.f, 1.f - cos2_theta(v)); }
static float sin_theta(v3 v) { return sqrtf(sin2_theta(v)); }
static float tan2_theta(v3 v) { return sin2_theta(v) / cos2_theta(v); }
static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static float cos_phi(v3 v) { float s = sin_theta(v); return (s == 0.f) ? 1.f : clampf(v.x / s, -1.f, 1.f); }
static float sin_phi(v3 v) { float s = sin_theta(v); return (s == 0.f) ? 0.f : clampf(v.y / s, -1.f, 1.f); }
static float cos2_phi(v3 v) { float c = cos_phi(v); return c * c; }
static float sin2_phi(v3 v) { float s = sin_phi(v); return s * s; }

/* frag:255-276 */
static v3 square_to_disk_concentric(float xix, float xiy)
{
    float ux = xix * 2.f - 1.f, uy = xiy * 2.f - 1.f;
    float x2 = ux * ux, y2 = uy * uy;
    float pr = 0.f, pa = 0.f;
    if (x2 > y2) {
        pr = ux;
        pa = (PI_F / 4.f) * uy / ux;
    } else if (x2 <= y2 && y2 > 0.f) {
        pr = uy;
        pa = (PI_F / 2.f) - (PI_F / 4.f) * ux / uy;
    }
    return V3(cosf(pa) * pr, sinf(pa) * pr, 0.f);
}
/* frag:285-290 */
static v3 square_to_hemisphere_cosine(float xix, float xiy)
{
    v3 s = square_to_disk_concentric(xix, xiy);
    s.z = sqrtf(fmaxf(0.f, 1.f - s.x * s.x - s.y * s.y));
    return s;
}

/* frag:483-496 */
static float schlick(float R0, float c) { return R0 + (1.f - R0) * powf(1.f - c, 5.f); }

/* frag:581-593 */
static void roughness_to_anisotropic(float roughness, float anisotropy, float *ax, float *ay)
{
    *ax = roughness * roughness;
    *ay = *ax;
    if (anisotropy > 0.f) {
        float aspect = sqrtf(1.f - anisotropy * 0.9f);
        *ax /= aspect;
        *ay *= aspect;
    }
}
/* frag:595-600 */
static float iso_roughness2(v3 v, float ax, float ay) { return ax * ax * cos2_phi(v) + ay * ay * sin2_phi(v); }
/* frag:607-621 */
static float tr_distribution(v3 m, float ax, float ay)
{
    float t2 = tan2_theta(m);
    if (isinf(t2)) return 0.f;
    float c2 = cos2_theta(m);
    float c4 = c2 * c2;
    float e = (cos2_phi(m) / (ax * ax) + sin2_phi(m) / (ay * ay)) * t2;
    return 1.f / (PI_F * ax * ay * c4 * (1.f + e) * (1.f + e));
}
/* frag:623-629 */
static float tr_lambda(v3 v, float ax, float ay)
{
    float t2 = tan2_theta(v);
    if (isinf(t2)) return 0.f;
    return (sqrtf(1.f + iso_roughness2(v, ax, ay) * t2) - 1.f) * 0.5f;
}
/* frag:631-634 */
static float tr_masking(v3 wo, v3 wi, float ax, float ay)
{
    return 1.f / (1.f + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay));
}
/* frag:636-658 */
static v3 tr_sample_normal(v3 wo, float xix, float xiy, float ax, float ay)
{
    v3 h = normalize3(V3(ax * wo.x, ay * wo.y, wo.z));
    if (h.z < 0.f) h = neg3(h);
    v3 T1 = (h.z < 0.99999f) ? normalize3(cross3(V3(0.f, 0.f, 1.f), h)) : V3(1.f, 0.f, 0.f);
    v3 T2 = cross3(h, T1);
    /* frag:278-283 */
    float r = sqrtf(xix);
    float theta = 2.f * PI_F * xiy;
    float px = r * cosf(theta), py = r * sinf(theta);
    float hh = sqrtf(1.f - px * px);
    /* mix(x, y, a) = x*(1-a) + y*a with x = (1-h.z)/2, y = hh, a = py (frag:647) */
    float mx = (1.f - h.z) / 2.f;
    py = mx * (1.f - py) + hh * py;
    float pz = sqrtf(fmaxf(0.f, 1.f - (px * px + py * py)));
    v3 nh = add3(add3(mul3(T1, px), mul3(T2, py)), mul3(h, pz));
    return normalize3(V3(ax * nh.x, ay * nh.y, fmaxf(0.000001f, nh.z)));
}
/* frag:660-665, :674-677 */
static float tr_density(v3 wo, v3 m, float ax, float ay)
{
    float G1 = 1.f / (1.f + tr_lambda(wo, ax, ay));
    return G1 * tr_distribution(m, ax, ay) * fabsf(dot3(wo, m)) / fabsf(wo.z);
}
static v3 reflect3(v3 I, v3 N)
{
    float k = 2.f * dot3(N, I);
    return sub3(I, mul3(N, k));
}

/* Result of one BxDF call.  early = the shader took a "return vec3(0)" branch
 * that leaves inDir/pdf unwritten (frag:729,753,774,818,827). */
typedef struct { v3 f, wi; float pdf; int early; } bxdf_t;

static bxdf_t bxdf_zero(void)
{
    bxdf_t r;
    r.f = V3(0.f, 0.f, 0.f);
    r.wi = V3(0.f, 0.f, 0.f);
    r.pdf = 0.f;
    r.early = 1;
    return r;
}

/* frag:706-721 */
static bxdf_t diffuse_bxdf(v3 n, material_t mat, float xix, float xiy, v3 wo)
{
    bxdf_t r;
    if (dot3(n, wo) < 0.f) n = mul3(n, -1.f);
    v3 li = square_to_hemisphere_cosine(xix, xiy);
    r.wi = to_world(frame_from_normal(n), li);
    r.pdf = li.z * INV_PI_F;
    r.f = mul3(mat.albedo, INV_PI_F);
    r.early = 0;
    return r;
}

/* frag:549-577 */
static int refract3(v3 in, v3 normal, float eta, float *rel_eta, v3 *out)
{
    float ci = dot3(normal, in);
    if (ci < 0.f) {
        eta = 1.f / eta;
        ci = -ci;
        normal = neg3(normal);
    }
    float s2i = 1.f - ci * ci;
    float s2t = s2i / (eta * eta);
    if (s2t >= 1.f) return 0;
    float ct = sqrtf(1.f - s2t);
    float k = ci / eta - ct;
    *out = V3(-in.x / eta + k * normal.x, -in.y / eta + k * normal.y, -in.z / eta + k * normal.z);
    *rel_eta = eta;
    return 1;
}

/* frag:723-798 */
static bxdf_t dielectric_bxdf(v3 n, material_t mat, float xix, float xiy, v3 wo, uint32_t seed[2])
{
    float ax = mat.roughness, ay = mat.roughness;
    frame_t fr = frame_from_normal(n);
    v3 lo = to_local(fr, wo);
    if (lo.z == 0.f) return bxdf_zero();
    v3 m = tr_sample_normal(lo, xix, xiy, ax, ay);

    float R = schlick(0.05f, fabsf(dot3(lo, m)));
    float T = 1.f - R;
    float p_reflect = R;
    float p_transmit = T * mat.transmission;
    float p_diffuse = T * (1.f - mat.transmission);
    if (lo.z < 0.f) {
        p_transmit = T;
        p_diffuse = 0.f;
    }
    float u = oracle_rng(seed);
    bxdf_t r;
    r.early = 0;
    if (u <= p_reflect) {
        v3 li = reflect3(neg3(lo), m);
        if (li.z * lo.z <= 0.f) return bxdf_zero();
        r.wi = to_world(fr, li);
        r.pdf = tr_density(lo, m, ax, ay) / (4.f * dot3(lo, m)) * p_reflect;
        float D = tr_distribution(m, ax, ay);
        float G = tr_masking(lo, li, ax, ay);
        float v = D * G * R / (4.f * li.z * lo.z);
        r.f = V3(v, v, v);
        return r;
    } else if (u <= p_reflect + p_transmit) {
        float rel_eta = 0.f;
        v3 li = V3(0.f, 0.f, 0.f);
        int tir = !refract3(lo, m, mat.ior, &rel_eta, &li);
        if (tir || lo.z * li.z > 0.f || li.z == 0.f) return bxdf_zero();
        r.wi = to_world(fr, li);
        float denom = dot3(li, m) + dot3(lo, m) / rel_eta;
        float dm = fabsf(dot3(li, m)) / (denom * denom);
        r.pdf = tr_density(lo, m, ax, ay) * dm * p_transmit;
        float D = tr_distribution(m, ax, ay);
        float G = tr_masking(lo, li, ax, ay);
        float v = D * G * T * dot3(li, m) * dot3(lo, m) / (li.z * lo.z * denom * denom);
        r.f = mul3(mat.albedo, v);
        return r;
    } else {
        r = diffuse_bxdf(n, mat, xix, xiy, wo);
        r.pdf *= p_diffuse;
        return r;
    }
}

/* frag:686-702 */
static v3 microfacet_attenuation(material_t mat, v3 lo, v3 li, float ax, float ay)
{
    float co = fabsf(lo.z), ci = fabsf(li.z);
    v3 m = add3(lo, li);
    if (ci <= 0.f || co <= 0.f) return V3(0.f, 0.f, 0.f);
    if (m.x == 0.f && m.y == 0.f && m.z == 0.f) return V3(0.f, 0.f, 0.f);
    m = normalize3(m);
    float p5 = powf(1.f - fabsf(dot3(lo, m)), 5.f);
    v3 F = V3(mat.albedo.x + (1.f - mat.albedo.x) * p5, mat.albedo.y + (1.f - mat.albedo.y) * p5,
              mat.albedo.z + (1.f - mat.albedo.z) * p5);
    float D = tr_distribution(m, ax, ay);
    float G = tr_masking(lo, li, ax, ay);
    float k = 4.f * ci * co;
    return V3(D * G * F.x / k, D * G * F.y / k, D * G * F.z / k);
}

/* frag:800-837 (the axisAngle rotation by 0 degrees is the identity) */
static bxdf_t microfacet_bxdf(v3 n, material_t mat, float xix, float xiy, v3 wo)
{
    if (dot3(n, wo) < 0.f) n = mul3(n, -1.f);
    float ax, ay;
    roughness_to_anisotropic(mat.roughness, mat.anisotropy, &ax, &ay);
    frame_t fr = frame_from_normal_tangent(n, V3(0.f, 0.f, 1.f));
    v3 lo = to_local(fr, wo);
    if (lo.z == 0.f) return bxdf_zero();
    v3 m = tr_sample_normal(lo, xix, xiy, ax, ay);
    v3 li = reflect3(neg3(lo), m);
    if (li.z * lo.z <= 0.f) return bxdf_zero();
    bxdf_t r;
    r.early = 0;
    r.wi = to_world(fr, li);
    r.pdf = tr_density(lo, m, ax, ay) / (4.f * dot3(lo, m));
    r.f = microfacet_attenuation(mat, lo, li, ax, ay);
    return r;
}

/* frag:841-858 */
static bxdf_t sample_surface(material_t mat, v3 n, float xix, float xiy, v3 wo, uint32_t seed[2])
{
    if (mat.metallic >= oracle_rng(seed)) return microfacet_bxdf(n, mat, xix, xiy, wo);
    if (mat.roughness < 1.f) return dielectric_bxdf(n, mat, xix, xiy, wo, seed);
    return diffuse_bxdf(n, mat, xix, xiy, wo);
}

int oracle_sample_surface(const float material[8], const float normal[3], const float wo[3],
                          const float xi[2], uint32_t seed[2], float wi[3], float *pdf, float f[3])
{
    material_t m;
    m.albedo = ld3(material);
    m.roughness = material[3];
    m.metallic = material[4];
    m.ior = material[5];
    m.anisotropy = material[6];
    m.transmission = material[7];
    bxdf_t r = sample_surface(m, ld3(normal), xi[0], xi[1], ld3(wo), seed);
    st3(wi, r.wi);
    st3(f, r.f);
    *pdf = r.pdf;
    return r.early;
}

/* ----------------------------------------------------------- path loop */

/* frag:864-901: one pixel-sample; returns attenuation*intensity. */
static v3 trace_path(const oracle_scene *sc, const float *light_inv, const oracle_camera *cam,
                     uint32_t x, uint32_t y, uint32_t iteration, uint32_t max_depth, oracle_stats *st)
{
    float of[3], df[3];
    uint32_t seed[2];
    oracle_raycast(cam, x, y, iteration, of, df, seed);
    v3 o = ld3(of), d = ld3(df);
    v3 att = V3(1.f, 1.f, 1.f);
    v3 intensity = V3(0.f, 0.f, 0.f);
    st->paths++;
    for (uint32_t i = 0; i < max_depth; ++i) {
        st->segments++;
        hit_t h = intersect_scan(sc, light_inv, o, d, NULL);
        if (h.index == -1) { st->escaped_paths++; break; }
        if (h.type == 1) {
            intensity = ld3(sc->lights + 20 * (size_t)h.index);
            st->light_paths++;
            break;
        }
        /* frag:434-453 */
        const float *tri = sc->indices + 4 * (size_t)h.index;
        int data = (int)tri[3];
        v3 p = V3(o.x + d.x * h.t, o.y + d.y * h.t, o.z + d.z * h.t);
        v3 v0 = ld3(sc->vertices + 3 * (size_t)(int)tri[0]);
        v3 v1 = ld3(sc->vertices + 3 * (size_t)(int)tri[1]);
        v3 v2 = ld3(sc->vertices + 3 * (size_t)(int)tri[2]);
        v3 b = barycentric(p, v0, v1, v2);
        v3 n1 = ld3(sc->vertex_data + 6 * ((size_t)data * 3 + 0));
        v3 n2 = ld3(sc->vertex_data + 6 * ((size_t)data * 3 + 1));
        v3 n3 = ld3(sc->vertex_data + 6 * ((size_t)data * 3 + 2));
        v3 n = add3(add3(mul3(n1, b.x), mul3(n2, b.y)), mul3(n3, b.z));

        float xix = oracle_rng(seed);
        float xiy = oracle_rng(seed);
        material_t mat = get_material(sc, sc->material_map[h.index]);
        bxdf_t r = sample_surface(mat, n, xix, xiy, neg3(d), seed);
        if (r.early) {
            /* f = 0 and pdf unwritten: the path carries nothing from here on. */
            att = V3(0.f, 0.f, 0.f);
            st->zeroed_paths++;
            break;
        }
        float c = fabsf(dot3(n, r.wi));
        att = V3(att.x * (r.f.x * c), att.y * (r.f.y * c), att.z * (r.f.z * c));
        if (r.pdf <= 0.f) {
            att = V3(0.f, 0.f, 0.f);
            st->zeroed_paths++;
            break;
        }
        att = V3(att.x / r.pdf, att.y / r.pdf, att.z / r.pdf);
        o = V3(p.x + r.wi.x * 0.0001f, p.y + r.wi.y * 0.0001f, p.z + r.wi.z * 0.0001f);
        d = r.wi;
    }
    return V3(att.x * intensity.x, att.y * intensity.y, att.z * intensity.z);
}

static void stats_add(oracle_stats *a, const oracle_stats *b)
{
    a->paths += b->paths;
    a->segments += b->segments;
    a->light_paths += b->light_paths;
    a->escaped_paths += b->escaped_paths;
    a->zeroed_paths += b->zeroed_paths;
    a->nan_samples += b->nan_samples;
}

void oracle_render(const oracle_scene *sc, const oracle_camera *cam, uint32_t it_begin,
                   uint32_t it_end, uint32_t max_depth, float *sum_rgba, oracle_stats *stats,
                   int nthreads)
{
    float *inv = make_light_inverses(sc);
    oracle_stats total;
    memset(&total, 0, sizeof total);
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        oracle_stats local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = 0; y < (int64_t)cam->height; ++y) {
            for (uint32_t x = 0; x < cam->width; ++x) {
                float *px = sum_rgba + 4 * ((size_t)y * cam->width + x);
                for (uint32_t it = it_begin; it < it_end; ++it) {
                    v3 c = trace_path(sc, inv, cam, x, (uint32_t)y, it, max_depth, &local);
                    if (!(isfinite(c.x) && isfinite(c.y) && isfinite(c.z))) {
                        local.nan_samples++;
                        c = V3(0.f, 0.f, 0.f);
                    }
                    px[0] += c.x;
                    px[1] += c.y;
                    px[2] += c.z;
                    px[3] += 1.f;
                }
            }
        }
#pragma omp critical
        stats_add(&total, &local);
    }
    if (stats) *stats = total;
    free(inv);
}

/* UNORM8 conversion as GL defines it: clamp to [0,1], round(x*255). NaN -> 0. */
static uint8_t to_unorm8(float x)
{
    if (!(x > 0.f)) return 0;
    if (x >= 1.f) return 255;
    return (uint8_t)(int)(x * 255.f + 0.5f);
}

void oracle_render_rgba8(const oracle_scene *sc, const oracle_camera *cam, uint32_t it_begin,
                         uint32_t it_end, uint32_t max_depth, uint8_t *accum, int nthreads)
{
    float *inv = make_light_inverses(sc);
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t y = 0; y < (int64_t)cam->height; ++y) {
        oracle_stats local;
        memset(&local, 0, sizeof local);
        for (uint32_t x = 0; x < cam->width; ++x) {
            uint8_t *px = accum + 4 * ((size_t)y * cam->width + x);
            for (uint32_t it = it_begin; it < it_end; ++it) {
                v3 c = trace_path(sc, inv, cam, x, (uint32_t)y, it, max_depth, &local);
                /* frag:900-904 */
                float fi = (float)it, fi1 = (float)(it + 1u);
                float a0 = (float)px[0] / 255.f, a1 = (float)px[1] / 255.f, a2 = (float)px[2] / 255.f;
                px[0] = to_unorm8((a0 * fi + c.x) / fi1);
                px[1] = to_unorm8((a1 * fi + c.y) / fi1);
                px[2] = to_unorm8((a2 * fi + c.z) / fi1);
                px[3] = 255;
            }
        }
    }
    free(inv);
}

/* post.frag:11-15 */
void oracle_tonemap(const float *in, float *out, uint64_t n)
{
    for (uint64_t i = 0; i < 3 * n; ++i) {
        float c = in[i];
        c = c / (1.f + c);
        out[i] = powf(c, 1.f / 2.2f);
    }
}
