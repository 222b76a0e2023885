// Wavefront path-tracing integrator for gfx950 (MI355X): raygen, extend
// (BVH traversal with an LDS-resident stack, or a linear scan), shade (BSDF
// sampling + wave64 ballot compaction) and accumulate.
//
// Arithmetic contract.  This file is compiled with -ffp-contract=off: every
// '*', '+', '-', '/' below is one IEEE binary32 operation, and a fused
// multiply-add happens only where fmaf() is spelled out (the conservative box
// test).  The RNG, the camera ray, the triangle test and the light test follow
// the evaluation trees of the reference shader as Mesa llvmpipe runs it
// (pathtracer.frag.glsl:297-320, 363-431):
//     dot(a,b)     = a.x*b.x + (a.y*b.y + a.z*b.z)
//     normalize(v) = v * (1.0f / sqrtf(dot(v,v)))
// so the primary hit buffer can be compared bit for bit.
#include "kernels.h"
#include "tuning.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

namespace pt {

namespace {

constexpr int kBlock = 256;
// Experiment knobs of the traversal kernel: threads per block (the block shares
// one LDS copy of the scene) and the type of a stack entry.
#ifndef PT_STACK_T
#define PT_STACK_T int
#endif
using StackT = PT_STACK_T;
static_assert(sizeof(StackT) == 4, "a stack entry holds a child code or the end marker, 32 bits");
// Node steps per iteration of the traversal's inner loop beyond the first.  Measured on MI355X: one more step
// per iteration makes the traversal 2 % (Box) to 3.7 % (Veggie, four waves per SIMD) faster -- the loop's
// bookkeeping is some twenty scalar instructions around 27 vector ones --, two or three more change nothing further.
#ifndef PT_PHASE_A_UNROLL
#define PT_PHASE_A_UNROLL 1
#endif
// Minimum waves per SIMD the register allocator must leave room for (second
// argument of __launch_bounds__).
#ifndef PT_EXTEND_WAVES
#define PT_EXTEND_WAVES 1
#endif
// Measured on MI355X (Box / Veggie at 1080p): shade is fastest at 4-5 waves per SIMD
// (96 VGPRs, 24 bytes of scratch); extend is fastest with its natural 74 VGPRs.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4
#endif
constexpr float kPi = 3.14159265358979323f;     // frag:3
constexpr float kInvPi = 0.31830988618379067f;  // frag:4

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dotm(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ f3 crossm(f3 a, f3 b)
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ f3 normalizem(f3 v)
{
    float r = 1.0f / sqrtf(dotm(v, v));
    return v * r;
}

// frag:297-304: all uint32 wrap-around; float(n) * 2^-32.
__device__ __forceinline__ float rng_next(uint32_t &sx, uint32_t &sy)
{
    sx += 1u;
    sy += 1u;
    uint32_t qx = 1103515245u * ((sx >> 1) ^ sy);
    uint32_t qy = 1103515245u * ((sy >> 1) ^ sx);
    uint32_t n = 1103515245u * (qx ^ (qy >> 3));
    return static_cast<float>(n) * 2.3283064365386963e-10f;
}

// ---------------------------------------------------------------- raygen

// The word stored beside a ray (PathQueue::lt).  Its magnitude is the distance at which the ray meets the
// scene's one light, or a quiet NaN when it does not (or when the scene has another number of lights); its
// sign bit marks a path whose throughput is no longer finite in every channel (see k_shade).
constexpr uint32_t kNoLightBits = 0x7FC00000u;
// What a shading pass is for (shade_body): an ordinary bounce, the hits of the last allowed segment (only
// non-finite samples come of them), or the bounce whose rays will travel the last allowed segment.
constexpr uint32_t kShadeBounce = 0u, kShadeFinal = 1u, kShadeBeforeLast = 2u;
constexpr uint32_t kPoisonBit = 0x80000000u;

// frag:363-382 for one light: distance along the ray to the light's rectangle,
// NaN when the ray does not meet it (from behind, beside, or never).
// Rows evaluated ((m0*x + m1*y) + m2*z) + m3*w, as in test_lights below.
__device__ __forceinline__ float light_distance(const DeviceLight &L, f3 o, f3 d)
{
    const float none = __uint_as_float(kNoLightBits);
    const float *m = L.inv;
    const float lox = ((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.f;
    const float loy = ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.f;
    const float loz = ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.f;
    const float ldx = ((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.f;
    const float ldy = ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.f;
    const float ldz = ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.f;
    const float dt = -ldz;
    if (dt < 0.f) return none;
    const float tl = loz / dt;
    if (tl < 0.f) return none;
    const float px = lox + ldx * tl, py = loy + ldy * tl;
    if (!(fabsf(px) <= 0.5f && fabsf(py) <= 0.5f)) return none;
    return fabsf(tl);  // finite and >= 0 here (an infinite or NaN tl fails the rectangle test); -0 becomes +0
}

// Which pixel a path slot of an iteration renders.  Consecutive slots are
// grouped into units of 64 pixels that lie together in the image (an 8 x 8 tile
// when both image sizes are multiples of 8, else 64 pixels of a row), which
// keeps the rays of a wave coherent; consecutive units of the queue are taken
// from all over the image (unit u of the queue = unit u * stride mod n of the
// image, stride near n / golden ratio), so that every region of the queue
// holds the same mix of empty background and geometry and the persistent waves
// get equal shares of the work.  The image does not depend on the mapping: a
// sample is a function of (pixel, iteration) only.
// x / d for x < 2^31 with the magic number the host made for d (device_types.h); rem = x % d.
__device__ __forceinline__ uint32_t div_by(uint32_t x, uint32_t d, FastDiv f, uint32_t &rem)
{
    const uint32_t q = f.magic ? (__umulhi(x, f.magic) >> f.shift) : x;
    rem = x - q * d;
    return q;
}

__device__ __forceinline__ uint32_t pixel_of_slot(const DeviceCamera &cam, uint32_t j)
{
    const uint32_t u = j >> 6;
    if (cam.map_mode == 0u || u >= cam.map_units) return j;
    // unit u of the queue = unit (u * stride) mod n of the image, stride near n / golden ratio and coprime to n:
    // any run of consecutive units (a region) is spread evenly over the whole image.  (A cheaper permutation by
    // bit reversal was measured: traversal 10 % slower -- the waves' shares of the work are less equal.)
    uint32_t unit;
    if (cam.div_map_units.magic != 0xFFFFFFFFu)  // the usual case: the product fits in 31 bits
        (void)div_by(u * cam.map_stride, cam.map_units, cam.div_map_units, unit);
    else
        unit = static_cast<uint32_t>((static_cast<unsigned long long>(u) * cam.map_stride) % cam.map_units);
    const uint32_t in_unit = j & 63u;
    if (cam.map_mode == 1u) return (unit << 6) + in_unit;
    uint32_t tx;
    const uint32_t ty = div_by(unit, cam.tiles_x, cam.div_tiles_x, tx);
    return ((ty << 3) + (in_unit >> 3)) * cam.width + (tx << 3) + (in_unit & 7u);
}

// frag:866 + frag:306-320.  Path slot i = k * (W*H) + j, j-th slot of iteration k -> pixel_of_slot.
__global__ void __launch_bounds__(64) k_set_batch(BatchParams *params, DeviceCamera cam, uint32_t it_first, uint32_t sample_tag)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        params->cam = cam;
        params->it_first = it_first;
        params->sample_tag = sample_tag;
    }
}

// The camera's frame as the ray generation uses it (frag:306-320).
struct PrimaryFrame {
    f3 eye, ref, H, V;
};
__device__ __forceinline__ PrimaryFrame primary_frame(const DeviceCamera &cam)
{
    const f3 eye = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    const f3 fwd = mk(cam.forward[0], cam.forward[1], cam.forward[2]);
    const f3 up = mk(cam.up[0], cam.up[1], cam.up[2]);
    const f3 right = mk(cam.right[0], cam.right[1], cam.right[2]);
    return PrimaryFrame{eye, eye + fwd, (right * cam.tan_half_fov) * cam.aspect, up * cam.tan_half_fov};
}
// The primary ray of pixel (x, y) in iteration `iteration` (frag:866, :306-320).
struct PrimaryRay {
    f3 d;
    uint32_t sx, sy;  // the path's RNG state after the two jitter draws
};
__device__ __forceinline__ PrimaryRay primary_ray(const DeviceCamera &cam, const PrimaryFrame &f, uint32_t x, uint32_t y, uint32_t iteration)
{
    const float fx = static_cast<float>(x) + 0.5f, fy = static_cast<float>(y) + 0.5f;
    PrimaryRay r;
    r.sx = (iteration + 1u) * x;
    r.sy = (iteration + 2u) * y;
    const float jx = rng_next(r.sx, r.sy);
    const float jy = rng_next(r.sx, r.sy);
    float scx = (fx + jx) / cam.width_f;
    float scy = (fy + jy) / cam.height_f;
    scx = scx * 2.f - 1.f;
    scy = scy * 2.f - 1.f;
    const f3 p = (f.ref + f.H * scx) + f.V * scy;
    r.d = normalizem(p - f.eye);
    return r;
}

// Readbacks: every pixel's ray at the slot the slot -> pixel mapping gives it, no path state (lane p2) stored.
__global__ void __launch_bounds__(kBlock) k_raygen(SceneView sc, const BatchParams *__restrict__ bp, uint32_t it_stride,
                                                    uint32_t n_iterations, PathQueue out, uint32_t *counts_out,
                                                    uint32_t region_slots)
{
    const DeviceCamera cam = bp->cam;
    const uint32_t it_first = bp->it_first;
    const uint32_t n_pixels = cam.width * cam.height;
    const uint32_t total = n_pixels * n_iterations;
    const PrimaryFrame frame = primary_frame(cam);
    // A thread takes slot j of every iteration of the batch: the slot -> pixel arithmetic is done once for all
    // of them, and the stores of one iteration are still consecutive over the threads.
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n_pixels; j += gridDim.x * kBlock) {
        const uint32_t pixel = pixel_of_slot(cam, j);
        uint32_t x;
        const uint32_t y = div_by(pixel, cam.width, cam.div_width, x);
        for (uint32_t k = 0; k < n_iterations; ++k) {
            const PrimaryRay r = primary_ray(cam, frame, x, y, it_first + k * it_stride);
            const uint32_t i = k * n_pixels + j;
            out.p0[i] = make_float4(frame.eye.x, frame.eye.y, frame.eye.z, r.d.x);
            out.p1[i] = make_float4(r.d.y, r.d.z, __uint_as_float(r.sx), __uint_as_float(r.sy));
            out.lt[i] = sc.n_lights == 1u ? light_distance(sc.lights[0], frame.eye, r.d) : __uint_as_float(kNoLightBits);
        }
    }
    // rays per region of the new queue: full regions and a ragged last one
    const uint32_t n_regions = (total + region_slots - 1u) / region_slots;
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < n_regions; r += gridDim.x * kBlock)
        counts_out[r] = min(region_slots, total - r * region_slots);
}

// ---------------------------------------------------------------- extend

// The wave's index within its block and the region counts it reads are the same in all 64 lanes.  Said so in a
// way the compiler can see (v_readfirstlane), everything derived from them -- the wave's turn, its region, the
// region's base and counts, the hit-list fill counts -- lives in scalar registers and is stepped by scalar
// instructions, and the branches on them are scalar branches instead of exec-mask manipulations.  The counts
// are loaded by a vector load all the same (a relaxed atomic load, never a scalar load): k_bounces reads counts
// that the same wave wrote earlier in the same launch, and the scalar cache is not coherent with vector stores.
#ifndef PT_UNIFORM_WAVE
#define PT_UNIFORM_WAVE 1
#endif
__device__ __forceinline__ uint32_t wave_in_block()
{
#if PT_UNIFORM_WAVE
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
#else
    return threadIdx.x >> 6;
#endif
}
__device__ __forceinline__ uint32_t load_count(const uint32_t *p)
{
#if PT_UNIFORM_WAVE
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
#else
    return *p;
#endif
}

// Region bookkeeping shared by extend and shade: which region a wave handles
// on its k-th turn.
__device__ __forceinline__ uint32_t region_of(const Regions &rg, uint32_t k)
{
    return static_cast<uint32_t>((static_cast<unsigned long long>(k) * rg.stride) % rg.n_regions);
}
// A wave takes the turns k0, k0 + n_waves, ... of its grid.  The 64-bit modulo above is a long routine; a wave
// does it twice, when it starts, and from then on steps from one of its regions to the next with an add and
// a conditional subtract: region(k + n_waves) = (region(k) + step) mod n_regions, step = (n_waves * stride)
// mod n_regions.  The same (r, step) serve every phase of k_bounces: all phases walk the same regions.
struct WaveTurns {
    uint32_t r;     // region of the wave's next turn
    uint32_t step;  // (n_waves * stride) mod n_regions
};
__device__ __forceinline__ WaveTurns wave_turns(const Regions &rg, uint32_t k0, uint32_t n_waves)
{
    return WaveTurns{region_of(rg, k0), region_of(rg, n_waves)};
}
__device__ __forceinline__ uint32_t next_region(const Regions &rg, uint32_t r, uint32_t step)
{
    r += step;  // both below n_regions < 2^31
    return r >= rg.n_regions ? r - rg.n_regions : r;
}

// LDS by address.  The traversal kernel keeps LDS addresses (32-bit, as the ds_* instructions take them) in its
// per-lane state instead of indices, so that following a link costs no arithmetic.
typedef __attribute__((address_space(3))) char LdsChar;
typedef int LdsInt2 __attribute__((ext_vector_type(2)));
typedef float LdsFloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_address(const void *p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const LdsChar *)p));
}
template <typename T>
__device__ __forceinline__ T lds_load(uint32_t address)
{
    return *(const __attribute__((address_space(3))) T *)static_cast<uintptr_t>(address);
}
template <typename T>
__device__ __forceinline__ void lds_store(uint32_t address, T value)
{
    *(__attribute__((address_space(3))) T *)static_cast<uintptr_t>(address) = value;
}

struct Hit {
    float t;
    int id;  // original triangle index (| kClassBit for a surface that is not purely diffuse), -1 none
};
constexpr int kClassBit = 1 << 30;
constexpr int kFinished = static_cast<int>(0x80000000u);  // traversal state of a lane with nothing left to visit
constexpr int kIdle = static_cast<int>(0x80000001u);      // ... of a lane that holds no ray (no leaf code can look like it)
constexpr int kIdMask = kClassBit - 1;

// frag:393-415 with n and dot(n,v0) hoisted (same operations, same bits).
// Among equal t the larger original index wins, which is what visiting the
// triangles in ascending index order with "reject only if st > t" produces.
//
// SCAN_PREDICATE = true is the shader's acceptance test verbatim: "st < 0 || st > t" lets a NaN or
// infinite st through (every comparison is false on them), which is how a scan over ALL triangles comes
// to report hits at t = inf on planes the ray is parallel to, and "hits" on zero-area triangles (NaN
// normal) by every ray.  A tree visits only the leaves whose boxes the ray meets, so it could reproduce
// those only by accident; its leaf test (false) therefore accepts finite st >= 0 only (-0 included, as
// the shader does), and the hits at infinity are added by the separate pass over all triangles (EXACT).
// A zero-area triangle can then never be hit through the tree, nor by that pass (NaN == 0 is false).
template <bool SCAN_PREDICATE>
__device__ __forceinline__ void test_triangle(const float4 *__restrict__ tp, f3 o, f3 d, Hit &h)
{
    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
    const f3 v0 = mk(t0.x, t0.y, t0.z), v1 = mk(t1.x, t1.y, t1.z), v2 = mk(t2.x, t2.y, t2.z);
    const f3 n = mk(t1.w, t2.w, t3.x);
    const float st = (t0.w - dotm(n, o)) / dotm(n, d);
    if (SCAN_PREDICATE) {
        if (st < 0.f || st > h.t) return;
    } else {
        // v_cmp_class: -0, +0, positive denormal, positive normal
        if (!__builtin_amdgcn_classf(st, 0x1E0) || st > h.t) return;
    }
    const f3 Q = mk(o.x + st * d.x, o.y + st * d.y, o.z + st * d.z);
    if (dotm(crossm(v1 - v0, Q - v0), n) < 0.f) return;
    if (dotm(crossm(v2 - v1, Q - v1), n) < 0.f) return;
    if (dotm(crossm(v0 - v2, Q - v2), n) < 0.f) return;
    const int orig = __float_as_int(t3.y);  // index | class bit
    if (st == h.t && (orig & kIdMask) < (h.id < 0 ? h.id : (h.id & kIdMask))) return;
    h.t = st;
    h.id = orig;
}

// frag:363-382, :418-430.  Rows evaluated ((m0*x + m1*y) + m2*z) + m3*w.
__device__ __forceinline__ void test_lights(const SceneView &sc, f3 o, f3 d, float &t, int &code)
{
    for (uint32_t l = 0; l < sc.n_lights; ++l) {
        const float *m = sc.lights[l].inv;
        const float lox = ((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.f;
        const float loy = ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.f;
        const float loz = ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.f;
        const float ldx = ((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.f;
        const float ldy = ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.f;
        const float ldz = ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.f;
        const float dt = -ldz;
        if (dt < 0.f) continue;
        const float tl = loz / dt;
        if (tl < 0.f) continue;
        const float px = lox + ldx * tl, py = loy + ldy * tl;
        if (!(fabsf(px) <= 0.5f && fabsf(py) <= 0.5f)) continue;
        // A later light competes against t = 0 once one has been hit (frag:427).
        const float limit = (code <= -2) ? 0.f : t;
        if (tl < 0.f || tl > limit) continue;
        t = tl;
        code = -(2 + static_cast<int>(l));
    }
}

// A path whose throughput has a non-finite channel and that ends without reaching a light: the reference's
// sample is attenuation * vec3(0) (frag:901), NaN in exactly those channels.
__device__ __forceinline__ void write_poisoned_sample(float4 *__restrict__ samples, float4 p2, uint32_t sample_tag)
{
    samples[__float_as_uint(p2.w)] = make_float4(p2.x * 0.f, p2.y * 0.f, p2.z * 0.f, __uint_as_float(sample_tag));
}

// The hit list of a region has two ends: purely diffuse surfaces are appended
// upwards from the region's first slot, all other surfaces downwards from its
// last, so that the shade kernel's waves mostly see one kind of material.
// Ballot + prefix rank per end; n0 / n1 are the wave-uniform counts so far.
__device__ __forceinline__ void append_hit(float4 *__restrict__ hit_list, uint32_t region_base, uint32_t size, bool append,
                                           bool other, float4 rec, uint32_t &n0, uint32_t &n1)
{
    const unsigned long long m0 = __ballot(append && !other), m1 = __ballot(append && other);
    if (append) {
        const unsigned long long m = other ? m1 : m0;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        const uint32_t at = other ? (size - 1u - (n1 + rank)) : (n0 + rank);
        hit_list[region_base + at] = rec;
    }
    n0 += static_cast<uint32_t>(__popcll(m0));
    n1 += static_cast<uint32_t>(__popcll(m1));
}

// Linear scan in the reference's triangle order (frag:391-416).
template <bool RENDER>
__global__ void __launch_bounds__(kBlock) k_extend_scan(SceneView sc, Regions rg, PathQueue q,
                                                         float4 *__restrict__ hit_list, float4 *__restrict__ samples,
                                                         uint32_t last_segment,
                                                         const BatchParams *__restrict__ bp)
{
    const uint32_t sample_tag = RENDER ? bp->sample_tag : 0u;
    float2 *__restrict__ results = reinterpret_cast<float2 *>(hit_list);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    const uint32_t k0 = blockIdx.x * (kBlock / 64) + wave_in_block();
    const WaveTurns turns = wave_turns(rg, k0, n_waves);
    uint32_t r = turns.r;
    for (uint32_t k = k0; k < rg.n_regions; k += n_waves, r = next_region(rg, r, turns.step)) {
      const uint32_t n = load_count(rg.counts_in + r);
      const uint32_t region_base = r * rg.size;
      uint32_t hit_n0 = 0, hit_n1 = 0;
      for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
        bool append = false;
        float t = 0.f;
        int code = -1;
        bool other = false;
        const uint32_t i = region_base + j0 + lane;
        if (j0 + lane < n) {
            const float4 a = q.p0[i], b = q.p1[i];
            const f3 o = mk(a.x, a.y, a.z), d = mk(a.w, b.x, b.y);
            Hit h{__builtin_inff(), -1};
            for (uint32_t tri = 0; tri < sc.n_triangles; ++tri) test_triangle<true>(sc.triangles + 4 * static_cast<size_t>(tri), o, d, h);
            t = h.t;
            code = h.id < 0 ? h.id : (h.id & kIdMask);
            other = h.id >= 0 && (h.id & kClassBit) != 0;
            test_lights(sc, o, d, t, code);
            if (!RENDER) {
                results[i] = make_float2(t, __int_as_float(code));
            } else if (code <= -2) {
                const float4 c = q.p2[i];
                const float *rad = sc.lights[-(code + 2)].radiance;
                const f3 L = mk(c.x * rad[0], c.y * rad[1], c.z * rad[2]);
                // a non-finite sample is left to the accumulate kernel (which drops and counts it)
                samples[__float_as_uint(c.w)] = make_float4(L.x, L.y, L.z, __uint_as_float(sample_tag));
            } else {
                append = code >= 0 && last_segment == 0u;
                if (!append && (__float_as_uint(q.lt[i]) & kPoisonBit)) write_poisoned_sample(samples, q.p2[i], sample_tag);
            }
        }
        if (RENDER)
            append_hit(hit_list, region_base, rg.size, append, other,
                       make_float4(t, __int_as_float(code), __uint_as_float(j0 + lane), 0.f), hit_n0, hit_n1);
      }
      if (RENDER && lane == 0) rg.counts_out[r] = hit_n0 | (hit_n1 << 16);
    }
}

// BVH traversal.  Each lane keeps its stack of deferred children in LDS,
// laid out [entry][lane] so that a push/pop of all lanes touches consecutive
// banks.  The stack can never hold more than one entry per inner level of the
// tree, plus the end marker in entry 0; the host passes stack_entries > tree depth.
//
// Box test: the reference accepts hit points that lie a few ulp outside the
// exact triangle, so a box may only be culled when the ray misses it by a
// clear margin: every slab is widened by P = 2^-17 * max(|origin|, scene
// radius) (64 ulp of the largest coordinate involved), and a box is skipped
// only if its padded entry distance exceeds the best t so far, never on
// equality, so ties in other subtrees are still visited.
//
// EXACT adds the one case a tree cannot see: a ray exactly parallel to a
// triangle's plane (dot(n,d) == 0 in f32) has st = +inf in the reference,
// passes every comparison there (frag:405-411 are all false on inf/NaN) and is
// reported as a hit at t = inf when nothing else is hit.  Such a hit carries
// no radiance (every later value is NaN and the sample is dropped), so the
// render loop normally treats it as the miss it geometrically is; the
// hit-buffer readbacks run the extra pass so that they equal the reference's
// buffer, and so does a render loop that has to produce the reference's
// non-finite samples one for one (the 8-bit accumulator emulation, the exact
// count of such samples): the hit at infinity goes to the shade kernel like
// any other, where everything computed from it is NaN.
//
// Scheduling inside a wave.  A wave owns one region of rays at a time and keeps
// its 64 lanes fed from it: lanes whose ray has finished take the next unread
// rays of the region (ballot + prefix rank, no atomics) once enough of them
// are idle.  Traversal alternates two phases so that lanes do the same kind of
// work together: (A) every lane that stands on an inner node descends until
// it reaches a leaf or runs out of nodes, (B) the lanes standing on a leaf
// test its triangles.
//
// BLOCK: threads per block.  The block shares one LDS copy of the scene, and a CU holds 16 waves of this kernel
// either way: as four blocks of 256 threads, each with a copy that has to fit in 40 KB with its stacks, or as
// one block of 1024 threads with the whole 160 KB to itself -- which is what keeps a scene of a thousand
// triangles (Veggie.obj: 810 nodes) entirely in LDS (+36 % paths/s over caching the top of its tree).
// SCENE_LDS: 0 = nodes and triangles are read from global memory (L1/L2);
// 1 = the block keeps a copy of the first n_lds_nodes nodes in LDS (the host
// numbers the top of the tree breadth-first, so any prefix is the part every
// ray walks through) and reads deeper nodes from global memory; 3 = the whole
// node array; 2 = all nodes and all triangles.  The host picks the variant by
// the LDS a block may use without costing occupancy.
//
// RENDER = false (readbacks): every ray's (t, hit code) goes to results[slot].
// RENDER = true (the render loop): the kernel also ends the paths that end
// here -- a ray that reaches a light writes its sample (frag:878-882), a ray
// that escapes or sits on the last allowed segment adds nothing -- and appends
// the remaining surface hits, compacted per region with ballot + prefix rank,
// to the hit list the shade kernel consumes: (t, triangle, slot in region).
//
// The body is a device function so that it can also run inside k_bounces (all bounces of a batch in one launch);
// fill_lds: copy the scene to LDS first (k_bounces does that once, not per bounce).
template <int BLOCK, bool COUNT, bool EXACT, int SCENE_LDS, bool RENDER, bool USE_LT>
__device__ __forceinline__ void extend_body(const SceneView &sc, const Regions &rg, const PathQueue &q, float4 *hit_list,
                                            float4 *samples, uint32_t last_segment, const TraversalCounters &tc, uint32_t n_nodes,
                                            uint32_t refill_at, uint32_t stack_entries, uint32_t n_lds_nodes, uint32_t lds_stride,
                                            const BatchParams *bp, uint32_t lds_stack_entries,
                                            int *stack_overflow, bool fill_lds, WaveTurns turns, uint32_t n_lds_tris)
{
    const uint32_t sample_tag = RENDER ? bp->sample_tag : 0u;
    // lds_stride: 16-byte lanes between consecutive triangle records of the LDS copy of the scene.  4 = packed: the four
    // ds_read_b128 of a 64-byte record all start on banks 0, 16, 32 or 48, so the 16 lanes such a read serves at
    // a time collide four deep on average.  5 spreads consecutive records over all banks (records of 80 bytes,
    // the last lane unused); the host asks for it when the part of the scene kept in LDS still fits that way.
    //
    // LDS, from address 0: the nodes, the triangles, the stacks.
    // When ALL nodes are in LDS (SCENE_LDS >= 2) they are kept in a layout of their own, 112 bytes per node:
    //   floats 8a + {0..3} = (lo0, hi0, lo1, hi1) of the two children's boxes on axis a, floats 8a + {4..7} =
    //   (hi0, lo0, hi1, lo1);  ints 24, 25 = the child codes, an inner child's code being the LDS address of its record.
    // A ray reads (entry plane, exit plane) of both children on an axis with one ds_read_b128, from the first
    // quadruple when it travels in the positive direction and from the second when in the negative one: the
    // planes arrive in the right order without a select per plane (12 per node step otherwise), in four LDS
    // instructions per node, and following a child takes no address arithmetic.  (Records of an odd number of
    // 16-byte lanes also spread over the banks.)
    // Otherwise (SCENE_LDS == 1: the first n_lds_nodes nodes) the 64-byte records are copied as they are.
    constexpr bool kWide = SCENE_LDS == 4;  // four-wide nodes read from memory (sc.wide_nodes), nothing of the tree in LDS
    constexpr bool kNodeAddresses = SCENE_LDS == 2 || SCENE_LDS == 3;
    constexpr uint32_t kLdsNodeBytes = kNodeAddresses ? 112u : 64u;
    float2 *__restrict__ results = reinterpret_cast<float2 *>(hit_list);
    extern __shared__ float4 lds[];
    char *const lds_bytes = reinterpret_cast<char *>(lds);
    float4 *lds_nodes = lds;                                                        // the first n_lds_nodes nodes
    // n_lds_tris: triangle records kept in LDS behind the nodes: all of them (SCENE_LDS == 2) or, when only the
    // nodes fit with room to spare (SCENE_LDS == 3), the first n_lds_tris of the leaf-ordered array -- which the
    // host has ordered so that the leaves with the largest boxes, the ones most rays come to, are at its start.
    float4 *lds_tris = lds + (kLdsNodeBytes / 16u) * static_cast<size_t>(n_lds_nodes);
    const uint32_t lds_rec_bytes = 16u * lds_stride;
    const uint32_t tid = threadIdx.x;
    // stacks: [entry][lane]; a lane's entry e lies at stack0 + e * kStackStep.
    // A tree read from memory (SCENE_LDS <= 1) is deep, and its traversal waits for memory, which only more
    // resident waves hide: such a launch keeps just the first lds_stack_entries entries of every stack in LDS
    // -- a near-first descent rarely has more children deferred -- and the entries above them in
    // stack_overflow[entry - lds_stack_entries][lane of the grid], so that more blocks fit on a CU.
    constexpr uint32_t kStackStep = BLOCK * sizeof(StackT);
    constexpr bool kDeepStack = SCENE_LDS <= 1 || kWide;
    const uint32_t lds_base = lds_address(lds_bytes);
    const uint32_t stack0 = lds_base + kLdsNodeBytes * n_lds_nodes + (kNodeAddresses ? lds_rec_bytes * n_lds_tris : 0u) +
                            tid * static_cast<uint32_t>(sizeof(StackT));  // LDS address of the lane's entry 0
    if (SCENE_LDS >= 1 && !kWide && fill_lds) {
        if (kNodeAddresses) {
            for (uint32_t i = tid; i < n_lds_nodes; i += BLOCK) {
                const float4 n0 = sc.nodes[4 * i], n1 = sc.nodes[4 * i + 1], n2 = sc.nodes[4 * i + 2], n3 = sc.nodes[4 * i + 3];
                float4 *w = reinterpret_cast<float4 *>(lds_bytes + kLdsNodeBytes * i);
                // child 0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child 1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w)
                w[0] = make_float4(n0.x, n0.w, n1.z, n2.y), w[1] = make_float4(n0.w, n0.x, n2.y, n1.z);
                w[2] = make_float4(n0.y, n1.x, n1.w, n2.z), w[3] = make_float4(n1.x, n0.y, n2.z, n1.w);
                w[4] = make_float4(n0.z, n1.y, n2.x, n2.w), w[5] = make_float4(n1.y, n0.z, n2.w, n2.x);
                const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                w[6] = make_float4(__int_as_float(c0 >= 0 ? static_cast<int>(lds_base + kLdsNodeBytes * static_cast<uint32_t>(c0)) : c0),
                                   __int_as_float(c1 >= 0 ? static_cast<int>(lds_base + kLdsNodeBytes * static_cast<uint32_t>(c1)) : c1), 0.f, 0.f);
            }
        } else {
            for (uint32_t i = tid; i < 4u * n_lds_nodes; i += BLOCK) lds_nodes[i] = sc.nodes[i];
        }
        if (kNodeAddresses)
            for (uint32_t i = tid; i < 4u * n_lds_tris; i += BLOCK)
                lds_tris[(i >> 2) * lds_stride + (i & 3u)] = sc.triangles[i];
        __syncthreads();
    }
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    unsigned long long my_nodes = 0, my_tris = 0;
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-level diagnostics (COUNT only)
    const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;

    // The wave streams through its regions: when the current one has handed out
    // all its rays, idle lanes start on the next one while the last rays of the
    // previous region are still in flight (at most two regions at a time, told
    // apart by one parity bit per lane), so only the very end of the kernel
    // runs with thinning lanes.
    uint32_t k = blockIdx.x * (BLOCK / 64) + wave_in_block();  // next turn of this wave
    uint32_t cur_r = 0, cur_n = 0, cur_base = 0, cur_next = 0, cur_h0 = 0, cur_h1 = 0;  // region being handed out
    uint32_t prv_r = 0, prv_base = 0, prv_h0 = 0, prv_h1 = 0;                           // region before it
    bool cur_valid = false, prv_valid = false;
    bool cur_gen = false;  // parity of the current region; a lane whose gen differs works for the previous one
    uint32_t nxt_r = turns.r;                                          // region of turn k and its ray count
    uint32_t nxt_n = k < rg.n_regions ? load_count(rg.counts_in + nxt_r) : 0u;
    {
        // per-lane ray state
        // A lane's traversal state is `node` (>= 0: an inner node to test -- its index, or with kNodeAddresses the
        // LDS address of its record; < 0: a leaf code to test; kFinished: nothing left) and `sp`, the LDS address
        // of the first free entry of its stack.  Entry 0 of every stack holds kFinished, so the pop that finds
        // nothing deferred yields the finished state without a test.  A lane without a ray stands on kIdle: whether
        // a lane is busy is then a compare on a register the loops test anyway, not a flag of its own.
        bool gen = false;
        uint32_t slot = 0;
        f3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f);
        float ix = 0.f, iy = 0.f, iz = 0.f, cnx = 0.f, cny = 0.f, cnz = 0.f, cfx = 0.f, cfy = 0.f, cfz = 0.f;
        bool negx = false, negy = false, negz = false;
        uint32_t offx = 0, offy = 0, offz = 0;  // kNodeAddresses, kWide: 16 bytes on the axes along which the ray travels backwards
        Hit h{__builtin_inff(), -1};
        float ltv = __uint_as_float(kNoLightBits);  // the word beside the ray: |ltv| = distance to the light or NaN, sign = poisoned path
        int node = kIdle;
        uint32_t sp = stack0 + kStackStep;  // LDS address of the lane's first free stack entry
        lds_store<StackT>(stack0, static_cast<StackT>(kFinished));
        const uint32_t sp_lds_end = stack0 + lds_stack_entries * kStackStep;  // addresses from here on stand for overflow entries
        const size_t grid_lanes = static_cast<size_t>(gridDim.x) * BLOCK, grid_lane = static_cast<size_t>(blockIdx.x) * BLOCK + tid;
        auto push = [&](int v) {
            if (kDeepStack && sp >= sp_lds_end)
                stack_overflow[((sp - sp_lds_end) / kStackStep) * grid_lanes + grid_lane] = v;
            else
                lds_store<StackT>(sp, static_cast<StackT>(v));
            sp += kStackStep;
        };
        auto pop = [&]() -> int {
            sp -= kStackStep;
            if (kDeepStack && sp >= sp_lds_end) return stack_overflow[((sp - sp_lds_end) / kStackStep) * grid_lanes + grid_lane];
            return lds_load<StackT>(sp);
        };

        for (;;) {
            // ---- hand fresh rays to idle lanes
            const unsigned long long idle = __ballot(node == kIdle);
            const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
            if (n_idle >= refill_at || n_idle == 64u) {
                // Current region dry: once the previous one has no ray in flight, publish it and open the next.
                while (cur_next >= cur_n && k < rg.n_regions && __ballot(node != kIdle && gen != cur_gen) == 0ull) {
                    if (RENDER && prv_valid && (tid & 63u) == 0) rg.counts_out[prv_r] = prv_h0 | (prv_h1 << 16);
                    prv_r = cur_r, prv_base = cur_base, prv_h0 = cur_h0, prv_h1 = cur_h1, prv_valid = cur_valid;
                    cur_r = nxt_r;
                    cur_n = nxt_n;
                    cur_base = cur_r * rg.size;
                    cur_next = 0;
                    cur_h0 = cur_h1 = 0;
                    cur_valid = true;
                    cur_gen = !cur_gen;
                    k += n_waves;
                    // the count of the region after this one is fetched now, long before it is needed
                    nxt_r = next_region(rg, nxt_r, turns.step);
                    nxt_n = k < rg.n_regions ? load_count(rg.counts_in + nxt_r) : 0u;
                }
            }
            if (cur_next < cur_n && (n_idle >= refill_at || n_idle == 64u)) {
                if (node == kIdle) {
                    const uint32_t my = cur_next + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
                    if (my < cur_n) {
                        gen = cur_gen;
                        slot = cur_base + my;
                        const float4 a = q.p0[slot], b = q.p1[slot];
                        o = mk(a.x, a.y, a.z);
                        d = mk(a.w, b.x, b.y);
                        // Reciprocal direction, kept within +-1e30 so that no product below is inf * 0 (a zero
                        // component gives +-inf with its sign, v_med3 brings it back).  The hardware reciprocal
                        // (1 ulp) is enough: the slabs are padded by far more.
                        ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f);
                        iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f);
                        iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f);
                        const float pad = 7.62939453125e-6f *
                                          fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), sc.scene_radius));
                        const float mx = pad * fabsf(ix), my_ = pad * fabsf(iy), mz = pad * fabsf(iz);
                        const float tx = o.x * ix, ty = o.y * iy, tz = o.z * iz;
                        cnx = -tx - mx;  // entry planes
                        cny = -ty - my_;
                        cnz = -tz - mz;
                        cfx = mx - tx;  // exit planes
                        cfy = my_ - ty;
                        cfz = mz - tz;
                        negx = ix < 0.f;
                        negy = iy < 0.f;
                        negz = iz < 0.f;
                        offx = negx ? 16u : 0u;
                        offy = negy ? 16u : 0u;
                        offz = negz ? 16u : 0u;
                        // one-light scenes: the light's distance, known since the ray was made, is the limit
                        // the traversal starts with (a surface beyond the light cannot change the outcome)
                        ltv = (USE_LT || RENDER) ? q.lt[slot] : __uint_as_float(kNoLightBits);
                        h.t = (USE_LT && fabsf(ltv) >= 0.f) ? fabsf(ltv) : __builtin_inff();
                        h.id = -1;
                        node = kNodeAddresses ? static_cast<int>(lds_base) : 0;  // the root
                        sp = stack0 + kStackStep;
                    }
                }
                if (COUNT) {
                }
                cur_next += n_idle;
            }
            if (__ballot(node != kIdle) == 0ull) {
                if (cur_next >= cur_n && k >= rg.n_regions) break;  // nothing in flight and nothing left to hand out
                continue;
            }
            if (COUNT) {
                dg[4] += 1;
                dg[5] += __popcll(__ballot(node != kIdle));
            }

            // ---- phase A: inner nodes
            // (the limit does not change while lanes descend; quieting it here keeps that out of the loop)
            const float t_limit = __builtin_canonicalizef(h.t);
            // One step of a lane that stands on an inner node (go): its children's boxes, the nearer one next, the
            // other deferred, or the last deferred child taken up again.
            auto node_step = [&](const bool go) {
                if (kWide) {
                  if (go) {
                    // One 128-byte record = the four nodes two levels down: per axis the low planes of the four in
                    // one 16-byte lane and the high planes in the next, so a ray reads its entry planes from the
                    // lane at offset (direction negative ? 16 : 0) and its exit planes from the other one -- no
                    // select per plane; the record is 128-byte aligned, so the offset goes into the address with an OR.
                    const uintptr_t nb = reinterpret_cast<uintptr_t>(sc.wide_nodes) + (static_cast<uintptr_t>(static_cast<uint32_t>(node)) << 7);
                    const float4 ex = *reinterpret_cast<const float4 *>(nb | offx), xx = *reinterpret_cast<const float4 *>(nb | (offx ^ 16u));
                    const float4 ey = *reinterpret_cast<const float4 *>((nb | offy) + 32u), xy = *reinterpret_cast<const float4 *>((nb | (offy ^ 16u)) + 32u);
                    const float4 ez = *reinterpret_cast<const float4 *>((nb | offz) + 64u), xz = *reinterpret_cast<const float4 *>((nb | (offz ^ 16u)) + 64u);
                    const int4 cc = *reinterpret_cast<const int4 *>(nb + 96u);
                    if (COUNT) ++my_nodes;
                    // Keys: a child that is hit gets the bits of its entry distance (>= 0, so they order like the
                    // distances) with the child's slot in the two lowest bits, a child that is missed all ones.
                    // Sorted ascending, the nearest hit comes first and the misses last.
#define PT_WIDE_KEY(comp, slot)                                                                                               \
    ([&]() -> uint32_t {                                                                                                      \
        const float tn = fmaxf(fmaxf(fmaf(ex.comp, ix, cnx), fmaf(ey.comp, iy, cny)), fmaxf(fmaf(ez.comp, iz, cnz), 0.f));    \
        const float tf = fminf(fminf(fmaf(xx.comp, ix, cfx), fmaf(xy.comp, iy, cfy)), fminf(fmaf(xz.comp, iz, cfz), t_limit)); \
        return tn <= tf ? ((__float_as_uint(tn) & ~3u) | slot) : 0xFFFFFFFFu;                                                 \
    }())
                    uint32_t k0 = PT_WIDE_KEY(x, 0u), k1 = PT_WIDE_KEY(y, 1u), k2 = PT_WIDE_KEY(z, 2u), k3 = PT_WIDE_KEY(w, 3u);
#undef PT_WIDE_KEY
                    auto order = [](uint32_t &a, uint32_t &b) {
                        const uint32_t lo = min(a, b), hi = max(a, b);
                        a = lo, b = hi;
                    };
                    order(k0, k1), order(k2, k3), order(k0, k2), order(k1, k3), order(k1, k2);
                    auto code_of = [&](uint32_t key) {
                        const uint32_t s = key & 3u;
                        return s == 0u ? cc.x : (s == 1u ? cc.y : (s == 2u ? cc.z : cc.w));
                    };
                    // the nearest child is visited next, the others are deferred farthest first (so that the
                    // nearest of them is the first to be taken up again); with no child hit the lane takes up
                    // its last deferred child (or finds the end marker)
                    if (k3 != 0xFFFFFFFFu) push(code_of(k3));
                    if (k2 != 0xFFFFFFFFu) push(code_of(k2));
                    if (k1 != 0xFFFFFFFFu) push(code_of(k1));
                    node = k0 != 0xFFFFFFFFu ? code_of(k0) : pop();
                  }
                } else if (go) {
                    float e0x, x0x, e0y, x0y, e0z, x0z, e1x, x1x, e1y, x1y, e1z, x1z;
                    int c0, c1;
                    if (kNodeAddresses) {
                        const uint32_t nb = static_cast<uint32_t>(node);
                        const LdsFloat4 ax = lds_load<LdsFloat4>(nb + offx), ay = lds_load<LdsFloat4>(nb + offy + 32u),
                                        az = lds_load<LdsFloat4>(nb + offz + 64u);
                        e0x = ax.x, x0x = ax.y, e1x = ax.z, x1x = ax.w;
                        e0y = ay.x, x0y = ay.y, e1y = ay.z, x1y = ay.w;
                        e0z = az.x, x0z = az.y, e1z = az.z, x1z = az.w;
                        const LdsInt2 cc = lds_load<LdsInt2>(nb + 96u);
                        c0 = cc.x, c1 = cc.y;
                    } else {
                        float4 n0, n1, n2, n3;
                        if (SCENE_LDS == 1 && static_cast<uint32_t>(node) < n_lds_nodes) {
                            const float4 *np = lds_nodes + 4u * static_cast<uint32_t>(node);
                            n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                        } else {
                            const float4 *np = sc.nodes + 4 * static_cast<size_t>(node);
                            n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                        }
                        // child 0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child 1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w)
                        e0x = negx ? n0.w : n0.x, x0x = negx ? n0.x : n0.w;
                        e0y = negy ? n1.x : n0.y, x0y = negy ? n0.y : n1.x;
                        e0z = negz ? n1.y : n0.z, x0z = negz ? n0.z : n1.y;
                        e1x = negx ? n2.y : n1.z, x1x = negx ? n1.z : n2.y;
                        e1y = negy ? n2.z : n1.w, x1y = negy ? n1.w : n2.z;
                        e1z = negz ? n2.w : n2.x, x1z = negz ? n2.x : n2.w;
                        c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                    }
                    if (COUNT) ++my_nodes;
                    const float tn0 = fmaxf(fmaxf(fmaf(e0x, ix, cnx), fmaf(e0y, iy, cny)), fmaxf(fmaf(e0z, iz, cnz), 0.f));
                    const float tf0 = fminf(fminf(fmaf(x0x, ix, cfx), fmaf(x0y, iy, cfy)), fminf(fmaf(x0z, iz, cfz), t_limit));
                    const float tn1 = fmaxf(fmaxf(fmaf(e1x, ix, cnx), fmaf(e1y, iy, cny)), fmaxf(fmaf(e1z, iz, cnz), 0.f));
                    const float tf1 = fminf(fminf(fmaf(x1x, ix, cfx), fmaf(x1y, iy, cfy)), fminf(fmaf(x1z, iz, cfz), t_limit));
                    const bool hit0 = tn0 <= tf0, hit1 = tn1 <= tf1;
                    // The nearer of the children that are hit is visited next, the other one deferred; with no
                    // child hit the lane takes up its last deferred child (or finds the end marker).
                    const bool second_first = hit1 && !(hit0 && tn0 <= tn1);
                    const int near_c = second_first ? c1 : c0, far_c = second_first ? c0 : c1;
                    if (hit0 && hit1) push(far_c);
                    node = near_c;
                    if (!(hit0 || hit1)) node = pop();
                }
            };
            for (;;) {
                const bool go = node >= 0;
                const unsigned long long going = __ballot(go);
                if (going == 0ull) break;
                if (COUNT) {
                    dg[0] += 1;
                    dg[1] += __popcll(going);
                }
                node_step(go);
                // further steps right away for the lanes still on an inner node: the loop's own bookkeeping (ballot,
                // exit test, the masks of lanes that have left) is scalar work that is paid per iteration
#pragma unroll
                for (int u = 0; u < PT_PHASE_A_UNROLL; ++u)
                    if (!COUNT) node_step(node >= 0);
            }

            // ---- phase B: leaves
            if (COUNT) {
                const uint32_t c = (node > kIdle) ? ((~static_cast<uint32_t>(node)) & 7u) : 0u;
                uint32_t mx = c;
                for (int off = 32; off > 0; off >>= 1) mx = max(mx, static_cast<uint32_t>(__shfl_xor(mx, off)));
                uint32_t sm = c;
                for (int off = 32; off > 0; off >>= 1) sm += static_cast<uint32_t>(__shfl_xor(sm, off));
                dg[2] += mx;
                dg[3] += sm;
            }
            if (node > kIdle) {  // (no lane stands on an inner node here: what is left above kIdle is a leaf code)
                const uint32_t code = ~static_cast<uint32_t>(node);
                const uint32_t first = code >> 3, cnt = code & 7u;
                const char *lds_tri = reinterpret_cast<const char *>(lds_tris) + __umul24(first, lds_rec_bytes);  // (SCENE_LDS == 2)
                for (uint32_t j = 0; j < cnt; ++j) {
                    if (SCENE_LDS == 2) {
                        test_triangle<false>(reinterpret_cast<const float4 *>(lds_tri), o, d, h);
                        lds_tri += lds_rec_bytes;
                    } else if (SCENE_LDS == 3) {
                        // a record from LDS or from memory, lane by lane: one flat load serves both
                        const float4 *in_lds = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(lds_tris) + __umul24(first + j, lds_rec_bytes));
                        const float4 *in_mem = sc.triangles + 4 * static_cast<size_t>(first + j);
                        test_triangle<false>(first + j < n_lds_tris ? in_lds : in_mem, o, d, h);
                    } else {
                        test_triangle<false>(sc.triangles + 4 * static_cast<size_t>(first + j), o, d, h);
                    }
                }
                if (COUNT) my_tris += cnt;
                node = pop();
            }

            // ---- finished rays: lights, result, lane becomes idle
            bool append = false;
            float hit_t = 0.f;
            int hit_code = -1;
            bool other = false;
            if (node == kFinished) {
                if (EXACT && !(h.t < __builtin_inff())) {
                    for (uint32_t j = 0; j < sc.n_triangles; ++j) {
                        const float4 *tp = sc.triangles + 4 * static_cast<size_t>(j);
                        const f3 nn = mk(tp[1].w, tp[2].w, tp[3].x);
                        if (dotm(nn, d) == 0.f) test_triangle<true>(tp, o, d, h);
                    }
                }
                hit_t = h.t;
                hit_code = h.id < 0 ? h.id : (h.id & kIdMask);
                other = h.id >= 0 && (h.id & kClassBit) != 0;
                if (USE_LT) {
                    // the light wins unless a surface lies strictly in front of it (frag:427: reject if tl > t)
                    if (fabsf(ltv) >= 0.f && !(fabsf(ltv) > h.t)) hit_code = -2;
                } else {
                    test_lights(sc, o, d, hit_t, hit_code);
                }
                if (!RENDER) {
                    results[slot] = make_float2(hit_t, __int_as_float(hit_code));
                } else if (hit_code <= -2) {
                    // passCol = attenuation * intensity (frag:878-882, :901)
                    const float4 c = q.p2[slot];
                    // (with the light's distance carried beside the ray there is exactly one light: a scalar load)
                    const float *rad = USE_LT ? sc.lights[0].radiance : sc.lights[-(hit_code + 2)].radiance;
                    const f3 L = mk(c.x * rad[0], c.y * rad[1], c.z * rad[2]);
                    // a non-finite sample is left to the accumulate kernel (which drops and counts it)
                    samples[__float_as_uint(c.w)] = make_float4(L.x, L.y, L.z, __uint_as_float(sample_tag));
                } else {
                    append = hit_code >= 0 && last_segment == 0u;
                    // a poisoned path that escapes, or whose last allowed segment ends on a surface
                    if (!append && (__float_as_uint(ltv) & kPoisonBit)) write_poisoned_sample(samples, q.p2[slot], sample_tag);
                }
                node = kIdle;
            }
            if (RENDER) {
                const bool for_prev = gen != cur_gen;
                append_hit(hit_list, cur_base, rg.size, append && !for_prev, other,
                           make_float4(hit_t, __int_as_float(hit_code), __uint_as_float(slot - cur_base), 0.f), cur_h0, cur_h1);
                if (__ballot(append && for_prev) != 0ull)
                    append_hit(hit_list, prv_base, rg.size, append && for_prev, other,
                               make_float4(hit_t, __int_as_float(hit_code), __uint_as_float(slot - prv_base), 0.f), prv_h0, prv_h1);
            }
        }
        if (RENDER && (tid & 63u) == 0) {
            if (prv_valid) rg.counts_out[prv_r] = prv_h0 | (prv_h1 << 16);
            if (cur_valid) rg.counts_out[cur_r] = cur_h0 | (cur_h1 << 16);
        }
    }
    if (COUNT) {
        atomicAdd(tc.nodes_visited, my_nodes);
        atomicAdd(tc.tris_tested, my_tris);
        if (tc.diag && (tid & 63u) == 0) {
            for (int j = 0; j < 6; ++j) atomicAdd(tc.diag + j, dg[j]);
            // [6] sum over waves of the time each was busy, [7] sum over launches of the longest (100 MHz ticks)
            const unsigned long long dt = wall_clock64() - t_start;
            atomicAdd(tc.diag + 6, dt);
#ifndef PT_DIAG_BLOCK
            atomicMax(tc.diag + 7, dt);
#endif
        }
#ifdef PT_DIAG_BLOCK
        // experiment: [7] = sum over blocks of the longest wave's time in the block (the slot is the start of the
        // stack area, free once every wave of the block has finished)
        const unsigned long long dt_wave = wall_clock64() - t_start;
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(
            lds_bytes + kLdsNodeBytes * n_lds_nodes + (kNodeAddresses ? lds_rec_bytes * n_lds_tris : 0u));
        __syncthreads();
        if (tid == 0) *slot = 0ull;
        __syncthreads();
        if (tc.diag && (tid & 63u) == 0) atomicMax(slot, dt_wave);
        __syncthreads();
        if (tc.diag && tid == 0) atomicAdd(tc.diag + 7, *slot);
#endif
    }
}

// (Four-wide nodes are for trees read from memory, whose traversal waits for node fetches: six waves per SIMD
// hide more of that than five, so that instantiation is held to the 80 registers six waves leave each.)
#ifndef PT_WIDE_WAVES
#define PT_WIDE_WAVES 6
#endif
template <int BLOCK, bool COUNT, bool EXACT, int SCENE_LDS, bool RENDER, bool USE_LT>
__global__ void __launch_bounds__(BLOCK, SCENE_LDS == 4 ? PT_WIDE_WAVES : PT_EXTEND_WAVES) k_extend_bvh(SceneView sc, Regions rg, PathQueue q,
                                                        float4 *__restrict__ hit_list, float4 *__restrict__ samples,
                                                        uint32_t last_segment,
                                                        TraversalCounters tc, uint32_t n_nodes, uint32_t refill_at,
                                                        uint32_t stack_entries, uint32_t n_lds_nodes, uint32_t lds_stride,
                                                        const BatchParams *__restrict__ bp,
                                                        uint32_t lds_stack_entries, int *__restrict__ stack_overflow, uint32_t n_lds_tris)
{
    const WaveTurns turns = wave_turns(rg, blockIdx.x * (BLOCK / 64) + wave_in_block(), gridDim.x * (BLOCK / 64));
    extend_body<BLOCK, COUNT, EXACT, SCENE_LDS, RENDER, USE_LT>(sc, rg, q, hit_list, samples, last_segment, tc, n_nodes, refill_at,
                                                                stack_entries, n_lds_nodes, lds_stride, bp,
                                                                lds_stack_entries, stack_overflow, true, turns, n_lds_tris);
}

// ----------------------------------------------------------------- shade

// Arithmetic of the shading stage.  Division, square root, sine and cosine go through these helpers.  The
// shader's own sin / cos / pow / inversesqrt are whatever the GL driver makes of them (implementation-defined
// precision in GLSL), so nothing past the first hit can be compared with the reference bit for bit; what the
// reference's images can tell apart is decided on them (1920x1080 and 256x256 float images of seven
// scene / depth cases, the 8-bit accumulation texture).  Measured on MI355X, per build:
//   correctly rounded everything (PT_SHADE_FAST_MATH=0 PT_FAST_DIV=0 PT_FAST_SQRT=0): every reference image case
//     green; against the CPU restatement of the shader 4 of 32 891 random BSDF samples differ by more than 2e-3;
//   hardware reciprocal and square root (1 ulp), libm sine / cosine -- THE DEFAULT: every reference image case
//     green (per-pixel relMSE against the reference at 1080p 7e-6, the noise floor being 0.61); 89 such samples --
//     ill-conditioned ones (grazing microfacet normals, refraction near total reflection) where an ulp is
//     amplified; 40 % fewer VALU instructions in the shading code, +1.5 % (Box) to +2 % (Veggie) paths/s;
//   native sine / cosine as well (PT_FAST_SINCOS=1): 99 such samples, the same speed, and the 8-bit emulation then
//     differs from the CPU restatement's by 3/255 in a texel where 2/255 is the bound: not taken.
// The ray generation, the intersection tests and the light test never use these helpers.
#ifndef PT_SHADE_FAST_MATH
#define PT_SHADE_FAST_MATH 0
#endif
#ifndef PT_FAST_DIV
#define PT_FAST_DIV 1
#endif
#ifndef PT_FAST_SQRT
#define PT_FAST_SQRT 1
#endif
#ifndef PT_FAST_SINCOS
#define PT_FAST_SINCOS PT_SHADE_FAST_MATH
#endif
__device__ __forceinline__ float sdiv(float a, float b)
{
#if PT_FAST_DIV
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
__device__ __forceinline__ float ssqrt(float x)
{
#if PT_FAST_SQRT
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
__device__ __forceinline__ f3 snormalize(f3 v) { return v * sdiv(1.f, ssqrt(dotm(v, v))); }
__device__ __forceinline__ void ssincos(float x, float *s, float *c)
{
#if PT_FAST_SINCOS
    *s = __sinf(x);
    *c = __cosf(x);
#else
    sincosf(x, s, c);
#endif
}

struct Material {
    f3 albedo;
    float roughness, metallic, ior, anisotropy, transmission;
};

struct Frame {
    f3 t, b, n;  // mat3(tangent, bitangent, normal) columns
};
__device__ __forceinline__ f3 to_world(const Frame &f, f3 l) { return (f.t * l.x + f.b * l.y) + f.n * l.z; }
__device__ __forceinline__ f3 to_local(const Frame &f, f3 w) { return mk(dotm(f.t, w), dotm(f.b, w), dotm(f.n, w)); }

// frag:154-165, :176-181
__device__ __forceinline__ Frame frame_from_normal(f3 n)
{
    Frame f;
    f.n = n;
    if (fabsf(n.x) > fabsf(n.y)) {
        const float il = sdiv(1.f, ssqrt(n.x * n.x + n.z * n.z));
        f.t = mk(-n.z * il, 0.f * il, n.x * il);
    } else {
        const float il = sdiv(1.f, ssqrt(n.y * n.y + n.z * n.z));
        f.t = mk(0.f * il, n.z * il, -n.y * il);
    }
    f.b = crossm(n, f.t);
    return f;
}
// frag:169-174 with the shader's fixed tangent (0,0,1) (frag:816): a normal
// parallel to z yields a NaN frame there, and so it does here.
__device__ __forceinline__ Frame frame_from_normal_fixed_tangent(f3 n)
{
    Frame f;
    f.b = snormalize(crossm(n, mk(0.f, 0.f, 1.f)));
    f.t = crossm(f.b, n);
    f.n = n;
    return f;
}

// frag:197-249
__device__ __forceinline__ float cos2_theta(f3 v) { return v.z * v.z; }
__device__ __forceinline__ float sin2_theta(f3 v) { return fmaxf(0.f, 1.f - cos2_theta(v)); }
__device__ __forceinline__ float tan2_theta(f3 v) { return sdiv(sin2_theta(v), cos2_theta(v)); }
__device__ __forceinline__ float clamp11(float x) { return fminf(fmaxf(x, -1.f), 1.f); }
__device__ __forceinline__ float cos2_phi(f3 v)
{
    const float s = ssqrt(sin2_theta(v));
    const float c = (s == 0.f) ? 1.f : clamp11(sdiv(v.x, s));
    return c * c;
}
__device__ __forceinline__ float sin2_phi(f3 v)
{
    const float s = ssqrt(sin2_theta(v));
    const float c = (s == 0.f) ? 0.f : clamp11(sdiv(v.y, s));
    return c * c;
}

// frag:255-290
__device__ __forceinline__ f3 square_to_hemisphere_cosine(float xix, float xiy)
{
    const float ux = xix * 2.f - 1.f, uy = xiy * 2.f - 1.f;
    const float x2 = ux * ux, y2 = uy * uy;
    float pr = 0.f, pa = 0.f;
    if (x2 > y2) {
        pr = ux;
        pa = sdiv((kPi / 4.f) * uy, ux);
    } else if (y2 > 0.f) {
        pr = uy;
        pa = (kPi / 2.f) - sdiv((kPi / 4.f) * ux, uy);
    }
    float s, c;
    ssincos(pa, &s, &c);
    f3 r;
    r.x = c * pr;
    r.y = s * pr;
    r.z = ssqrt(fmaxf(0.f, 1.f - r.x * r.x - r.y * r.y));
    return r;
}

__device__ __forceinline__ float pow5(float x)
{
    const float x2 = x * x;
    return x2 * x2 * x;
}

// frag:607-634
__device__ __forceinline__ float tr_distribution(f3 m, float ax, float ay)
{
    const float t2 = tan2_theta(m);
    if (isinf(t2)) return 0.f;
    const float c2 = cos2_theta(m);
    const float c4 = c2 * c2;
    const float e = (sdiv(cos2_phi(m), ax * ax) + sdiv(sin2_phi(m), ay * ay)) * t2;
    return sdiv(1.f, kPi * ax * ay * c4 * (1.f + e) * (1.f + e));
}
__device__ __forceinline__ float tr_lambda(f3 v, float ax, float ay)
{
    const float t2 = tan2_theta(v);
    if (isinf(t2)) return 0.f;
    const float a2 = ax * ax * cos2_phi(v) + ay * ay * sin2_phi(v);
    return (ssqrt(1.f + a2 * t2) - 1.f) * 0.5f;
}
__device__ __forceinline__ float tr_masking(f3 wo, f3 wi, float ax, float ay)
{
    return sdiv(1.f, 1.f + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay));
}
// frag:660-665
__device__ __forceinline__ float tr_density(f3 wo, f3 m, float ax, float ay)
{
    const float g1 = sdiv(1.f, 1.f + tr_lambda(wo, ax, ay));
    return sdiv(g1 * tr_distribution(m, ax, ay) * fabsf(dotm(wo, m)), fabsf(wo.z));
}
// frag:636-658: visible-normal sampling; the warp of the disk sample keeps the
// shader's mix((1 - h.z)/2, sqrt(1 - px^2), py).
__device__ __forceinline__ f3 tr_sample_normal(f3 wo, float xix, float xiy, float ax, float ay)
{
    f3 h = snormalize(mk(ax * wo.x, ay * wo.y, wo.z));
    if (h.z < 0.f) h = -h;
    const f3 T1 = (h.z < 0.99999f) ? snormalize(crossm(mk(0.f, 0.f, 1.f), h)) : mk(1.f, 0.f, 0.f);
    const f3 T2 = crossm(h, T1);
    const float r = ssqrt(xix);
    float s, c;
    ssincos(2.f * kPi * xiy, &s, &c);
    const float px = r * c;
    float py = r * s;
    const float hh = ssqrt(1.f - px * px);
    const float lo = (1.f - h.z) / 2.f;
    py = lo * (1.f - py) + hh * py;
    const float pz = ssqrt(fmaxf(0.f, 1.f - (px * px + py * py)));
    const f3 nh = (T1 * px + T2 * py) + h * pz;
    return snormalize(mk(ax * nh.x, ay * nh.y, fmaxf(0.000001f, nh.z)));
}
__device__ __forceinline__ f3 reflectm(f3 I, f3 N)
{
    const float k = 2.f * dotm(N, I);
    return I - N * k;
}

// One BxDF sample.  ok == false means the lobe returned vec3(0) without
// writing a direction (frag:729,753,774,818,827): the path ends with no
// contribution.
struct Bxdf {
    f3 f, wi;
    float pdf;
    bool ok;
};

// frag:706-721
__device__ __forceinline__ Bxdf diffuse_bxdf(f3 n, const Material &mat, float xix, float xiy, f3 wo)
{
    Bxdf r;
    if (dotm(n, wo) < 0.f) n = n * -1.f;
    const f3 li = square_to_hemisphere_cosine(xix, xiy);
    r.wi = to_world(frame_from_normal(n), li);
    r.pdf = li.z * kInvPi;
    r.f = mat.albedo * kInvPi;
    r.ok = true;
    return r;
}

// frag:800-837 (the 0-degree axisAngle rotation is the identity)
__device__ __forceinline__ Bxdf microfacet_bxdf(f3 n, const Material &mat, float xix, float xiy, f3 wo)
{
    Bxdf r;
    r.ok = false;
    r.f = mk(0.f, 0.f, 0.f);
    r.wi = mk(0.f, 0.f, 0.f);
    r.pdf = 0.f;
    if (dotm(n, wo) < 0.f) n = n * -1.f;
    // frag:581-593
    float ax = mat.roughness * mat.roughness, ay = ax;
    if (mat.anisotropy > 0.f) {
        const float aspect = ssqrt(1.f - mat.anisotropy * 0.9f);
        ax = sdiv(ax, aspect);
        ay *= aspect;
    }
    const Frame fr = frame_from_normal_fixed_tangent(n);
    const f3 lo = to_local(fr, wo);
    if (lo.z == 0.f) return r;
    const f3 m = tr_sample_normal(lo, xix, xiy, ax, ay);
    const f3 li = reflectm(-lo, m);
    if (li.z * lo.z <= 0.f) return r;
    r.ok = true;
    r.wi = to_world(fr, li);
    r.pdf = sdiv(tr_density(lo, m, ax, ay), 4.f * dotm(lo, m));
    // frag:686-702
    const float co = fabsf(lo.z), ci = fabsf(li.z);
    f3 hm = lo + li;
    if (ci <= 0.f || co <= 0.f) return r;
    if (hm.x == 0.f && hm.y == 0.f && hm.z == 0.f) return r;
    hm = snormalize(hm);
    const float p5 = pow5(1.f - fabsf(dotm(lo, hm)));
    const f3 F = mk(mat.albedo.x + (1.f - mat.albedo.x) * p5, mat.albedo.y + (1.f - mat.albedo.y) * p5,
                    mat.albedo.z + (1.f - mat.albedo.z) * p5);
    const float dg = tr_distribution(hm, ax, ay) * tr_masking(lo, li, ax, ay);
    const float ik = sdiv(1.f, 4.f * ci * co);
    r.f = mk(dg * F.x * ik, dg * F.y * ik, dg * F.z * ik);
    return r;
}

// frag:549-577
__device__ __forceinline__ bool refractm(f3 in, f3 normal, float eta, float &rel_eta, f3 &out)
{
    float ci = dotm(normal, in);
    if (ci < 0.f) {
        eta = sdiv(1.f, eta);
        ci = -ci;
        normal = -normal;
    }
    const float s2i = 1.f - ci * ci;
    const float ie = sdiv(1.f, eta);
    const float s2t = s2i * (ie * ie);
    if (s2t >= 1.f) return false;
    const float ct = ssqrt(1.f - s2t);
    const float k = ci * ie - ct;
    out = mk(-in.x * ie + k * normal.x, -in.y * ie + k * normal.y, -in.z * ie + k * normal.z);
    rel_eta = eta;
    return true;
}

// frag:723-798: rough dielectric with alpha = roughness (not squared), fixed
// R0 = 0.05, and a 3-way lobe choice that consumes one more random number.
__device__ __forceinline__ Bxdf dielectric_bxdf(f3 n, const Material &mat, float xix, float xiy, f3 wo, uint32_t &sx,
                                                uint32_t &sy)
{
    Bxdf r;
    r.ok = false;
    r.f = mk(0.f, 0.f, 0.f);
    r.wi = mk(0.f, 0.f, 0.f);
    r.pdf = 0.f;
    const float ax = mat.roughness, ay = mat.roughness;
    const Frame fr = frame_from_normal(n);
    const f3 lo = to_local(fr, wo);
    if (lo.z == 0.f) return r;
    const f3 m = tr_sample_normal(lo, xix, xiy, ax, ay);
    const float R = 0.05f + (1.f - 0.05f) * pow5(1.f - fabsf(dotm(lo, m)));
    const float T = 1.f - R;
    const float p_reflect = R;
    float p_transmit = T * mat.transmission;
    float p_diffuse = T * (1.f - mat.transmission);
    if (lo.z < 0.f) {
        p_transmit = T;
        p_diffuse = 0.f;
    }
    const float u = rng_next(sx, sy);
    if (u <= p_reflect) {
        const f3 li = reflectm(-lo, m);
        if (li.z * lo.z <= 0.f) return r;
        r.ok = true;
        r.wi = to_world(fr, li);
        r.pdf = sdiv(tr_density(lo, m, ax, ay), 4.f * dotm(lo, m)) * p_reflect;
        const float v = sdiv(tr_distribution(m, ax, ay) * tr_masking(lo, li, ax, ay) * R, 4.f * li.z * lo.z);
        r.f = mk(v, v, v);
        return r;
    }
    if (u <= p_reflect + p_transmit) {
        float rel_eta = 0.f;
        f3 li = mk(0.f, 0.f, 0.f);
        const bool tir = !refractm(lo, m, mat.ior, rel_eta, li);
        if (tir || lo.z * li.z > 0.f || li.z == 0.f) return r;
        r.ok = true;
        r.wi = to_world(fr, li);
        const float denom = dotm(li, m) + sdiv(dotm(lo, m), rel_eta);
        const float dm = sdiv(fabsf(dotm(li, m)), denom * denom);
        r.pdf = tr_density(lo, m, ax, ay) * dm * p_transmit;
        const float v = sdiv(tr_distribution(m, ax, ay) * tr_masking(lo, li, ax, ay) * T * dotm(li, m) * dotm(lo, m),
                             li.z * lo.z * denom * denom);
        r.f = mat.albedo * v;
        return r;
    }
    r = diffuse_bxdf(n, mat, xix, xiy, wo);
    r.pdf *= p_diffuse;
    return r;
}

// frag:841-858
__device__ __forceinline__ Bxdf sample_surface(const Material &mat, f3 n, float xix, float xiy, f3 wo, uint32_t &sx,
                                               uint32_t &sy)
{
    if (mat.metallic >= rng_next(sx, sy)) return microfacet_bxdf(n, mat, xix, xiy, wo);
    if (mat.roughness < 1.f) return dielectric_bxdf(n, mat, xix, xiy, wo, sx, sy);
    return diffuse_bxdf(n, mat, xix, xiy, wo);
}

__device__ __forceinline__ Material load_material(const SceneView &sc, uint32_t idx)
{
    Material m;
    if (idx < sc.n_materials) {
        const float4 a = sc.materials[2 * idx], b = sc.materials[2 * idx + 1];
        m.albedo = mk(a.x, a.y, a.z);
        m.roughness = a.w;
        m.metallic = b.x;
        m.ior = b.y;
        m.anisotropy = b.z;
        m.transmission = b.w;
    } else {
        // A face without usemtl carries index 0xFFFFFFFF (scene.h:133); an
        // out-of-range buffer-texture fetch reads zeros.
        m.albedo = mk(0.f, 0.f, 0.f);
        m.roughness = m.metallic = m.ior = m.anisotropy = m.transmission = 0.f;
    }
    return m;
}

// frag:128-147, :434-450 on one shading record: barycentric coordinates of the hit point, then the corner
// normals weighted by them, not renormalised.
template <bool EXACT_DIVISION = true>
__device__ __forceinline__ f3 interpolated_normal(float4 s0, float4 s1, float4 s2, float4 s3, float4 s4, f3 p)
{
    const f3 v0 = mk(s0.x, s0.y, s0.z), v1 = mk(s1.x, s1.y, s1.z), v2 = mk(s2.x, s2.y, s2.z);
    const f3 e0 = v1 - v0, e1 = v2 - v0, eP = p - v0;
    const float e00 = dotm(e0, e0), e01 = dotm(e0, e1), e11 = dotm(e1, e1);
    const float e0P = dotm(e0, eP), e1P = dotm(e1, eP);
    // (a correctly rounded division where the primary hits' normals are read back and compared with the reference's
    // bit for bit; the render loop takes the shading stage's own division)
    const float inv = EXACT_DIVISION ? 1.f / (e00 * e11 - e01 * e01) : sdiv(1.f, e00 * e11 - e01 * e01);
    const float bv = (e11 * e0P - e01 * e1P) * inv;
    const float bw = (e00 * e1P - e01 * e0P) * inv;
    const float bu = 1.f - bv - bw;
    const f3 n1 = mk(s0.w, s1.w, s2.w), n2 = mk(s3.x, s3.y, s3.z), n3 = mk(s4.x, s4.y, s4.z);
    return (n1 * bu + n2 * bv) + n3 * bw;  // frag:450
}

// One bounce for every surface hit of the hit list (frag:884-897), then wave64
// compaction of the surviving paths into the wave's own region of the next
// queue: one ballot + prefix rank per 64 hits, no atomics.  A path that dies
// here (pdf <= 0, a lobe that returned without a direction, zero throughput)
// simply is not written: its sample slot already holds zero.  final_bounce: the
// hits are those of the paths' last allowed segment; nothing survives, the BSDF
// is evaluated only because a non-finite value it produces is the path's sample.
// (A device function for the same reason as extend_body; BLOCK = threads per block of the launch it runs in.)
template <int BLOCK>
__device__ __forceinline__ void shade_body(const SceneView &sc, const Regions &rg, const PathQueue &in, const float4 *hit_list,
                                           const PathQueue &out, float4 *samples, const BatchParams *bp,
                                           uint32_t mode, WaveTurns turns, unsigned long long *unlit_skipped)
{
    // mode: kShadeFinal = the hits are those of the paths' last allowed segment (see above); kShadeBeforeLast = the
    // rays made here travel the last allowed segment, where only reaching the light can add anything (a surface
    // hit there is not shaded, an escape is black): in a scene with one light a ray that does not point at the
    // light's rectangle -- the light's distance beside it is NaN -- is therefore not written at all, unless its
    // path is poisoned (then whatever ends it writes its non-finite sample).  Such rays are counted, per region,
    // into *unlit_skipped: they are segments of their paths all the same.
    const bool final_bounce = mode == kShadeFinal;
    const bool drop_unlit = mode == kShadeBeforeLast && sc.n_lights == 1u;
    const uint32_t sample_tag = bp->sample_tag;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    uint32_t k = blockIdx.x * (BLOCK / 64) + wave_in_block();
    uint32_t nxt_r = turns.r;
    uint32_t nxt_packed = k < rg.n_regions ? load_count(rg.counts_in + nxt_r) : 0u;
    while (k < rg.n_regions) {
      const uint32_t r = nxt_r;
      // hits on purely diffuse surfaces from the region's first slot upwards (low half of the count),
      // all others from its last slot downwards (high half).  Full batches of 64 of one kind, then of
      // the other; what is left of both shares one batch when it fits.
      const uint32_t packed = nxt_packed;
      k += n_waves;
      if (k < rg.n_regions) {  // the next region's count is fetched while this one is shaded
          nxt_r = next_region(rg, nxt_r, turns.step);
          nxt_packed = load_count(rg.counts_in + nxt_r);
      }
      const uint32_t n0 = packed & 0xFFFFu, n1 = packed >> 16;
      const uint32_t full0 = n0 >> 6, full1 = n1 >> 6, rem0 = n0 & 63u, rem1 = n1 & 63u;
      const bool merged = rem0 + rem1 <= 64u;
      const uint32_t batches = full0 + full1 + (merged ? ((rem0 + rem1) != 0u ? 1u : 0u) : 2u);
      const uint32_t region_base = r * rg.size;
      const uint32_t last_slot = rg.size - 1u;
      uint32_t out_n = 0;  // wave-uniform: survivors written to this region so far
      uint32_t skipped_n = 0;  // wave-uniform: survivors not written because they cannot reach the light on their last segment
      for (uint32_t bt = 0; bt < batches; ++bt) {
        // which hit this lane takes: index from the front (take0) or from the back (take1) of the list
        bool take0 = false, take1 = false;
        uint32_t j = 0;
        if (bt < full0) {
            take0 = true;
            j = (bt << 6) + lane;
        } else if (bt < full0 + full1) {
            take1 = true;
            j = ((bt - full0) << 6) + lane;
        } else if (merged) {
            take0 = lane < rem0;
            take1 = !take0 && lane < rem0 + rem1;
            j = take0 ? (full0 << 6) + lane : (full1 << 6) + (lane - rem0);
        } else if (bt == full0 + full1) {
            take0 = lane < rem0;
            j = (full0 << 6) + lane;
        } else {
            take1 = lane < rem1;
            j = (full1 << 6) + lane;
        }
        bool alive = false;
        uint32_t flag = 0u;  // kPoisonBit for a survivor whose throughput is not finite in every channel
        float4 o0, o1, o2;
        if (take0 || take1) {
            const float4 rec = hit_list[region_base + (take1 ? last_slot - j : j)];
            const uint32_t i = region_base + __float_as_uint(rec.z);
            const float4 a = in.p0[i], b = in.p1[i], c = in.p2[i];
            const int code = __float_as_int(rec.y);
            const f3 thr = mk(c.x, c.y, c.z);
            {
                const f3 o = mk(a.x, a.y, a.z), d = mk(a.w, b.x, b.y);
                uint32_t sx = __float_as_uint(b.z), sy = __float_as_uint(b.w);
                // frag:434-453
                const float t = rec.x;
                const f3 p = mk(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
                const float4 *sp = sc.shade + 5 * static_cast<size_t>(code);
                const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
                const f3 nrm = interpolated_normal<false>(s0, s1, s2, s3, s4, p);

                const float xix = rng_next(sx, sy);
                const float xiy = rng_next(sx, sy);
                const Material mat = load_material(sc, __float_as_uint(s3.w));
                const Bxdf r = sample_surface(mat, nrm, xix, xiy, -d, sx, sy);
                bool dead = !r.ok;
                f3 att = thr;
                if (!dead) {
                    const float cs = fabsf(dotm(nrm, r.wi));
                    att = mk(att.x * (r.f.x * cs), att.y * (r.f.y * cs), att.z * (r.f.z * cs));
                    if (r.pdf <= 0.f) {
                        dead = true;  // frag:890-894: attenuation = vec3(0), whatever it was
                    } else {
                        const float ipdf = sdiv(1.f, r.pdf);
                        att = mk(att.x * ipdf, att.y * ipdf, att.z * ipdf);
                        // A path whose throughput is exactly zero can only ever add zero.
                        if (att.x == 0.f && att.y == 0.f && att.z == 0.f) dead = true;
                    }
                }
                if (!dead) {
                    const f3 no = mk(p.x + r.wi.x * 0.0001f, p.y + r.wi.y * 0.0001f, p.z + r.wi.z * 0.0001f);
                    // The reference never ends a path because its throughput went NaN or inf (frag:888-897): the
                    // path goes on, and its sample is attenuation * intensity (frag:901) -- non-finite in the
                    // affected channels wherever it ends, unless a later bounce zeroes the attenuation
                    // (pdf <= 0).  Such a path is traced on with its flag set beside the ray, so that the kernel
                    // that ends it writes that sample.  A ray that is itself non-finite (the NaN tangent frame of
                    // frag:171) turns everything after it NaN: its sample is written here.
                    const bool poisoned = !(isfinite(att.x) && isfinite(att.y) && isfinite(att.z));
                    const bool bad_ray = !(isfinite(no.x) && isfinite(no.y) && isfinite(no.z) && isfinite(r.wi.x) &&
                                           isfinite(r.wi.y) && isfinite(r.wi.z));
                    if (bad_ray) {
                        const float nan = __builtin_nanf("");
                        samples[__float_as_uint(c.w)] = make_float4(nan, nan, nan, __uint_as_float(sample_tag));
                    } else if (final_bounce) {
                        // the path's last allowed segment ended on this surface: intensity stays vec3(0)
                        if (poisoned) write_poisoned_sample(samples, make_float4(att.x, att.y, att.z, c.w), sample_tag);
                    } else {
                        alive = true;
                        o0 = make_float4(no.x, no.y, no.z, r.wi.x);
                        o1 = make_float4(r.wi.y, r.wi.z, __uint_as_float(sx), __uint_as_float(sy));
                        o2 = make_float4(att.x, att.y, att.z, c.w);
                        flag = poisoned ? kPoisonBit : 0u;
                    }
                }
            }
        }
        float tl = __uint_as_float(kNoLightBits);
        bool unlit = false;
        if (alive) {
            if (sc.n_lights == 1u) tl = light_distance(sc.lights[0], mk(o0.x, o0.y, o0.z), mk(o0.w, o1.x, o1.y));
            if (drop_unlit && !(fabsf(tl) >= 0.f) && flag == 0u) {
                alive = false;
                unlit = true;
            }
        }
        const unsigned long long live = __ballot(alive);
        if (drop_unlit) skipped_n += static_cast<uint32_t>(__popcll(__ballot(unlit)));
        if (alive) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
            const uint32_t dst = region_base + out_n + rank;
            out.p0[dst] = o0;
            out.p1[dst] = o1;
            out.p2[dst] = o2;
            out.lt[dst] = __uint_as_float(__float_as_uint(tl) | flag);
        }
        out_n += static_cast<uint32_t>(__popcll(live));
      }
      if (lane == 0) {
          rg.counts_out[r] = out_n;
          if (drop_unlit && skipped_n != 0u) atomicAdd(unlit_skipped, static_cast<unsigned long long>(skipped_n));
      }
    }
}

__global__ void __launch_bounds__(kBlock, PT_SHADE_WAVES) k_shade(SceneView sc, Regions rg, PathQueue in,
                                                   const float4 *__restrict__ hit_list, PathQueue out,
                                                   float4 *__restrict__ samples, const BatchParams *__restrict__ bp,
                                                   uint32_t mode, unsigned long long *__restrict__ unlit_skipped)
{
    const WaveTurns turns = wave_turns(rg, blockIdx.x * (kBlock / 64) + wave_in_block(), gridDim.x * (kBlock / 64));
    shade_body<kBlock>(sc, rg, in, hit_list, out, samples, bp, mode, turns, unlit_skipped);
}

// ------------------------------------------------- all bounces in one launch

// A batch from its primary rays through every bounce in ONE launch.  A region's data never leaves its wave: the
// traversal reads region r of the ray queue and writes region r of the hit list, the shading reads those and
// writes region r of the other queue, and a wave visits the same regions in both (turns k, k + n_waves, ... of
// the same grid).  So a wave carries its regions through all bounces with no synchronisation between waves;
// between two phases it only has to see its own writes (a workgroup-scope fence: its stores have reached the
// CU's cache, through which it also reads).  Against 2 * depth - 1 launches this removes the start-up and drain
// of kernels over thinning queues, which is what a small batch -- the reference's one sample per draw -- spends
// its time on.  Render loop only: no counters, no exact pass.
// Small blocks are compiled for five waves per SIMD (102 registers; the shading code alone takes 107 at four):
// when the whole scene is in LDS in small blocks, five of them per CU retire and are replaced one by one, and
// the kernel then beats the separate launches at every batch size (Box +4 %).  One block of 1024 threads per
// CU cannot overlap its waves' tails with anything (Veggie -8 % on large batches).
// The primary rays are region-local as well: slot i of the queue is sample i / (W * H) of the pixel that slot
// i mod (W * H) maps to, so the wave that will trace a region also makes its rays (what k_raygen does for the
// whole queue at once), and reads them back from the cache rather than from memory.
// The primary rays of the render loop, region by region: slot i of the batch is sample i / (W * H) of the pixel
// that slot i mod (W * H) maps to, so a region's rays are a function of its slot numbers, and the wave that will
// trace a region can also make its rays (and read them back from the cache rather than from memory).
//
// cull: a ray that meets neither the scene's bounds nor a light can only end as a miss, which adds nothing
// (frag:876: escape = black).  The host projects the scene's bounding box and the lights' rectangles onto the
// image and hands over the window of pixels outside of which every ray is of that kind (DeviceCamera::cull_*;
// more than half of all pixels with the reference's camera, which sees the box in the middle of a wide frame).
// Rays of pixels outside the window are counted and never made -- the 64 slots a wave handles together are an
// 8 x 8 tile of the image, so mostly a whole wave skips the work --, the others are packed to the front of the
// region (ballot + prefix rank, as everywhere).  Because a ray's position in the region then no longer says
// which pixel it belongs to, lane p2 (throughput 1, the path's sample slot iteration_in_batch * W*H + pixel) is
// stored for primary rays like for all others.  cull = 0 when the reference's non-finite samples are wanted one
// for one (a ray exactly parallel to a triangle's plane is a hit at t = inf there wherever it points).
// culled_out[r]: rays of region r not made.
template <int BLOCK>
__device__ __forceinline__ void raygen_regions(const SceneView &sc, const BatchParams *bp, uint32_t it_stride, uint32_t n_iterations,
                                               const PathQueue &out, const Regions &rg, uint32_t *culled_out, uint32_t cull,
                                               WaveTurns turns)
{
    const DeviceCamera cam = bp->cam;
    const uint32_t it_first = bp->it_first;
    const uint32_t n_pixels = cam.width * cam.height;
    const uint32_t total = n_pixels * n_iterations;
    const PrimaryFrame frame = primary_frame(cam);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    uint32_t r = turns.r;
    for (uint32_t k = blockIdx.x * (BLOCK / 64) + wave_in_block(); k < rg.n_regions; k += n_waves, r = next_region(rg, r, turns.step)) {
        const uint32_t base = r * rg.size, n = min(rg.size, total - base);
        uint32_t out_n = 0;  // wave-uniform: rays written to this region so far
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            // The 64 slots from base + i0 on are one 8 x 8 tile of the image when slots map to tiles (base and the
            // number of pixels are multiples of 64 then): whether the tile lies wholly outside the window is worked
            // out once for the wave, in scalar arithmetic, before any lane computes its pixel.
            bool whole_tile = false;  // (wave-uniform) the tile's corner and iteration are known: x0, y0, tile_it
            uint32_t x0 = 0, y0 = 0, tile_it = 0;
            if (cam.map_mode == 2u && cam.div_map_units.magic != 0xFFFFFFFFu && i0 + 64u <= n) {
                uint32_t j0, unit, tx;
                tile_it = div_by(base + i0, n_pixels, cam.div_pixels, j0);
                const uint32_t u = j0 >> 6;
                if ((j0 & 63u) == 0u && u < cam.map_units) {
                    (void)div_by(u * cam.map_stride, cam.map_units, cam.div_map_units, unit);
                    const uint32_t ty = div_by(unit, cam.tiles_x, cam.div_tiles_x, tx);
                    x0 = tx << 3, y0 = ty << 3;
                    whole_tile = true;
                    if (cull && (x0 > cam.cull_x1 || x0 + 7u < cam.cull_x0 || y0 > cam.cull_y1 || y0 + 7u < cam.cull_y0)) continue;
                }
            }
            bool keep = false;
            uint32_t x = 0, y = 0, it = 0, pixel = 0;
            if (whole_tile) {
                x = x0 + (lane & 7u);
                y = y0 + (lane >> 3);
                it = tile_it;
                pixel = y * cam.width + x;
                keep = !cull || (x >= cam.cull_x0 && x <= cam.cull_x1 && y >= cam.cull_y0 && y <= cam.cull_y1);
            } else if (i0 + lane < n) {
                uint32_t j;
                it = div_by(base + i0 + lane, n_pixels, cam.div_pixels, j);
                pixel = pixel_of_slot(cam, j);
                y = div_by(pixel, cam.width, cam.div_width, x);
                keep = !cull || (x >= cam.cull_x0 && x <= cam.cull_x1 && y >= cam.cull_y0 && y <= cam.cull_y1);
            }
            const unsigned long long kept = __ballot(keep);
            if (kept == 0ull) continue;  // (wave-uniform) nothing of this tile can meet anything
            float4 o0, o1, o2;
            float lt = __uint_as_float(kNoLightBits);
            if (keep) {
                const PrimaryRay pr = primary_ray(cam, frame, x, y, it_first + it * it_stride);
                if (sc.n_lights == 1u) lt = light_distance(sc.lights[0], frame.eye, pr.d);
                o0 = make_float4(frame.eye.x, frame.eye.y, frame.eye.z, pr.d.x);
                o1 = make_float4(pr.d.y, pr.d.z, __uint_as_float(pr.sx), __uint_as_float(pr.sy));
                o2 = make_float4(1.f, 1.f, 1.f, __uint_as_float(it * n_pixels + pixel));
            }
            if (keep) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(kept >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(kept), 0u));
                const uint32_t dst = base + out_n + rank;
                out.p0[dst] = o0;
                out.p1[dst] = o1;
                out.p2[dst] = o2;
                out.lt[dst] = lt;
            }
            out_n += static_cast<uint32_t>(__popcll(kept));
        }
        if (lane == 0) {
            rg.counts_out[r] = out_n;
            culled_out[r] = n - out_n;
        }
    }
}

// The same as a kernel of its own, for batches that run as one launch per kernel and bounce.
__global__ void __launch_bounds__(kBlock) k_raygen_regions(SceneView sc, const BatchParams *__restrict__ bp, uint32_t it_stride,
                                                            uint32_t n_iterations, PathQueue out, Regions rg, uint32_t *culled_out,
                                                            uint32_t cull)
{
    const WaveTurns turns = wave_turns(rg, blockIdx.x * (kBlock / 64) + wave_in_block(), gridDim.x * (kBlock / 64));
    raygen_regions<kBlock>(sc, bp, it_stride, n_iterations, out, rg, culled_out, cull, turns);
}

template <int BLOCK, int SCENE_LDS, bool USE_LT>
__global__ void __launch_bounds__(BLOCK, BLOCK == 256 ? 5 : 4) k_bounces(SceneView sc, BounceLoop L, uint32_t n_nodes, uint32_t refill_at,
                                                      uint32_t stack_entries, uint32_t n_lds_nodes, uint32_t lds_stride,
                                                      const BatchParams *__restrict__ bp, uint32_t lds_stack_entries,
                                                      int *stack_overflow, uint32_t it_stride, uint32_t n_iterations, uint32_t n_lds_tris)
{
    const TraversalCounters no_counters{nullptr, nullptr, nullptr};
    // the wave's regions are the same in every phase: where it starts and how it steps is worked out once
    WaveTurns turns;
    {
        const Regions rg{nullptr, L.counts, L.n_regions, L.stride, L.size};
        turns = wave_turns(rg, blockIdx.x * (BLOCK / 64) + wave_in_block(), gridDim.x * (BLOCK / 64));
        raygen_regions<BLOCK>(sc, bp, it_stride, n_iterations, L.q[0], rg, L.counts + static_cast<size_t>(2u * L.depth + 1u) * L.pitch, 1u, turns);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    for (uint32_t b = 0; b < L.depth; ++b) {
        const uint32_t last = b + 1u == L.depth ? 1u : 0u;
        uint32_t *rays = L.counts + static_cast<size_t>(b) * L.pitch, *hits = L.counts + static_cast<size_t>(L.depth + 1u + b) * L.pitch;
        const Regions re{rays, hits, L.n_regions, L.stride, L.size};
        extend_body<BLOCK, false, false, SCENE_LDS, true, USE_LT>(sc, re, L.q[b & 1u], L.hit_list, L.samples, last, no_counters, n_nodes,
                                                                 refill_at, stack_entries, n_lds_nodes, lds_stride, bp,
                                                                 lds_stack_entries, stack_overflow, b == 0u, turns, n_lds_tris);
        if (last) break;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const Regions rs{hits, L.counts + static_cast<size_t>(b + 1u) * L.pitch, L.n_regions, L.stride, L.size};
        shade_body<BLOCK>(sc, rs, L.q[b & 1u], L.hit_list, L.q[(b + 1u) & 1u], L.samples, bp,
                          b + 2u == L.depth ? kShadeBeforeLast : kShadeBounce, turns, L.unlit_skipped);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// ------------------------------------------------------------ accumulate

// GL's float -> UNORM8 store: clamp to [0,1], round(x * 255); NaN -> 0.
__device__ __forceinline__ unsigned char unorm8(float x)
{
    if (!(x > 0.f)) return 0;
    if (x >= 1.f) return 255;
    return static_cast<unsigned char>(static_cast<int>(x * 255.f + 0.5f));
}

// sum[pixel] += each of the batch's samples of that pixel, in iteration order
// (so the image does not depend on how many iterations share a batch).
// RGBA8: also keep the reference's 8-bit running average (a variant of its own, so that the usual one stays within
// 32 registers: that is what five resident waves of the kernel that runs all bounces of a batch, 96 registers each,
// leave free on a SIMD -- this kernel's blocks then start beside them instead of waiting for one of them to retire,
// and the other batch's next launch waits for this kernel).
template <bool RGBA8>
__global__ void __launch_bounds__(kBlock) k_accumulate(const float4 *__restrict__ samples, float4 *__restrict__ sum,
                                                        uint32_t n_pixels, uint32_t n_iterations,
                                                        const uint32_t *__restrict__ counts, uint32_t n_rows,
                                                        uint32_t n_regions, uint32_t pitch,
                                                        unsigned long long *__restrict__ segments_total,
                                                        unsigned long long *__restrict__ shade_totals,
                                                        unsigned long long *__restrict__ nan_counter,
                                                        unsigned long long *__restrict__ culled_total,
                                                        uint8_t *__restrict__ rgba8, const BatchParams *__restrict__ bp,
                                                        uint32_t it_stride, uint32_t culled)
{
    const uint32_t it_first = bp->it_first, sample_tag = bp->sample_tag;
    {
        // surface hits shaded in this batch and the 64-lane batches the shade kernel formed of them
        // (rows n_rows + 1 ... 2 * n_rows - 1 hold the packed hit counts of the bounces that are shaded)
        unsigned long long hits = 0, batches = 0;
        for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < (n_rows - 1u) * n_regions; k += gridDim.x * kBlock) {
            const uint32_t packed = counts[(n_rows + 1u + k / n_regions) * pitch + (k % n_regions)];
            const uint32_t n0 = packed & 0xFFFFu, n1 = packed >> 16, rem0 = n0 & 63u, rem1 = n1 & 63u;
            hits += n0 + n1;
            batches += (n0 >> 6) + (n1 >> 6) + (rem0 + rem1 <= 64u ? (rem0 + rem1 != 0u ? 1u : 0u) : 2u);
        }
        for (int off = 32; off > 0; off >>= 1) {
            hits += __shfl_down(hits, off);
            batches += __shfl_down(batches, off);
        }
        if ((threadIdx.x & 63u) == 0 && hits != 0) {
            atomicAdd(shade_totals, hits);
            atomicAdd(shade_totals + 1, batches);
        }
    }
    {
        // segments of this batch = sum of all live counts of all bounces, plus the primary rays that ray
        // generation found to miss the scene and did not enqueue (row 2 * n_rows + 1): a path's first segment
        // all the same
        unsigned long long s = 0, culled = 0;
        for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_rows * n_regions; k += gridDim.x * kBlock)
            s += counts[(k / n_regions) * pitch + (k % n_regions)];
        for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_regions; k += gridDim.x * kBlock)
            culled += counts[static_cast<size_t>(2u * n_rows + 1u) * pitch + k];
        s += culled;
        for (int off = 32; off > 0; off >>= 1) {
            s += __shfl_down(s, off);
            culled += __shfl_down(culled, off);
        }
        if ((threadIdx.x & 63u) == 0 && s != 0) atomicAdd(segments_total, s);
        if ((threadIdx.x & 63u) == 0 && culled != 0) atomicAdd(culled_total, culled);
    }
    unsigned long long dropped = 0;
    // culled: ray generation made no ray for the pixels outside the camera's window (raygen_regions), so none of
    // their sample slots was written in this batch: they get their count and nothing is read for them
    const uint32_t width = bp->cam.width, cx0 = bp->cam.cull_x0, cx1 = bp->cam.cull_x1, cy0 = bp->cam.cull_y0, cy1 = bp->cam.cull_y1;
    const FastDiv div_width = bp->cam.div_width;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += gridDim.x * kBlock) {
        float4 acc = sum[p];
        if (culled) {
            uint32_t x;
            const uint32_t y = div_by(p, width, div_width, x);
            if (x < cx0 || x > cx1 || y < cy0 || y > cy1) {
                for (uint32_t k = 0; k < n_iterations; ++k) acc.w += 1.f;  // (one by one, like the samples that are read)
                sum[p] = acc;
                continue;
            }
        }
        uchar4 tex = RGBA8 ? reinterpret_cast<uchar4 *>(rgba8)[p] : make_uchar4(0, 0, 0, 0);
        for (uint32_t k = 0; k < n_iterations; ++k) {
            float4 s = samples[static_cast<size_t>(k) * n_pixels + p];
            if (__float_as_uint(s.w) != sample_tag) s = make_float4(0.f, 0.f, 0.f, 0.f);  // not written in this batch: the path escaped
            if (isfinite(s.x) && isfinite(s.y) && isfinite(s.z)) {
                acc.x += s.x;
                acc.y += s.y;
                acc.z += s.z;
            } else {
                ++dropped;  // the float image counts a non-finite sample as zero
            }
            acc.w += 1.f;
            if (RGBA8) {
                // frag:900-904 on an 8-bit UNORM texel: col = (accumCol * it + passCol) / (it + 1), then the
                // store clamps to [0,1] and rounds to 8 bits (NaN becomes 0).
                const float fi = static_cast<float>(it_first + k * it_stride), fi1 = static_cast<float>(it_first + k * it_stride + 1u);
                const float cx = ((static_cast<float>(tex.x) / 255.f) * fi + s.x) / fi1;
                const float cy = ((static_cast<float>(tex.y) / 255.f) * fi + s.y) / fi1;
                const float cz = ((static_cast<float>(tex.z) / 255.f) * fi + s.z) / fi1;
                tex.x = unorm8(cx);
                tex.y = unorm8(cy);
                tex.z = unorm8(cz);
                tex.w = 255;
            }
        }
        sum[p] = acc;
        if (RGBA8) reinterpret_cast<uchar4 *>(rgba8)[p] = tex;
    }
    for (int off = 32; off > 0; off >>= 1) dropped += __shfl_down(dropped, off);
    if ((threadIdx.x & 63u) == 0 && dropped != 0) atomicAdd(nan_counter, dropped);
}

// --------------------------------------------------------------- helpers

// slot_is_pixel == 0: slot i holds the ray of pixel pixel_of_slot(i) (rays made by k_raygen).
__global__ void __launch_bounds__(kBlock) k_unpack_primary(uint32_t n, PathQueue q, const float2 *__restrict__ hits,
                                                            int32_t *ids, float *t, float *dirs, DeviceCamera cam,
                                                            uint32_t slot_is_pixel)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float2 h = hits[i];
        const size_t at = slot_is_pixel ? i : pixel_of_slot(cam, i);
        if (ids) ids[at] = __float_as_int(h.y);
        if (t) t[at] = h.x;
        if (dirs) {
            const float4 a = q.p0[i], b = q.p1[i];
            dirs[3 * at + 0] = a.w;
            dirs[3 * at + 1] = b.x;
            dirs[3 * at + 2] = b.y;
        }
    }
}

// Intersection::normal of the primary hits (frag:434-460): the interpolated normal on a surface, the
// light's transform * (0,0,1,0) on a light, zero on a miss.  Slot i holds the ray of pixel pixel_of_slot(i).
__global__ void __launch_bounds__(kBlock) k_primary_normals(uint32_t n, SceneView sc, PathQueue q,
                                                             const float2 *__restrict__ hits, float *__restrict__ normals,
                                                             DeviceCamera cam)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float2 h = hits[i];
        const int code = __float_as_int(h.y);
        f3 nrm = mk(0.f, 0.f, 0.f);
        if (code >= 0) {
            const float4 a = q.p0[i], b = q.p1[i];
            const float t = h.x;
            const f3 p = mk(a.x + a.w * t, a.y + b.x * t, a.z + b.y * t);
            const float4 *sp = sc.shade + 5 * static_cast<size_t>(code);
            nrm = interpolated_normal(sp[0], sp[1], sp[2], sp[3], sp[4], p);
        } else if (code <= -2) {
            const float *ln = sc.lights[-(code + 2)].normal;
            nrm = mk(ln[0], ln[1], ln[2]);
        }
        const size_t at = pixel_of_slot(cam, i);
        normals[3 * at + 0] = nrm.x;
        normals[3 * at + 1] = nrm.y;
        normals[3 * at + 2] = nrm.z;
    }
}

__global__ void __launch_bounds__(kBlock) k_pack_rays(uint32_t n, const float *__restrict__ origins,
                                                       const float *__restrict__ dirs, PathQueue q, uint32_t *counts_out,
                                                       uint32_t region_slots)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (i % region_slots == 0u) counts_out[i / region_slots] = min(region_slots, n - i);
        const size_t k = 3 * static_cast<size_t>(i);
        q.p0[i] = make_float4(origins[k], origins[k + 1], origins[k + 2], dirs[k]);
        q.p1[i] = make_float4(dirs[k + 1], dirs[k + 2], 0.f, 0.f);
        q.p2[i] = make_float4(1.f, 1.f, 1.f, __uint_as_float(i));
    }
}

// post.frag:11-15 on the mean; output rows top-down.
__global__ void __launch_bounds__(kBlock) k_tonemap(const float4 *__restrict__ sum, uint8_t *__restrict__ rgb,
                                                     uint32_t width, uint32_t height)
{
    const uint32_t n = width * height;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const float4 s = sum[p];
        const float inv = s.w > 0.f ? 1.f / s.w : 0.f;
        float c[3] = {s.x * inv, s.y * inv, s.z * inv};
        const uint32_t y = p / width, x = p - y * width;
        uint8_t *dst = rgb + 3 * (static_cast<size_t>(height - 1u - y) * width + x);
        for (int k = 0; k < 3; ++k) {
            float v = c[k] / (1.f + c[k]);
            v = powf(v, 1.f / 2.2f);
            v = fminf(fmaxf(v, 0.f), 1.f);  // NaN -> 0
            dst[k] = static_cast<uint8_t>(v * 255.f + 0.5f);
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_sample_surface(uint32_t n, const float *__restrict__ materials,
                                                            const float *__restrict__ normals,
                                                            const float *__restrict__ wo, const float *__restrict__ xi,
                                                            const uint32_t *__restrict__ seeds, float *wi_out,
                                                            float *pdf_out, float *f_out, uint32_t *flags_out,
                                                            uint32_t *seeds_out)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float *mp = materials + 8 * static_cast<size_t>(i);
        Material mat;
        mat.albedo = mk(mp[0], mp[1], mp[2]);
        mat.roughness = mp[3];
        mat.metallic = mp[4];
        mat.ior = mp[5];
        mat.anisotropy = mp[6];
        mat.transmission = mp[7];
        const size_t k = 3 * static_cast<size_t>(i);
        uint32_t sx = seeds[2 * static_cast<size_t>(i)], sy = seeds[2 * static_cast<size_t>(i) + 1];
        const Bxdf r = sample_surface(mat, mk(normals[k], normals[k + 1], normals[k + 2]), xi[2 * static_cast<size_t>(i)],
                                      xi[2 * static_cast<size_t>(i) + 1], mk(wo[k], wo[k + 1], wo[k + 2]), sx, sy);
        wi_out[k] = r.wi.x;
        wi_out[k + 1] = r.wi.y;
        wi_out[k + 2] = r.wi.z;
        f_out[k] = r.f.x;
        f_out[k + 1] = r.f.y;
        f_out[k + 2] = r.f.z;
        pdf_out[i] = r.pdf;
        flags_out[i] = r.ok ? 0u : 1u;
        seeds_out[2 * static_cast<size_t>(i)] = sx;
        seeds_out[2 * static_cast<size_t>(i) + 1] = sy;
    }
}

}  // namespace

// ------------------------------------------------------------ launchers

namespace {
// The region kernels are persistent: the grid is exactly what the device keeps
// resident (blocks per CU from the occupancy query for this kernel and LDS
// size), so no block waits for another to retire, and never more waves than
// there are regions to hand out.
// Blocks of `kernel` (block size `block`, `lds_bytes` of dynamic LDS) a CU keeps resident.  The runtime's
// occupancy query is asked once per (device, kernel, LDS size) and process: it costs more than a kernel launch,
// and the reference's usage pattern is one sample per draw (renderer.cpp:124-150), i.e. many small launches.
// The device is part of the key because the dynamic-LDS limit raised below is an attribute of the kernel ON
// the current device: a process that drives several GPUs has to raise it on each of them.
int blocks_per_cu(const void *kernel, int block, size_t lds_bytes)
{
    struct Key {
        int device;
        const void *kernel;
        size_t lds;
        int block;
        bool operator<(const Key &o) const
        {
            if (device != o.device) return device < o.device;
            return kernel != o.kernel ? kernel < o.kernel : (lds != o.lds ? lds < o.lds : block < o.block);
        }
    };
    static std::mutex mutex;
    static std::map<Key, int> cache;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    std::lock_guard<std::mutex> lock(mutex);
    const Key key{device, kernel, lds_bytes, block};
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)) != hipSuccess) {
        // the launch that follows would fail with "invalid value"; leave the error for the caller's hipGetLastError
        return 1;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    cache[key] = per_cu;
    return per_cu;
}

template <typename K>
uint32_t resident_grid(K kernel, size_t lds_bytes, LaunchShape g, int block = kBlock, const char *cap_env = nullptr)
{
    int per_cu = blocks_per_cu(reinterpret_cast<const void *>(kernel), block, lds_bytes);
    if (cap_env) {  // tuning experiments only: fewer blocks per CU than fit
        const char *e = pt::tuning_env(cap_env);
        const int cap = e ? std::atoi(e) : 0;
        if (cap >= 1 && cap < per_cu) per_cu = cap;
    }
    const uint32_t resident = g.n_cus * static_cast<uint32_t>(per_cu);
    const uint32_t needed = (g.wave_items + (block / 64) - 1) / (block / 64);
    if (pt::tuning_env("PT_EVEN_GRID")) {  // tuning experiments only
        const uint32_t wpb = block / 64, rw = resident * wpb, items = g.wave_items < 1 ? 1 : g.wave_items;
        const uint32_t turns = (items + rw - 1) / rw, waves = (items + turns - 1) / turns;
        return (waves + wpb - 1) / wpb;
    }
    return needed < resident ? (needed < 1 ? 1 : needed) : resident;
}
}  // namespace

void launch_set_batch(hipStream_t s, BatchParams *params, const DeviceCamera &cam, uint32_t it_first, uint32_t sample_tag)
{
    hipLaunchKernelGGL(k_set_batch, dim3(1), dim3(64), 0, s, params, cam, it_first, sample_tag);
}

void launch_raygen(hipStream_t s, LaunchShape g, const SceneView &scene, const BatchParams *params, uint32_t it_stride,
                   uint32_t n_iterations, PathQueue out, uint32_t *counts_out, uint32_t region_size)
{
    hipLaunchKernelGGL(k_raygen, dim3(g.blocks), dim3(kBlock), 0, s, scene, params, it_stride, n_iterations, out,
                       counts_out, region_size);
}

void launch_raygen_regions(hipStream_t s, LaunchShape g, const SceneView &scene, const BatchParams *params, uint32_t it_stride,
                           uint32_t n_iterations, PathQueue out, Regions regions, uint32_t *culled_out, bool cull)
{
    hipLaunchKernelGGL(k_raygen_regions, dim3(resident_grid(k_raygen_regions, 0, g)), dim3(kBlock), 0, s, scene, params, it_stride,
                       n_iterations, out, regions, culled_out, cull ? 1u : 0u);
}

namespace {
// Dynamic LDS of the traversal kernel: as much of the scene as the block may hold, then the per-lane stacks.
// Four blocks of 256 threads per CU with 40 KB each, two blocks of 512 with 80 KB each, or one block of 1024
// threads with 160 KB (16 waves per CU in each case; measured: 3 waves per SIMD is 27 % slower).  In order of
// preference: all nodes and triangles (mode 2) in the smallest blocks that hold them, all nodes (mode 3) in small
// blocks, in a large block, else small blocks with the top of the tree (1) or nothing (0).
constexpr size_t kLdsPerCu = 160 * 1024;
struct LdsPlan {
    int block;  // threads per block: 256, 512 or 1024
    int mode;
    uint32_t lds_stack_entries;  // entries of every lane's stack kept in LDS (the rest: the launch's overflow buffer)
    uint32_t n_lds_nodes;
    uint32_t stride;  // 16-byte lanes per triangle record in LDS: 5 (spread over the banks) when everything wanted still fits, else 4
    size_t bytes;
    uint32_t n_lds_tris = 0;  // leading records of the leaf-ordered triangle array kept in LDS (all of them in mode 2)
};
LdsPlan plan_lds(const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds, bool overflow_buffer)
{
    static const size_t budget_small = [] {
        const char *e = pt::tuning_env("PT_LDS_BUDGET_KB");  // tuning experiments only
        return static_cast<size_t>(e ? std::atoi(e) : 40) * 1024;
    }();
    static const int only_block = [] {
        const char *e = pt::tuning_env("PT_EXT_BLOCK");  // tuning experiments only: 256, 512 or 1024
        return e ? std::atoi(e) : 0;
    }();
    constexpr size_t budget_large = kLdsPerCu;
    static const bool no_hot_tris = pt::tuning_env("PT_NO_HOT_TRIS") != nullptr;  // tuning experiments only
    auto stack_bytes = [&](int block) { return static_cast<size_t>(stack_entries) * block * sizeof(StackT); };
    // mode 2 then mode 3; within a mode small blocks before the large one, triangle records spread over the
    // banks (80 bytes) before packed ones (64).  With all nodes in LDS a node takes 112 bytes (the kernel's own
    // layout of them); inner-child codes are LDS addresses there, which keeps the nodes below 2^31 by far.
    for (int mode : {2, 3})
        for (int block : {256, 512, 1024}) {
            if (only_block && block != only_block) continue;
            // (two blocks of 512 threads per CU, for scenes that fit twice with all their triangles: the same 16 waves
            // per CU as one block of 1024, but blocks retire and are replaced at half the granularity, so the tails of
            // one batch's launch and the head of the next overlap better -- Box_Transmission +3.8 %)
            if (block == 512 && mode != 2) continue;
            const size_t budget = block == 256 ? budget_small : (block == 512 ? budget_large / 2 : budget_large);
            for (uint32_t stride : {5u, 4u}) {
                const size_t bytes = stack_bytes(block) + static_cast<size_t>(n_nodes) * 112 +
                                     (mode == 2 ? static_cast<size_t>(scene.n_triangles) * 16 * stride : 0);
                if (scene_in_lds && bytes <= budget) {
                    LdsPlan p{block, mode, stack_entries, n_nodes, stride, bytes};
                    p.n_lds_tris = mode == 2 ? scene.n_triangles : 0u;
                    if (mode == 3 && !no_hot_tris) {
                        // What the nodes and the stacks leave free holds the head of the triangle array (80-byte
                        // records), which the host orders by leaf area.  Veggie.obj: 71 of 1450 triangles in the
                        // 5.7 KB left, traversal -2 %.  (Making more room by keeping only the first 8 entries of every
                        // stack in LDS, the rest in the overflow buffer, was measured: the test on every push and pop
                        // costs 9 % of the traversal, the 400 more triangles return 2.5 %.)
                        p.n_lds_tris = std::min<uint32_t>(scene.n_triangles, static_cast<uint32_t>((budget - bytes) / (16 * stride)));
                        p.bytes += static_cast<size_t>(p.n_lds_tris) * 16 * stride;
                    }
                    return p;
                }
                if (mode == 3) break;  // the stride of the few records that may fit changes little
            }
        }
    // The tree does not fit: it is read from memory, and the kernel waits for it.  Measured on 10 M triangles
    // with 1 to 4 blocks of 256 threads per CU: time per launch = 1.56 ms + 8.52 ms / blocks -- waves in flight
    // are what counts, and LDS is what limits them.  With an overflow buffer for the deep ends of the stacks
    // (kDeepStack in the kernel) a block keeps kDeepLdsStack entries per lane in LDS and takes a sixth of the
    // CU's LDS (24 waves per CU; the kernel's registers allow no more), the rest of which caches the top of
    // the tree.  Without the buffer: whole stacks in LDS, four blocks per CU.
    static const uint32_t deep_entries = [] {
        const char *e = pt::tuning_env("PT_DEEP_STACK");  // tuning experiments only
        return static_cast<uint32_t>(e ? std::max(4, std::atoi(e)) : 16);
    }();
    static const size_t deep_budget = [] {
        const char *e = pt::tuning_env("PT_DEEP_BUDGET_KB");  // tuning experiments only
        return static_cast<size_t>(e ? std::atoi(e) : 26) * 1024;
    }();
    const uint32_t lds_entries = scene.deep_stack_lds_entries ? scene.deep_stack_lds_entries : deep_entries;
    // Four-wide nodes (mode 4), when the scene has them: half the dependent node fetches per ray.  Nothing of the
    // tree is kept in LDS; a ray's stack can hold three entries per level.
    const bool wide = scene.wide_nodes != nullptr;
    if (wide) stack_entries = std::max(stack_entries, scene.wide_stack_entries);
    const bool deep = overflow_buffer && stack_entries > lds_entries && only_block != 1024;
    if (wide) {
        const uint32_t in_lds = deep ? lds_entries : stack_entries;
        return LdsPlan{256, 4, in_lds, 0, 4, static_cast<size_t>(in_lds) * 256 * sizeof(StackT)};
    }
    if (deep) {
        LdsPlan p{256, 0, lds_entries, 0, 4, 0};
        const size_t stacks = static_cast<size_t>(lds_entries) * 256 * sizeof(StackT);
        if (scene_in_lds && stacks + 16 * 64 <= deep_budget) {
            p.mode = 1;
            p.n_lds_nodes = static_cast<uint32_t>(std::min<size_t>((deep_budget - stacks) / 64, n_nodes));
        }
        p.bytes = stacks + static_cast<size_t>(p.n_lds_nodes) * 64;
        return p;
    }
    // Whole stacks in LDS, the top of the tree beside them (packed records: here capacity counts for more than
    // conflicts).  A large block caches four times as many nodes as a small one; it is chosen when that is
    // a good part of the tree (at least a quarter) -- for a tree of millions of nodes a longer prefix changes nothing.
    static const int top_share = [] {
        const char *e = pt::tuning_env("PT_TOP_SHARE");  // tuning experiments only
        return e ? std::atoi(e) : 4;
    }();
    const size_t cap_large = stack_bytes(1024) + 64 * 64 <= budget_large ? (budget_large - stack_bytes(1024)) / 64 : 0;
    const int block = (scene_in_lds && only_block != 256 && cap_large * top_share >= n_nodes) || only_block == 1024 ? 1024 : 256;
    const size_t budget = block == 256 ? budget_small : budget_large;
    LdsPlan p{block, 0, stack_entries, 0, 4, 0};
    if (scene_in_lds && stack_bytes(block) + 64 * 64 <= budget) {
        p.mode = 1;
        p.n_lds_nodes = static_cast<uint32_t>(std::min<size_t>((budget - stack_bytes(block)) / 64, n_nodes));
    } else {
        p.block = 256;
    }
    p.bytes = stack_bytes(p.block) + static_cast<size_t>(p.n_lds_nodes) * 64;
    return p;
}

// The instantiation of the traversal kernel for a plan and a set of switches, as a generic function pointer.
using ExtendKernel = void (*)(SceneView, Regions, PathQueue, float4 *, float4 *, uint32_t, TraversalCounters, uint32_t, uint32_t,
                              uint32_t, uint32_t, uint32_t, const BatchParams *, uint32_t, int *, uint32_t);
template <int BLOCK, int L>
ExtendKernel extend_kernel_for(bool counting, bool exact, bool render, bool use_lt)
{
    if (exact && render && counting && use_lt) return k_extend_bvh<BLOCK, true, true, L, true, true>;
    if (exact && render && counting) return k_extend_bvh<BLOCK, true, true, L, true, false>;
    if (exact && render && use_lt) return k_extend_bvh<BLOCK, false, true, L, true, true>;
    if (exact && render) return k_extend_bvh<BLOCK, false, true, L, true, false>;
    if (exact) return k_extend_bvh<BLOCK, false, true, L, false, false>;
    if (!render) return k_extend_bvh<BLOCK, false, false, L, false, false>;
    if (counting && use_lt) return k_extend_bvh<BLOCK, true, false, L, true, true>;
    if (counting) return k_extend_bvh<BLOCK, true, false, L, true, false>;
    if (use_lt) return k_extend_bvh<BLOCK, false, false, L, true, true>;
    return k_extend_bvh<BLOCK, false, false, L, true, false>;
}
ExtendKernel extend_kernel(const LdsPlan &plan, bool counting, bool exact, bool render, bool use_lt)
{
    if (plan.block == 512) return extend_kernel_for<512, 2>(counting, exact, render, use_lt);
    if (plan.block == 1024) {  // large blocks exist for the modes that keep something in LDS
        if (plan.mode == 2) return extend_kernel_for<1024, 2>(counting, exact, render, use_lt);
        if (plan.mode == 3) return extend_kernel_for<1024, 3>(counting, exact, render, use_lt);
        return extend_kernel_for<1024, 1>(counting, exact, render, use_lt);
    }
    if (plan.mode == 4) return extend_kernel_for<256, 4>(counting, exact, render, use_lt);
    if (plan.mode == 2) return extend_kernel_for<256, 2>(counting, exact, render, use_lt);
    if (plan.mode == 3) return extend_kernel_for<256, 3>(counting, exact, render, use_lt);
    if (plan.mode == 1) return extend_kernel_for<256, 1>(counting, exact, render, use_lt);
    return extend_kernel_for<256, 0>(counting, exact, render, use_lt);
}
}  // namespace

uint32_t extend_resident_waves(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes,
                               bool scene_in_lds, bool overflow_buffer)
{
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, overflow_buffer);
    int per_cu = blocks_per_cu(reinterpret_cast<const void *>(extend_kernel(plan, false, false, true, scene.n_lights == 1u)),
                               plan.block, plan.bytes);
    if (per_cu < 1) per_cu = 1;
    return n_cus * static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(plan.block / 64);
}

bool extend_reads_tree_from_memory(const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds)
{
    SceneView binary = scene;
    binary.wide_nodes = nullptr;
    return plan_lds(binary, stack_entries, n_nodes, scene_in_lds, true).mode <= 1;
}

size_t extend_stack_overflow_bytes(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes,
                                   bool scene_in_lds)
{
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, true);
    if (plan.mode == 4) stack_entries = std::max(stack_entries, scene.wide_stack_entries);
    if (plan.lds_stack_entries >= stack_entries) return 0;
    // the largest grid any instantiation for this plan is launched with: what a CU can hold of such blocks
    const size_t lanes = static_cast<size_t>(n_cus) * (kLdsPerCu / std::max<size_t>(plan.bytes, 1024)) * plan.block;
    return static_cast<size_t>(stack_entries - plan.lds_stack_entries) * std::max<size_t>(lanes, 1) * sizeof(int);
}

void launch_extend(hipStream_t s, LaunchShape g, const SceneView &scene, uint32_t stack_entries, bool use_bvh,
                   Regions regions, PathQueue q, const ExtendOutput &out, TraversalCounters tc, uint32_t n_nodes,
                   bool scene_in_lds)
{
    // Idle lanes that trigger a refill from the wave's region: 16 when the tree is read from LDS, 32 when
    // nearly all of it comes from L2 / HBM (a refill stalls the wave on the new rays' loads, and a wave that
    // waits for node fetches anyway loses less by running thinner for a while: +7 % on 10 M triangles).
    // PT_REFILL_AT: tuning experiments only.
    static const int refill_env = [] {
        const char *e = pt::tuning_env("PT_REFILL_AT");
        const int v = e ? std::atoi(e) : 0;
        return v < 0 ? 0 : (v > 64 ? 64 : v);
    }();
    const bool render = out.render;
    const bool exact_parallel = out.exact_parallel;
    const uint32_t last = out.last_segment ? 1u : 0u;
    if (!use_bvh) {
        if (render)
            hipLaunchKernelGGL(k_extend_scan<true>, dim3(resident_grid(k_extend_scan<true>, 0, g)), dim3(kBlock), 0, s, scene,
                               regions, q, out.hit_list, out.samples, last, out.params);
        else
            hipLaunchKernelGGL(k_extend_scan<false>, dim3(resident_grid(k_extend_scan<false>, 0, g)), dim3(kBlock), 0, s, scene,
                               regions, q, out.hit_list, out.samples, last, out.params);
        return;
    }
    const bool counting = tc.nodes_visited != nullptr;
    static const bool no_lt = pt::tuning_env("PT_NO_LT") != nullptr;  // tuning experiments only
    const bool use_lt = out.light_distance_in_queue && scene.n_lights == 1u && !no_lt && render;
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, out.stack_overflow != nullptr);
    const uint32_t refill_at = refill_env ? static_cast<uint32_t>(refill_env)
                                          : ((plan.mode == 0 || plan.mode == 4 || (plan.mode == 1 && plan.n_lds_nodes < n_nodes / 8u)) ? 32u : 16u);
    const ExtendKernel kernel = extend_kernel(plan, counting, exact_parallel, render, use_lt);
    hipLaunchKernelGGL(kernel, dim3(resident_grid(kernel, plan.bytes, g, plan.block, "PT_EXT_PER_CU")), dim3(plan.block), plan.bytes, s,
                       scene, regions, q, out.hit_list, out.samples, last, tc, n_nodes, refill_at, stack_entries, plan.n_lds_nodes,
                       plan.stride, out.params, plan.lds_stack_entries, out.stack_overflow, plan.n_lds_tris);
}

namespace {
using BouncesKernel = void (*)(SceneView, BounceLoop, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const BatchParams *, uint32_t, int *,
                               uint32_t, uint32_t, uint32_t);
template <int BLOCK, int L>
BouncesKernel bounces_kernel_for(bool use_lt)
{
    return use_lt ? k_bounces<BLOCK, L, true> : k_bounces<BLOCK, L, false>;
}
BouncesKernel bounces_kernel(const LdsPlan &plan, bool use_lt)
{
    if (plan.block == 512) return bounces_kernel_for<512, 2>(use_lt);
    if (plan.block == 1024) {
        if (plan.mode == 2) return bounces_kernel_for<1024, 2>(use_lt);
        if (plan.mode == 3) return bounces_kernel_for<1024, 3>(use_lt);
        return bounces_kernel_for<1024, 1>(use_lt);
    }
    if (plan.mode == 4) return bounces_kernel_for<256, 4>(use_lt);
    if (plan.mode == 2) return bounces_kernel_for<256, 2>(use_lt);
    if (plan.mode == 3) return bounces_kernel_for<256, 3>(use_lt);
    if (plan.mode == 1) return bounces_kernel_for<256, 1>(use_lt);
    return bounces_kernel_for<256, 0>(use_lt);
}
}  // namespace

uint32_t bounces_resident_waves(uint32_t n_cus, const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds,
                                bool overflow_buffer)
{
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, overflow_buffer);
    int per_cu = blocks_per_cu(reinterpret_cast<const void *>(bounces_kernel(plan, scene.n_lights == 1u)), plan.block, plan.bytes);
    if (per_cu < 1) per_cu = 1;
    return n_cus * static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(plan.block / 64);
}

bool bounces_suit_any_batch(const SceneView &scene, uint32_t stack_entries, uint32_t n_nodes, bool scene_in_lds, bool overflow_buffer)
{
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, overflow_buffer);
    // (one large block per CU: only with every triangle in LDS -- a block whose leaf tests go to memory loses more
    // without a second batch's kernels beside it than it gains from having no launch tails)
    return (plan.block == 256 && (plan.mode == 2 || plan.mode == 3)) || (plan.block >= 512 && plan.mode == 2);
}

void launch_bounces(hipStream_t s, LaunchShape g, const SceneView &scene, uint32_t stack_entries, const BounceLoop &loop,
                    const BatchParams *params, uint32_t n_nodes, bool scene_in_lds, int *stack_overflow, uint32_t it_stride,
                    uint32_t n_iterations)
{
    const bool use_lt = scene.n_lights == 1u;  // raygen and shade store the light's distance beside every ray they make
    const LdsPlan plan = plan_lds(scene, stack_entries, n_nodes, scene_in_lds, stack_overflow != nullptr);
    static const int refill_env = [] {
        const char *e = pt::tuning_env("PT_REFILL");  // tuning experiments only
        return e ? std::atoi(e) : 0;
    }();
    const uint32_t refill_at = refill_env ? static_cast<uint32_t>(refill_env)
                                          : ((plan.mode == 0 || plan.mode == 4 || (plan.mode == 1 && plan.n_lds_nodes < n_nodes / 8u)) ? 32u : 16u);
    const BouncesKernel kernel = bounces_kernel(plan, use_lt);
    hipLaunchKernelGGL(kernel, dim3(resident_grid(kernel, plan.bytes, g, plan.block, "PT_BOUNCES_PER_CU")), dim3(plan.block), plan.bytes, s, scene, loop, n_nodes,
                       refill_at, stack_entries, plan.n_lds_nodes, plan.stride, params, plan.lds_stack_entries, stack_overflow, it_stride,
                       n_iterations, plan.n_lds_tris);
}

void launch_shade(hipStream_t s, LaunchShape g, const SceneView &scene, Regions regions, PathQueue in,
                  const float4 *hit_list, PathQueue out, float4 *samples, const BatchParams *params,
                  bool final_bounce, unsigned long long *unlit_skipped)
{
    const uint32_t mode = final_bounce ? kShadeFinal : (unlit_skipped ? kShadeBeforeLast : kShadeBounce);
    hipLaunchKernelGGL(k_shade, dim3(resident_grid(k_shade, 0, g, kBlock, "PT_SHADE_PER_CU")), dim3(kBlock), 0, s, scene, regions, in, hit_list, out,
                       samples, params, mode, unlit_skipped);
}

void launch_accumulate(hipStream_t s, LaunchShape g, const float4 *samples, float4 *sum, uint32_t n_pixels,
                       uint32_t n_iterations, const uint32_t *counts, uint32_t n_rows, uint32_t n_regions,
                       uint32_t pitch, unsigned long long *segments_total, unsigned long long *shade_totals,
                       unsigned long long *nan_counter, unsigned long long *culled_total, uint8_t *rgba8, const BatchParams *params,
                       uint32_t it_stride, bool culled)
{
    hipLaunchKernelGGL(rgba8 ? k_accumulate<true> : k_accumulate<false>, dim3(g.blocks), dim3(kBlock), 0, s, samples, sum, n_pixels,
                       n_iterations, counts, n_rows, n_regions, pitch, segments_total, shade_totals, nan_counter, culled_total,
                       rgba8, params, it_stride, culled ? 1u : 0u);
}

void launch_unpack_primary(hipStream_t s, LaunchShape g, uint32_t n, PathQueue q, const float2 *hits, int32_t *ids,
                           float *t, float *dirs, const DeviceCamera *cam)
{
    hipLaunchKernelGGL(k_unpack_primary, dim3(g.blocks), dim3(kBlock), 0, s, n, q, hits, ids, t, dirs,
                       cam ? *cam : DeviceCamera{}, cam ? 0u : 1u);
}

void launch_primary_normals(hipStream_t s, LaunchShape g, uint32_t n, const SceneView &scene, PathQueue q, const float2 *hits,
                            float *normals, const DeviceCamera &cam)
{
    hipLaunchKernelGGL(k_primary_normals, dim3(g.blocks), dim3(kBlock), 0, s, n, scene, q, hits, normals, cam);
}

void launch_pack_rays(hipStream_t s, LaunchShape g, uint32_t n, const float *origins, const float *dirs, PathQueue q,
                      uint32_t *counts_out, uint32_t region_size)
{
    hipLaunchKernelGGL(k_pack_rays, dim3(g.blocks), dim3(kBlock), 0, s, n, origins, dirs, q, counts_out, region_size);
}

void launch_tonemap(hipStream_t s, LaunchShape g, const float4 *sum, uint8_t *rgb, uint32_t width, uint32_t height)
{
    hipLaunchKernelGGL(k_tonemap, dim3(g.blocks), dim3(kBlock), 0, s, sum, rgb, width, height);
}

void launch_sample_surface(hipStream_t s, LaunchShape g, uint32_t n, const float *materials, const float *normals,
                           const float *wo, const float *xi, const uint32_t *seeds, float *wi_out, float *pdf_out,
                           float *f_out, uint32_t *flags_out, uint32_t *seeds_out)
{
    hipLaunchKernelGGL(k_sample_surface, dim3(g.blocks), dim3(kBlock), 0, s, n, materials, normals, wo, xi, seeds,
                       wi_out, pdf_out, f_out, flags_out, seeds_out);
}

}  // namespace pt
