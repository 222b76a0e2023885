// Binned-SAH BVH build on the GPU, level by level.  Every decision is made with
// the arithmetic of the host builder (bvh.cpp): centroid = 0.5 lo + 0.5 hi of
// the triangle's box, 16 bins per axis over the node's centroid box, bin index
// int((c - lo) * (16 / extent)), half areas and costs in double, first minimum in
// (axis, bin) order.  Box unions are min / max, which are exact and order-free, so
// the tree does not depend on the order in which the GPU visits the triangles;
// leaves list their triangles by index.
//
// One level = the nodes created by the level before, contiguous in the node
// array.  A node of up to kSmall triangles is handled by one wave: triangles in
// LDS, one lane per (axis, bin), partition by ballot.  Larger nodes are cut into
// chunks of 1024 triangle positions: a block bins its chunk in LDS and flushes the
// table with atomic min / max on order-preserving integer images of the floats,
// the per-node wave picks the split from the table, and a second kernel over the
// chunks moves the triangles, one atomic per block and side.  Children are
// compacted behind a prefix sum so that the node array stays in breadth-first
// order, which the final numbering relies on.
#include "bvh_device.h"

#include <cmath>
#include <cstring>
#include <vector>

#include "bvh.h"

namespace pt {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kSmall = 256;
constexpr uint32_t kBinWords = 13;  // box lo xyz, box hi xyz, centroid lo xyz, centroid hi xyz, count
constexpr uint32_t kNodeBinWords = 3 * kBins * kBinWords;
constexpr int32_t kLeaf = -1, kPending = -2;
constexpr uint32_t kErrDepth = 1u, kErrCoincident = 2u;
constexpr uint32_t kChunk = 1024;  // triangle positions per block in the kernels over positions
constexpr uint32_t kScanTile = 2048;

// A node while the tree is built.  left >= 0: inner, children left and left + 1.
struct NodeRec {
    float blo[3], bhi[3];  // box of the triangles
    float clo[3], chi[3];  // box of their centroids
    uint32_t begin, count;
    int32_t left;
    uint32_t slot;  // nodes of more than kSmall triangles: their bin table and split record
};
static_assert(sizeof(NodeRec) == 64, "one 64-byte record");

struct SplitRec {
    int32_t axis, bin;  // axis < 0: not split
    float lo, k;
    uint32_t begin, n_left, taken_left, taken_right;
};

struct Counters {
    uint32_t n_large_next, error, pad0, pad1;
};

__device__ __forceinline__ float min_h(float a, float b) { return b < a ? b : a; }  // std::min(a, b)
__device__ __forceinline__ float max_h(float a, float b) { return a < b ? b : a; }  // std::max(a, b)

// Order-preserving image of a float in the unsigned integers, and back.
__device__ __forceinline__ uint32_t enc(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

__device__ __forceinline__ float sel3(float x, float y, float z, int a) { return a == 0 ? x : (a == 1 ? y : z); }

__device__ __forceinline__ double half_area(const float *lo, const float *hi)
{
    const double dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ int bin_of(float c, float lo, float k)
{
    int b = static_cast<int>((c - lo) * k);
    return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
}

struct BoxAcc {
    float lo[3], hi[3];
    __device__ void reset()
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
    }
    __device__ void grow(float x, float y, float z)
    {
        lo[0] = min_h(lo[0], x), lo[1] = min_h(lo[1], y), lo[2] = min_h(lo[2], z);
        hi[0] = max_h(hi[0], x), hi[1] = max_h(hi[1], y), hi[2] = max_h(hi[2], z);
    }
    __device__ void grow(const float *l, const float *h)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = min_h(lo[a], l[a]);
            hi[a] = max_h(hi[a], h[a]);
        }
    }
};

__device__ __forceinline__ float wave_min(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = min_h(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = max_h(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int off)
{
    const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v));
    const uint32_t lo = __shfl_xor(static_cast<uint32_t>(u), off), hi = __shfl_xor(static_cast<uint32_t>(u >> 32), off);
    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

// ------------------------------------------------------------------ level 0

// Triangle boxes (two float4 per triangle: lo xyz + index, hi xyz) and the root's
// box and centroid box, the latter as 12 atomic maxima: words 0-2 ~enc(box lo),
// 3-5 enc(box hi), 6-8 ~enc(centroid lo), 9-11 enc(centroid hi).
__global__ void __launch_bounds__(256) k_prims_init(const float *__restrict__ vertices, const uint32_t *__restrict__ ids,
                                                     uint32_t n, float4 *__restrict__ prims, uint32_t *__restrict__ root_enc)
{
    BoxAcc box, cen;
    box.reset();
    cen.reset();
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        BoxAcc b;
        b.reset();
        for (int k = 0; k < 3; ++k) {
            const float *p = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + k]);
            b.grow(p[0], p[1], p[2]);
        }
        prims[2 * static_cast<size_t>(t)] = make_float4(b.lo[0], b.lo[1], b.lo[2], __uint_as_float(t));
        prims[2 * static_cast<size_t>(t) + 1] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.f);
        box.grow(b.lo, b.hi);
        cen.grow(0.5f * b.lo[0] + 0.5f * b.hi[0], 0.5f * b.lo[1] + 0.5f * b.hi[1], 0.5f * b.lo[2] + 0.5f * b.hi[2]);
    }
    for (int a = 0; a < 3; ++a) {
        const float v0 = wave_min(box.lo[a]), v1 = wave_max(box.hi[a]);
        const float v2 = wave_min(cen.lo[a]), v3 = wave_max(cen.hi[a]);
        if ((threadIdx.x & 63u) == 0u) {
            atomicMax(&root_enc[a], ~enc(v0));
            atomicMax(&root_enc[3 + a], enc(v1));
            atomicMax(&root_enc[6 + a], ~enc(v2));
            atomicMax(&root_enc[9 + a], enc(v3));
        }
    }
}

__global__ void k_root_finish(const uint32_t *__restrict__ root_enc, uint32_t n, NodeRec *__restrict__ nodes)
{
    NodeRec r;
    for (int a = 0; a < 3; ++a) {
        r.blo[a] = dec(~root_enc[a]);
        r.bhi[a] = dec(root_enc[3 + a]);
        r.clo[a] = dec(~root_enc[6 + a]);
        r.chi[a] = dec(root_enc[9 + a]);
    }
    r.begin = 0;
    r.count = n;
    r.left = kPending;
    r.slot = 0;
    nodes[0] = r;
}

// ------------------------------------------------------------------ large nodes: binning by chunk

__device__ __forceinline__ void bin_words(const float4 lo, const float4 hi, uint32_t w[12])
{
    const float cx = 0.5f * lo.x + 0.5f * hi.x, cy = 0.5f * lo.y + 0.5f * hi.y, cz = 0.5f * lo.z + 0.5f * hi.z;
    w[0] = ~enc(lo.x), w[1] = ~enc(lo.y), w[2] = ~enc(lo.z);
    w[3] = enc(hi.x), w[4] = enc(hi.y), w[5] = enc(hi.z);
    w[6] = ~enc(cx), w[7] = ~enc(cy), w[8] = ~enc(cz);
    w[9] = enc(cx), w[10] = enc(cy), w[11] = enc(cz);
}

// The large nodes of a level (more than kSmall triangles) are cut into chunks of kChunk
// positions, none of which crosses a node boundary: chunk_start[s] = first chunk of the
// node in slot s (exclusive prefix sums of the chunk counts), slot_node[s] = that node.
struct ChunkOfBlock {
    uint32_t node, slot, begin, end;
};
__device__ __forceinline__ ChunkOfBlock chunk_of_block(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ slot_node,
                                                       const uint32_t *__restrict__ chunk_start, uint32_t n_large)
{
    // last slot whose first chunk is <= blockIdx.x
    uint32_t lo = 0, hi = n_large;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= blockIdx.x)
            lo = mid;
        else
            hi = mid;
    }
    ChunkOfBlock c;
    c.slot = lo;
    c.node = slot_node[lo];
    const uint32_t first = nodes[c.node].begin, count = nodes[c.node].count;
    c.begin = first + (blockIdx.x - chunk_start[lo]) * kChunk;
    c.end = c.begin + kChunk < first + count ? c.begin + kChunk : first + count;
    return c;
}

__global__ void __launch_bounds__(256) k_chunk_counts(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ slot_node,
                                                       uint32_t n_large, uint32_t *__restrict__ chunks)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s < n_large) chunks[s] = (nodes[slot_node[s]].count + kChunk - 1u) / kChunk;
}

// Every triangle of a large node is added to the node's 3 x 16 bins: per chunk in LDS
// first, then one atomic per touched word of the node's table.
__global__ void __launch_bounds__(256) k_bin_large(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ slot_node,
                                                    const uint32_t *__restrict__ chunk_start, uint32_t n_large,
                                                    const float4 *__restrict__ cur, uint32_t *__restrict__ gbins)
{
    __shared__ uint32_t s[kNodeBinWords];
    const ChunkOfBlock ch = chunk_of_block(nodes, slot_node, chunk_start, n_large);
    const NodeRec nd = nodes[ch.node];
    for (uint32_t w = threadIdx.x; w < kNodeBinWords; w += 256u) s[w] = 0u;
    __syncthreads();
    for (uint32_t pos = ch.begin + threadIdx.x; pos < ch.end; pos += 256u) {
        const float4 lo = cur[2 * static_cast<size_t>(pos)], hi = cur[2 * static_cast<size_t>(pos) + 1];
        uint32_t w[12];
        bin_words(lo, hi, w);
        for (int a = 0; a < 3; ++a) {
            const float ext = nd.chi[a] - nd.clo[a];
            if (!(ext > 0.f)) continue;
            const float k = kBins / ext;
            const float c = 0.5f * sel3(lo.x, lo.y, lo.z, a) + 0.5f * sel3(hi.x, hi.y, hi.z, a);
            uint32_t *t = s + (a * kBins + bin_of(c, nd.clo[a], k)) * kBinWords;
            for (int i = 0; i < 12; ++i) atomicMax(&t[i], w[i]);
            atomicAdd(&t[12], 1u);
        }
    }
    __syncthreads();
    uint32_t *g = gbins + static_cast<size_t>(ch.slot) * kNodeBinWords;
    for (uint32_t w = threadIdx.x; w < kNodeBinWords; w += 256u) {
        const uint32_t v = s[w];
        if (v == 0u) continue;
        if (w % kBinWords == 12u)
            atomicAdd(&g[w], v);
        else
            atomicMax(&g[w], v);
    }
}

// ------------------------------------------------------------------ one wave per node: decide, split

// Per node of the level: leaf or split (bvh.cpp's rules), the children's boxes from
// the prefix / suffix unions of the bins, and for a node of up to kSmall triangles
// the partition itself.  Children go to child_scratch[2j], [2j + 1] and flags[j] = 1.
__global__ void __launch_bounds__(64) k_split_nodes(NodeRec *__restrict__ nodes, uint32_t lvl_start, uint32_t level,
                                                     const float4 *__restrict__ cur, float4 *__restrict__ nxt,
                                                     uint32_t *__restrict__ order, const uint32_t *__restrict__ gbins,
                                                     SplitRec *__restrict__ splits, NodeRec *__restrict__ child_scratch,
                                                     uint32_t *__restrict__ flags, Counters *__restrict__ ctr,
                                                     uint32_t leaf_max, double traversal_cost)
{
    __shared__ float4 s_lo[kSmall], s_hi[kSmall];
    __shared__ float s_bin[48 * kBinWords];
    __shared__ float s_res[32];
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    const uint32_t self = lvl_start + j;
    const NodeRec nd = nodes[self];
    const uint32_t n = nd.count, begin = nd.begin;
    const bool small = n <= kSmall;
    if (small)
        for (uint32_t i = lane; i < n; i += 64u) {
            s_lo[i] = cur[2 * static_cast<size_t>(begin + i)];
            s_hi[i] = cur[2 * static_cast<size_t>(begin + i) + 1];
        }
    __syncthreads();

    // a leaf lists its triangles by index
    auto make_leaf = [&]() {
        if (small)
            for (uint32_t i = lane; i < n; i += 64u) {
                const uint32_t id = __float_as_uint(s_lo[i].w);
                uint32_t rank = 0;
                for (uint32_t m = 0; m < n; ++m) rank += __float_as_uint(s_lo[m].w) < id ? 1u : 0u;
                order[begin + rank] = id;
            }
        if (lane == 0) {
            nodes[self].left = kLeaf;
            flags[j] = 0u;
        }
    };
    if (n <= 1u || (level >= 48u && n <= leaf_max)) {
        make_leaf();
        return;
    }
    if (level >= 48u) {  // the host builder continues with object-median splits here
        if (lane == 0) atomicOr(&ctr->error, kErrDepth);
        make_leaf();
        return;
    }

    // lane = (axis, bin): the bin's boxes and count
    const int axis = static_cast<int>(lane >> 4), b = static_cast<int>(lane & 15u);
    const float a_lo = axis < 3 ? sel3(nd.clo[0], nd.clo[1], nd.clo[2], axis) : 0.f;
    const float a_hi = axis < 3 ? sel3(nd.chi[0], nd.chi[1], nd.chi[2], axis) : 0.f;
    const float ext = a_hi - a_lo;
    const bool axis_ok = axis < 3 && ext > 0.f;
    const float k = axis_ok ? kBins / ext : 0.f;
    BoxAcc bb, bc;
    bb.reset();
    bc.reset();
    uint32_t cnt = 0;
    if (axis_ok) {
        if (small) {
            for (uint32_t i = 0; i < n; ++i) {
                const float4 lo = s_lo[i], hi = s_hi[i];
                const float c = 0.5f * sel3(lo.x, lo.y, lo.z, axis) + 0.5f * sel3(hi.x, hi.y, hi.z, axis);
                if (bin_of(c, a_lo, k) != b) continue;
                const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
                bb.grow(l3, h3);
                bc.grow(0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z);
                ++cnt;
            }
        } else {
            const size_t g0 = static_cast<size_t>(nd.slot) * kNodeBinWords + lane * kBinWords;
            uint32_t g[kBinWords];
            for (uint32_t w = 0; w < kBinWords; ++w) g[w] = gbins[g0 + w];
            cnt = g[12];
            for (int a = 0; a < 3; ++a) {
                bb.lo[a] = cnt ? dec(~g[a]) : INFINITY;
                bb.hi[a] = cnt ? dec(g[3 + a]) : -INFINITY;
                bc.lo[a] = cnt ? dec(~g[6 + a]) : INFINITY;
                bc.hi[a] = cnt ? dec(g[9 + a]) : -INFINITY;
            }
        }
    }
    if (lane < 48u) {
        float *t = s_bin + lane * kBinWords;
        for (int a = 0; a < 3; ++a) {
            t[a] = bb.lo[a];
            t[3 + a] = bb.hi[a];
            t[6 + a] = bc.lo[a];
            t[9 + a] = bc.hi[a];
        }
        t[12] = __uint_as_float(cnt);
    }
    __syncthreads();

    // lane = (axis, bin): the split after this bin
    BoxAcc lb, lc, rb, rc;
    lb.reset(), lc.reset(), rb.reset(), rc.reset();
    uint32_t ln = 0, rn = 0;
    double cost = INFINITY;
    if (axis_ok && b < kBins - 1) {
        for (int m = 0; m < kBins; ++m) {
            const float *t = s_bin + (axis * kBins + m) * kBinWords;
            if (m <= b) {
                lb.grow(t, t + 3);
                lc.grow(t + 6, t + 9);
                ln += __float_as_uint(t[12]);
            } else {
                rb.grow(t, t + 3);
                rc.grow(t + 6, t + 9);
                rn += __float_as_uint(t[12]);
            }
        }
        if (ln != 0u && rn != 0u) {
            const double c = half_area(lb.lo, lb.hi) * ln + half_area(rb.lo, rb.hi) * rn;
            if (c < double(INFINITY)) cost = c;
        }
    }
    double best = cost;
    uint32_t key = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = shfl_xor_f64(best, off);
        const uint32_t ok = __shfl_xor(key, off);
        if (oc < best || (oc == best && ok < key)) {
            best = oc;
            key = ok;
        }
    }
    const bool found = best < double(INFINITY);
    const double area = half_area(nd.blo, nd.bhi);
    if (n <= leaf_max && (!found || double(n) * area <= traversal_cost * area + best)) {
        make_leaf();
        return;
    }

    NodeRec c0, c1;
    c0.left = c1.left = kPending;
    c0.slot = c1.slot = 0;
    if (!found) {
        // All centroids coincide: halves by triangle index.
        if (!small) {
            if (lane == 0) {
                atomicOr(&ctr->error, kErrCoincident);
                splits[nd.slot].axis = -1;
            }
            make_leaf();
            return;
        }
        const uint32_t n_left = n / 2u;
        BoxAcc l, r;
        l.reset(), r.reset();
        for (uint32_t i = lane; i < n; i += 64u) {
            const float4 lo = s_lo[i], hi = s_hi[i];
            const uint32_t id = __float_as_uint(lo.w);
            uint32_t rank = 0;
            for (uint32_t m = 0; m < n; ++m) rank += __float_as_uint(s_lo[m].w) < id ? 1u : 0u;
            nxt[2 * static_cast<size_t>(begin + rank)] = lo;
            nxt[2 * static_cast<size_t>(begin + rank) + 1] = hi;
            const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
            if (rank < n_left)
                l.grow(l3, h3);
            else
                r.grow(l3, h3);
        }
        for (int a = 0; a < 3; ++a) {
            c0.blo[a] = wave_min(l.lo[a]), c0.bhi[a] = wave_max(l.hi[a]);
            c1.blo[a] = wave_min(r.lo[a]), c1.bhi[a] = wave_max(r.hi[a]);
            c0.clo[a] = c1.clo[a] = nd.clo[a];
            c0.chi[a] = c1.chi[a] = nd.chi[a];
        }
        c0.begin = begin, c0.count = n_left;
        c1.begin = begin + n_left, c1.count = n - n_left;
    } else {
        // the lane that holds the best split publishes it
        if (lane == key) {
            for (int a = 0; a < 3; ++a) {
                s_res[a] = lb.lo[a], s_res[3 + a] = lb.hi[a], s_res[6 + a] = lc.lo[a], s_res[9 + a] = lc.hi[a];
                s_res[12 + a] = rb.lo[a], s_res[15 + a] = rb.hi[a], s_res[18 + a] = rc.lo[a], s_res[21 + a] = rc.hi[a];
            }
            s_res[24] = __uint_as_float(ln);
            s_res[25] = a_lo;
            s_res[26] = k;
        }
        __syncthreads();
        for (int a = 0; a < 3; ++a) {
            c0.blo[a] = s_res[a], c0.bhi[a] = s_res[3 + a], c0.clo[a] = s_res[6 + a], c0.chi[a] = s_res[9 + a];
            c1.blo[a] = s_res[12 + a], c1.bhi[a] = s_res[15 + a], c1.clo[a] = s_res[18 + a], c1.chi[a] = s_res[21 + a];
        }
        const uint32_t n_left = __float_as_uint(s_res[24]);
        const float s_lo_a = s_res[25], s_k = s_res[26];
        const int s_axis = static_cast<int>(key >> 4), s_bin_i = static_cast<int>(key & 15u);
        c0.begin = begin, c0.count = n_left;
        c1.begin = begin + n_left, c1.count = n - n_left;
        if (small) {
            uint32_t off_l = 0, off_r = 0;
            for (uint32_t q = 0; q < n; q += 64u) {
                const uint32_t i = q + lane;
                const bool in = i < n;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                bool left = false;
                if (in) {
                    lo = s_lo[i], hi = s_hi[i];
                    const float c = 0.5f * sel3(lo.x, lo.y, lo.z, s_axis) + 0.5f * sel3(hi.x, hi.y, hi.z, s_axis);
                    left = bin_of(c, s_lo_a, s_k) <= s_bin_i;
                }
                const unsigned long long ml = __ballot(in && left), mr = __ballot(in && !left);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (in) {
                    const uint32_t dst = left ? begin + off_l + static_cast<uint32_t>(__popcll(ml & below))
                                              : begin + n_left + off_r + static_cast<uint32_t>(__popcll(mr & below));
                    nxt[2 * static_cast<size_t>(dst)] = lo;
                    nxt[2 * static_cast<size_t>(dst) + 1] = hi;
                }
                off_l += static_cast<uint32_t>(__popcll(ml));
                off_r += static_cast<uint32_t>(__popcll(mr));
            }
        } else if (lane == 0) {
            SplitRec sp;
            sp.axis = s_axis, sp.bin = s_bin_i, sp.lo = s_lo_a, sp.k = s_k;
            sp.begin = begin, sp.n_left = n_left, sp.taken_left = 0u, sp.taken_right = 0u;
            splits[nd.slot] = sp;
        }
    }
    if (lane == 0) {
        child_scratch[2 * static_cast<size_t>(j)] = c0;
        child_scratch[2 * static_cast<size_t>(j) + 1] = c1;
        flags[j] = 1u;
    }
}

// ------------------------------------------------------------------ large nodes: partition by chunk

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total, uint32_t *s_wave);

// The triangles of a chunk of a large node that is split move to their side of the node's
// range in the other buffer: the block counts its left and right triangles, reserves a run
// on each side with one atomic, and fills the two runs in position order.
__global__ void __launch_bounds__(256) k_partition_large(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ slot_node,
                                                          const uint32_t *__restrict__ chunk_start, uint32_t n_large,
                                                          const float4 *__restrict__ cur, float4 *__restrict__ nxt,
                                                          SplitRec *__restrict__ splits)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base[2];
    const ChunkOfBlock ch = chunk_of_block(nodes, slot_node, chunk_start, n_large);
    const SplitRec sp = splits[ch.slot];
    if (sp.axis < 0) return;
    constexpr uint32_t kPer = kChunk / 256u;
    float4 lo[kPer], hi[kPer];
    bool in[kPer], left[kPer];
    uint32_t n_l = 0, n_r = 0;
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t pos = ch.begin + threadIdx.x * kPer + q;
        in[q] = pos < ch.end;
        left[q] = false;
        if (in[q]) {
            lo[q] = cur[2 * static_cast<size_t>(pos)], hi[q] = cur[2 * static_cast<size_t>(pos) + 1];
            const float c = 0.5f * sel3(lo[q].x, lo[q].y, lo[q].z, sp.axis) + 0.5f * sel3(hi[q].x, hi[q].y, hi[q].z, sp.axis);
            left[q] = bin_of(c, sp.lo, sp.k) <= sp.bin;
            n_l += left[q] ? 1u : 0u;
            n_r += left[q] ? 0u : 1u;
        }
    }
    uint32_t total_l, total_r;
    uint32_t at_l = block_exclusive_scan(n_l, &total_l, s_wave);
    uint32_t at_r = block_exclusive_scan(n_r, &total_r, s_wave);
    if (threadIdx.x == 0) {
        s_base[0] = total_l ? atomicAdd(&splits[ch.slot].taken_left, total_l) : 0u;
        s_base[1] = total_r ? atomicAdd(&splits[ch.slot].taken_right, total_r) : 0u;
    }
    __syncthreads();
    at_l += sp.begin + s_base[0];
    at_r += sp.begin + sp.n_left + s_base[1];
    for (uint32_t q = 0; q < kPer; ++q) {
        if (!in[q]) continue;
        const uint32_t dst = left[q] ? at_l++ : at_r++;
        nxt[2 * static_cast<size_t>(dst)] = lo[q];
        nxt[2 * static_cast<size_t>(dst) + 1] = hi[q];
    }
}

// ------------------------------------------------------------------ exclusive prefix sum of 32-bit counts

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total, uint32_t *s_wave /* [4] */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off);
        if (static_cast<int>(lane) >= off) inc += u;
    }
    __syncthreads();  // s_wave may still be read from the call before
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 4u; ++w) {
        if (w < wave) before += s_wave[w];
        all += s_wave[w];
    }
    *total = all;
    return before + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_reduce(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile;
    uint32_t v = 0;
    for (uint32_t i = 0; i < kScanTile / 256u; ++i) {
        const uint32_t p = base + threadIdx.x * (kScanTile / 256u) + i;
        if (p < n) v += in[p];
    }
    uint32_t total;
    block_exclusive_scan(v, &total, s_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one block: sums[] -> its exclusive prefix sums in place, the grand total to *total
__global__ void __launch_bounds__(256) k_scan_top(uint32_t *__restrict__ sums, uint32_t n_blocks, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 256u) {
        const uint32_t p = base + threadIdx.x;
        const uint32_t v = p < n_blocks ? sums[p] : 0u;
        uint32_t tile;
        const uint32_t ex = block_exclusive_scan(v, &tile, s_wave);
        if (p < n_blocks) sums[p] = carry + ex;
        carry += tile;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ in, uint32_t n,
                                                     const uint32_t *__restrict__ sums, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * (kScanTile / 256u);
    uint32_t item[kScanTile / 256u];
    uint32_t v = 0;
    for (uint32_t i = 0; i < kScanTile / 256u; ++i) {
        item[i] = base + i < n ? in[base + i] : 0u;
        v += item[i];
    }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_exclusive_scan(v, &total, s_wave);
    for (uint32_t i = 0; i < kScanTile / 256u; ++i) {
        if (base + i < n) out[base + i] = run;
        run += item[i];
    }
}

// ------------------------------------------------------------------ children behind the level, in order

__global__ void __launch_bounds__(256) k_compact_children(NodeRec *__restrict__ nodes, uint32_t lvl_start, uint32_t lvl_count,
                                                           uint32_t next_start, const NodeRec *__restrict__ child_scratch,
                                                           const uint32_t *__restrict__ flags,
                                                           const uint32_t *__restrict__ rank, Counters *__restrict__ ctr,
                                                           uint32_t *__restrict__ slot_node)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= lvl_count || flags[j] == 0u) return;
    const uint32_t dst = next_start + 2u * rank[j];
    for (uint32_t c = 0; c < 2u; ++c) {
        NodeRec r = child_scratch[2 * static_cast<size_t>(j) + c];
        if (r.count > kSmall) {
            r.slot = atomicAdd(&ctr->n_large_next, 1u);
            slot_node[r.slot] = dst + c;
        }
        nodes[dst + c] = r;
    }
    nodes[lvl_start + j].left = static_cast<int32_t>(dst);
}

// ------------------------------------------------------------------ numbering and output

__global__ void __launch_bounds__(256) k_inner_flags(const NodeRec *__restrict__ nodes, uint32_t n, uint32_t *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = nodes[i].left >= 0 ? 1u : 0u;
}

// inner nodes in the subtree of each node of one level (levels are processed deepest first)
__global__ void __launch_bounds__(256) k_subtree_sizes(const NodeRec *__restrict__ nodes, uint32_t start, uint32_t count,
                                                        uint32_t *__restrict__ sizes)
{
    const uint32_t i = start + blockIdx.x * 256u + threadIdx.x;
    if (i >= start + count) return;
    const int32_t l = nodes[i].left;
    sizes[i] = l >= 0 ? 1u + sizes[l] + sizes[l + 1] : 0u;
}

// The first `top` inner nodes in breadth-first order keep that rank as their index.  The
// inner nodes just below them, in breadth-first order, are the roots of the subtrees that
// follow in depth-first order: root_size[rank - top] = size of that subtree.
__global__ void __launch_bounds__(256) k_root_sizes(const NodeRec *__restrict__ nodes, uint32_t n, const uint32_t *__restrict__ bfs,
                                                     const uint32_t *__restrict__ sizes, uint32_t top,
                                                     uint32_t *__restrict__ root_size)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t l = nodes[i].left;
    if (l < 0 || bfs[i] >= top) return;
    for (int c = 0; c < 2; ++c)
        if (nodes[l + c].left >= 0 && bfs[l + c] >= top) root_size[bfs[l + c] - top] = sizes[l + c];
}

__global__ void k_root_offsets(uint32_t *__restrict__ root_size, uint32_t n_roots, uint32_t top)
{
    uint32_t run = top;
    for (uint32_t i = 0; i < n_roots; ++i) {
        const uint32_t s = root_size[i];
        root_size[i] = run;
        run += s;
    }
}

// final index of the inner children of one level's inner nodes (levels top-down; index[0] = 0)
__global__ void __launch_bounds__(256) k_assign_index(const NodeRec *__restrict__ nodes, uint32_t start, uint32_t count,
                                                       const uint32_t *__restrict__ bfs, const uint32_t *__restrict__ sizes,
                                                       const uint32_t *__restrict__ root_offset, uint32_t top,
                                                       uint32_t *__restrict__ index)
{
    const uint32_t i = start + blockIdx.x * 256u + threadIdx.x;
    if (i >= start + count) return;
    const int32_t l = nodes[i].left;
    if (l < 0) return;
    const uint32_t mine = index[i];
    for (int c = 0; c < 2; ++c) {
        const uint32_t ch = static_cast<uint32_t>(l + c);
        if (nodes[ch].left < 0) continue;
        uint32_t v;
        if (bfs[ch] < top)
            v = bfs[ch];
        else if (bfs[i] < top)
            v = root_offset[bfs[ch] - top];
        else
            v = c == 0 ? mine + 1u : mine + 1u + sizes[l];
        index[ch] = v;
    }
}

__device__ __forceinline__ int32_t leaf_code(uint32_t first, uint32_t count)
{
    return ~static_cast<int32_t>((first << 3) | count);
}

__global__ void __launch_bounds__(256) k_emit_nodes(const NodeRec *__restrict__ nodes, uint32_t n, const uint32_t *__restrict__ index,
                                                     float4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t l = nodes[i].left;
    if (l < 0) return;
    const NodeRec a = nodes[l], b = nodes[l + 1];
    const int32_t ca = a.left >= 0 ? static_cast<int32_t>(index[l]) : leaf_code(a.begin, a.count);
    const int32_t cb = b.left >= 0 ? static_cast<int32_t>(index[l + 1]) : leaf_code(b.begin, b.count);
    float4 *o = out + 4 * static_cast<size_t>(index[i]);
    o[0] = make_float4(a.blo[0], a.blo[1], a.blo[2], a.bhi[0]);
    o[1] = make_float4(a.bhi[1], a.bhi[2], b.blo[0], b.blo[1]);
    o[2] = make_float4(b.blo[2], b.bhi[0], b.bhi[1], b.bhi[2]);
    o[3] = make_float4(__int_as_float(ca), __int_as_float(cb), 0.f, 0.f);
}

// A scene whose root is a leaf: a root node with that leaf and an empty one (bvh.cpp does the same).
__global__ void k_emit_leaf_root(const NodeRec *__restrict__ nodes, float4 *__restrict__ out)
{
    const NodeRec a = nodes[0];
    out[0] = make_float4(a.blo[0], a.blo[1], a.blo[2], a.bhi[0]);
    out[1] = make_float4(a.bhi[1], a.bhi[2], a.blo[0], a.blo[1]);
    out[2] = make_float4(a.blo[2], a.bhi[0], a.bhi[1], a.bhi[2]);
    out[3] = make_float4(__int_as_float(leaf_code(a.begin, a.count)), __int_as_float(leaf_code(0, 0)), 0.f, 0.f);
}

// The leaf-ordered triangle records (bvh.h), n and d0 exactly as pathtracer.frag.glsl:399-404
// evaluates them: one rounding per operation.
__global__ void __launch_bounds__(256) k_emit_triangles(const uint32_t *__restrict__ order, uint32_t n,
                                                         const float *__restrict__ vertices, const uint32_t *__restrict__ ids,
                                                         float4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = order[i];
    const float *v0 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 0]);
    const float *v1 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 1]);
    const float *v2 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 2]);
    const float e0x = v1[0] - v0[0], e0y = v1[1] - v0[1], e0z = v1[2] - v0[2];
    const float e1x = v2[0] - v0[0], e1y = v2[1] - v0[1], e1z = v2[2] - v0[2];
    const float cx = e0y * e1z - e0z * e1y;
    const float cy = e0z * e1x - e0x * e1z;
    const float cz = e0x * e1y - e0y * e1x;
    const float inv_len = 1.0f / sqrtf(cx * cx + (cy * cy + cz * cz));
    const float nx = cx * inv_len, ny = cy * inv_len, nz = cz * inv_len;
    const float d0 = nx * v0[0] + (ny * v0[1] + nz * v0[2]);
    float4 *o = out + 4 * static_cast<size_t>(i);
    o[0] = make_float4(v0[0], v0[1], v0[2], d0);
    o[1] = make_float4(v1[0], v1[1], v1[2], nx);
    o[2] = make_float4(v2[0], v2[1], v2[2], ny);
    o[3] = make_float4(nz, __uint_as_float(t), 0.f, 0.f);
}

// sum over nodes of (inner ? 1 : triangles) * half area, in units of the root's half area
__global__ void __launch_bounds__(256) k_sah_cost(const NodeRec *__restrict__ nodes, uint32_t n, double *__restrict__ sum)
{
    __shared__ double s[256];
    double v = 0.0;
    const double root_area = half_area(nodes[0].blo, nodes[0].bhi);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const NodeRec nd = nodes[i];
        if (root_area > 0.0) v += (nd.left >= 0 ? 1.0 : double(nd.count)) * half_area(nd.blo, nd.bhi) / root_area;
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(sum, s[0]);
}

// Bit 30 of a record's triangle index marks surfaces that are not purely diffuse
// (metallic > 0 or roughness < 1 reach the microfacet lobes); the shade kernel sorts on it.
__global__ void __launch_bounds__(256) k_mark_classes(float4 *__restrict__ tris, uint32_t n,
                                                       const uint32_t *__restrict__ material_map,
                                                       const float *__restrict__ materials, uint32_t n_materials)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 rec = tris[4 * static_cast<size_t>(i) + 3];
    const uint32_t t = __float_as_uint(rec.y);
    const uint32_t m = material_map[t];
    bool diffuse = false;
    if (m < n_materials) diffuse = materials[8 * static_cast<size_t>(m) + 4] <= 0.f && materials[8 * static_cast<size_t>(m) + 3] >= 1.f;
    if (!diffuse) {
        rec.y = __uint_as_float(t | kHitClassBit);
        tris[4 * static_cast<size_t>(i) + 3] = rec;
    }
}

// The shading record of every original triangle (device_types.h ShadeTriangle): corner
// positions, the corner normals of vertex-data record int(indices.w) (pathtracer.frag.glsl:
// 437,447-449) and the material index.
__global__ void __launch_bounds__(256) k_shade_records(const float *__restrict__ vertices, const uint32_t *__restrict__ ids,
                                                        const float *__restrict__ indices, const float *__restrict__ vertex_data,
                                                        const uint32_t *__restrict__ material_map, uint32_t n,
                                                        float4 *__restrict__ out)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float *v0 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 0]);
    const float *v1 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 1]);
    const float *v2 = vertices + 3 * static_cast<size_t>(ids[3 * static_cast<size_t>(t) + 2]);
    const float wf = indices[4 * static_cast<size_t>(t) + 3];
    const size_t data = (wf >= 0.f && wf < static_cast<float>(n)) ? static_cast<size_t>(wf) : t;
    const float *n1 = vertex_data + 6 * (3 * data + 0);
    const float *n2 = vertex_data + 6 * (3 * data + 1);
    const float *n3 = vertex_data + 6 * (3 * data + 2);
    float4 *o = out + 5 * static_cast<size_t>(t);
    o[0] = make_float4(v0[0], v0[1], v0[2], n1[0]);
    o[1] = make_float4(v1[0], v1[1], v1[2], n1[1]);
    o[2] = make_float4(v2[0], v2[1], v2[2], n1[2]);
    o[3] = make_float4(n2[0], n2[1], n2[2], __uint_as_float(material_map[t]));
    o[4] = make_float4(n3[0], n3[1], n3[2], 0.f);
}

// ------------------------------------------------------------------ host side

struct Buffers {
    std::vector<void *> all;
    ~Buffers()
    {
        for (void *p : all) (void)hipFree(p);
    }
    template <typename T>
    bool get(T **p, size_t count)
    {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count * sizeof(T), 256)) != hipSuccess) return false;
        all.push_back(q);
        *p = static_cast<T *>(q);
        return true;
    }
};

uint32_t blocks_for(uint64_t n, uint32_t per_block) { return static_cast<uint32_t>((n + per_block - 1) / per_block); }

// out[i] = sum of in[0 .. i), *d_total = sum of all (device memory)
void exclusive_scan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *sums, uint32_t *d_total, hipStream_t st)
{
    const uint32_t nb = blocks_for(n, kScanTile);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(256), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, sums, nb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, in, n, sums, out);
}

}  // namespace

void mark_triangle_classes(void *d_triangles, uint32_t n_triangles, const uint32_t *d_material_map, const float *d_materials,
                           uint32_t n_materials, hipStream_t stream)
{
    if (n_triangles == 0) return;
    hipLaunchKernelGGL(k_mark_classes, dim3((n_triangles + 255u) / 256u), dim3(256), 0, stream,
                       static_cast<float4 *>(d_triangles), n_triangles, d_material_map, d_materials, n_materials);
}

void build_shade_records(const float *d_vertices, const uint32_t *d_tri_vertex_ids, const float *d_indices,
                         const float *d_vertex_data, const uint32_t *d_material_map, uint32_t n_triangles, void *d_out,
                         hipStream_t stream)
{
    if (n_triangles == 0) return;
    hipLaunchKernelGGL(k_shade_records, dim3((n_triangles + 255u) / 256u), dim3(256), 0, stream, d_vertices, d_tri_vertex_ids,
                       d_indices, d_vertex_data, d_material_map, n_triangles, static_cast<float4 *>(d_out));
}

// ------------------------------------------------------------------ four-wide nodes from the binary tree

namespace {

// level[i] = distance of inner node i from the root; one pass per level, top-down: the nodes of level p hand
// p + 1 to their inner children.  (Any numbering of the node array will do.)
__global__ void __launch_bounds__(256) k_wide_levels(const float4 *__restrict__ nodes, uint32_t n, uint32_t p, uint32_t *__restrict__ level)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || level[i] != p) return;
    const float4 q3 = nodes[4 * static_cast<size_t>(i) + 3];
    const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
    if (c0 >= 0) level[c0] = p + 1u;
    if (c1 >= 0) level[c1] = p + 1u;
}

__global__ void __launch_bounds__(256) k_wide_flags(const uint32_t *__restrict__ level, uint32_t n, uint32_t *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = (level[i] != 0xFFFFFFFFu && (level[i] & 1u) == 0u) ? 1u : 0u;
}

// One 128-byte record per inner node of even level: the up to four nodes two levels down (a child that is a
// leaf stays as it is), as
//   q0 = lo.x of the four, q1 = hi.x, q2 = lo.y, q3 = hi.y, q4 = lo.z, q5 = hi.z, q6 = the four child codes, q7 unused.
// A child code >= 0 is the index of a four-wide node, < 0 a leaf code as in the binary tree.  A slot without a
// child has the box (+inf, -inf), which no ray meets.
__global__ void __launch_bounds__(256) k_wide_emit(const float4 *__restrict__ nodes, uint32_t n, const uint32_t *__restrict__ flags,
                                                    const uint32_t *__restrict__ widx, float4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || flags[i] == 0u) return;
    float lo[4][3], hi[4][3];
    int code[4];
    uint32_t m = 0;
    auto take = [&](const float4 *rec, int c, bool may_expand) {
        // child c of the binary node `rec`: lo / hi as BvhNode packs them
        const float4 a = rec[0], b = rec[1], d = rec[2], q3 = rec[3];
        const float blo[3] = {c == 0 ? a.x : b.z, c == 0 ? a.y : b.w, c == 0 ? a.z : d.x};
        const float bhi[3] = {c == 0 ? a.w : d.y, c == 0 ? b.x : d.z, c == 0 ? b.y : d.w};
        const int cc = c == 0 ? __float_as_int(q3.x) : __float_as_int(q3.y);
        (void)may_expand;
        for (int k = 0; k < 3; ++k) lo[m][k] = blo[k], hi[m][k] = bhi[k];
        code[m] = cc;
        ++m;
    };
    const float4 *me = nodes + 4 * static_cast<size_t>(i);
    const float4 q3 = me[3];
    const int c01[2] = {__float_as_int(q3.x), __float_as_int(q3.y)};
    for (int c = 0; c < 2; ++c) {
        if (c01[c] >= 0) {
            const float4 *child = nodes + 4 * static_cast<size_t>(c01[c]);
            take(child, 0, false);
            take(child, 1, false);
        } else {
            take(me, c, false);
        }
    }
    // inner grandchildren are four-wide nodes themselves: their codes become indices into this array
    for (uint32_t k = 0; k < m; ++k)
        if (code[k] >= 0) code[k] = static_cast<int>(widx[code[k]]);
    const float inf = __builtin_inff();
    for (uint32_t k = m; k < 4u; ++k) {
        for (int a = 0; a < 3; ++a) lo[k][a] = inf, hi[k][a] = -inf;
        code[k] = static_cast<int>(0x80000000u);
    }
    float4 *o = out + 8 * static_cast<size_t>(widx[i]);
    o[0] = make_float4(lo[0][0], lo[1][0], lo[2][0], lo[3][0]);
    o[1] = make_float4(hi[0][0], hi[1][0], hi[2][0], hi[3][0]);
    o[2] = make_float4(lo[0][1], lo[1][1], lo[2][1], lo[3][1]);
    o[3] = make_float4(hi[0][1], hi[1][1], hi[2][1], hi[3][1]);
    o[4] = make_float4(lo[0][2], lo[1][2], lo[2][2], lo[3][2]);
    o[5] = make_float4(hi[0][2], hi[1][2], hi[2][2], hi[3][2]);
    o[6] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]), __int_as_float(code[3]));
    o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

bool build_wide_nodes(const void *d_nodes, uint32_t n_nodes, uint32_t depth, hipStream_t st, WideBvh *out, std::string *error)
{
    *out = WideBvh();
    if (n_nodes == 0 || !d_nodes) {
        *error = "four-wide nodes: no tree";
        return false;
    }
    Buffers buf;
    uint32_t *level = nullptr, *flags = nullptr, *widx = nullptr, *sums = nullptr, *d_total = nullptr;
    if (!(buf.get(&level, n_nodes) && buf.get(&flags, n_nodes) && buf.get(&widx, n_nodes) &&
          buf.get(&sums, n_nodes / kScanTile + 2) && buf.get(&d_total, 4))) {
        *error = "four-wide nodes: out of device memory";
        return false;
    }
    const float4 *nodes = static_cast<const float4 *>(d_nodes);
    auto fail = [&](hipError_t e) {
        *error = std::string("four-wide nodes: ") + hipGetErrorString(e);
        return false;
    };
    hipError_t e = hipMemsetAsync(level, 0xFF, static_cast<size_t>(n_nodes) * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(level, 0, 4, st);  // the root
    if (e != hipSuccess) return fail(e);
    for (uint32_t p = 0; p + 1 < std::max(depth, 1u); ++p)
        hipLaunchKernelGGL(k_wide_levels, dim3(blocks_for(n_nodes, 256)), dim3(256), 0, st, nodes, n_nodes, p, level);
    hipLaunchKernelGGL(k_wide_flags, dim3(blocks_for(n_nodes, 256)), dim3(256), 0, st, level, n_nodes, flags);
    exclusive_scan(flags, n_nodes, widx, sums, d_total, st);
    uint32_t n_wide = 0;
    e = hipMemcpyAsync(&n_wide, d_total, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(e);
    if (n_wide == 0) {
        *error = "four-wide nodes: the tree has no root";
        return false;
    }
    void *d_wide = nullptr;
    e = hipMalloc(&d_wide, static_cast<size_t>(n_wide) * 128);
    if (e != hipSuccess) return fail(e);
    hipLaunchKernelGGL(k_wide_emit, dim3(blocks_for(n_nodes, 256)), dim3(256), 0, st, nodes, n_nodes, flags, widx,
                       static_cast<float4 *>(d_wide));
    e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(d_wide);
        return fail(e);
    }
    out->d_nodes = d_wide;
    out->n_nodes = n_wide;
    out->depth = (depth + 1u) / 2u;
    return true;
}

#define PT_BVH_TRY(expr)                                                                        \
    do {                                                                                        \
        const hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                                 \
            *error = std::string("device BVH build: ") + #expr + ": " + hipGetErrorString(e_); \
            return false;                                                                       \
        }                                                                                       \
    } while (0)

bool build_bvh_device(const float *d_vertices, const uint32_t *d_ids, uint32_t T, const DeviceBvhParams &params,
                      hipStream_t st, DeviceBvh *out, std::string *error)
{
    *out = DeviceBvh();
    if (T == 0) {
        *error = "device BVH build: no triangles";
        return false;
    }
    if (params.leaf_max > 7u) {
        *error = "device BVH build: a leaf holds at most 7 triangles";
        return false;
    }
    Buffers buf;
    float4 *prims[2] = {nullptr, nullptr};
    uint32_t *slot_node = nullptr, *chunks = nullptr, *chunk_start = nullptr, *order = nullptr, *flags = nullptr, *rank = nullptr, *sums = nullptr, *gbins = nullptr;
    uint32_t *root_enc = nullptr, *d_total = nullptr, *sizes = nullptr, *index = nullptr, *root_size = nullptr;
    NodeRec *nodes = nullptr, *child_scratch = nullptr;
    SplitRec *splits = nullptr;
    Counters *ctr = nullptr;
    double *d_cost = nullptr;
    const size_t max_nodes = 2 * static_cast<size_t>(T) + 2;
    const size_t max_large = static_cast<size_t>(T) / kSmall + 2;
    const uint32_t top = params.top_breadth_first;
    bool ok = buf.get(&prims[0], 2 * static_cast<size_t>(T)) && buf.get(&prims[1], 2 * static_cast<size_t>(T)) &&
              buf.get(&slot_node, max_large) && buf.get(&chunks, max_large) && buf.get(&chunk_start, max_large) && buf.get(&order, T) && buf.get(&flags, max_nodes) && buf.get(&rank, max_nodes) &&
              buf.get(&sums, max_nodes / kScanTile + 2) && buf.get(&gbins, max_large * kNodeBinWords) &&
              buf.get(&root_enc, 16) && buf.get(&d_total, 4) && buf.get(&sizes, max_nodes) && buf.get(&index, max_nodes) &&
              buf.get(&root_size, 2 * static_cast<size_t>(top) + 4) && buf.get(&nodes, max_nodes) &&
              buf.get(&child_scratch, 2 * static_cast<size_t>(T) + 2) && buf.get(&splits, max_large) && buf.get(&ctr, 1) &&
              buf.get(&d_cost, 1);
    if (!ok) {
        *error = "device BVH build: out of device memory";
        return false;
    }

    PT_BVH_TRY(hipMemsetAsync(root_enc, 0, 64, st));
    PT_BVH_TRY(hipMemsetAsync(slot_node, 0, 4, st));   // the root is the large node in slot 0 ...
    PT_BVH_TRY(hipMemsetAsync(chunk_start, 0, 4, st));  // ... and its chunks are all there are
    PT_BVH_TRY(hipMemsetAsync(ctr, 0, sizeof(Counters), st));
    hipLaunchKernelGGL(k_prims_init, dim3(std::min<uint32_t>(1024, blocks_for(T, 256))), dim3(256), 0, st, d_vertices, d_ids, T,
                       prims[0], root_enc);
    hipLaunchKernelGGL(k_root_finish, dim3(1), dim3(1), 0, st, root_enc, T, nodes);

    struct Level {
        uint32_t start, count;
    };
    std::vector<Level> levels;
    uint32_t lvl_start = 0, lvl_count = 1, n_large = T > kSmall ? 1u : 0u, level = 0, depth = 0;
    int cur = 0;
    uint32_t n_chunks = blocks_for(T, kChunk);
    while (lvl_count > 0) {
        levels.push_back(Level{lvl_start, lvl_count});
        if (n_large) {
            PT_BVH_TRY(hipMemsetAsync(gbins, 0, static_cast<size_t>(n_large) * kNodeBinWords * 4, st));
            hipLaunchKernelGGL(k_bin_large, dim3(n_chunks), dim3(256), 0, st, nodes, slot_node, chunk_start, n_large, prims[cur],
                               gbins);
        }
        PT_BVH_TRY(hipMemsetAsync(&ctr->n_large_next, 0, 4, st));
        hipLaunchKernelGGL(k_split_nodes, dim3(lvl_count), dim3(64), 0, st, nodes, lvl_start, level, prims[cur], prims[cur ^ 1],
                           order, gbins, splits, child_scratch, flags, ctr, params.leaf_max, params.traversal_cost);
        if (n_large)
            hipLaunchKernelGGL(k_partition_large, dim3(n_chunks), dim3(256), 0, st, nodes, slot_node, chunk_start, n_large,
                               prims[cur], prims[cur ^ 1], splits);
        exclusive_scan(flags, lvl_count, rank, sums, d_total, st);
        const uint32_t next_start = lvl_start + lvl_count;
        hipLaunchKernelGGL(k_compact_children, dim3(blocks_for(lvl_count, 256)), dim3(256), 0, st, nodes, lvl_start, lvl_count,
                           next_start, child_scratch, flags, rank, ctr, slot_node);
        uint32_t n_split = 0;
        Counters h{};
        PT_BVH_TRY(hipMemcpyAsync(&n_split, d_total, 4, hipMemcpyDeviceToHost, st));
        PT_BVH_TRY(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, st));
        PT_BVH_TRY(hipStreamSynchronize(st));
        if (h.error & kErrDepth) {
            *error = "device BVH build: 48 levels of SAH splits did not separate the triangles; use the host builder";
            return false;
        }
        if (h.error & kErrCoincident) {
            *error = "device BVH build: more than 256 triangles share one centroid; use the host builder";
            return false;
        }
        if (n_split) ++depth;
        if (depth > params.max_depth) {
            *error = "device BVH build: tree deeper than the traversal stack; use the host builder";
            return false;
        }
        if (h.n_large_next) {
            // the next level's chunk list
            hipLaunchKernelGGL(k_chunk_counts, dim3(blocks_for(h.n_large_next, 256)), dim3(256), 0, st, nodes, slot_node,
                               h.n_large_next, chunks);
            exclusive_scan(chunks, h.n_large_next, chunk_start, sums, d_total, st);
            PT_BVH_TRY(hipMemcpyAsync(&n_chunks, d_total, 4, hipMemcpyDeviceToHost, st));
            PT_BVH_TRY(hipStreamSynchronize(st));
        }
        lvl_start = next_start;
        lvl_count = 2u * n_split;
        n_large = h.n_large_next;
        cur ^= 1;
        ++level;
    }
    const uint32_t n_tmp = lvl_start;  // every node made

    NodeRec root{};
    PT_BVH_TRY(hipMemcpyAsync(&root, nodes, sizeof(root), hipMemcpyDeviceToHost, st));
    PT_BVH_TRY(hipMemsetAsync(d_cost, 0, 8, st));
    hipLaunchKernelGGL(k_sah_cost, dim3(std::min<uint32_t>(512, blocks_for(n_tmp, 256))), dim3(256), 0, st, nodes, n_tmp, d_cost);

    void *d_nodes = nullptr, *d_tris = nullptr;
    uint32_t n_inner = 0;
    if (depth == 0) {
        n_inner = 1;
        depth = 1;
        PT_BVH_TRY(hipMalloc(&d_nodes, 64));
        hipLaunchKernelGGL(k_emit_leaf_root, dim3(1), dim3(1), 0, st, nodes, static_cast<float4 *>(d_nodes));
    } else {
        // breadth-first rank of every inner node = inner nodes before it in the node array
        hipLaunchKernelGGL(k_inner_flags, dim3(blocks_for(n_tmp, 256)), dim3(256), 0, st, nodes, n_tmp, flags);
        exclusive_scan(flags, n_tmp, rank, sums, d_total, st);
        PT_BVH_TRY(hipMemcpyAsync(&n_inner, d_total, 4, hipMemcpyDeviceToHost, st));
        for (size_t l = levels.size(); l-- > 0;)
            hipLaunchKernelGGL(k_subtree_sizes, dim3(blocks_for(levels[l].count, 256)), dim3(256), 0, st, nodes, levels[l].start,
                               levels[l].count, sizes);
        const uint32_t n_roots = 2u * top + 2u;
        PT_BVH_TRY(hipMemsetAsync(root_size, 0, static_cast<size_t>(n_roots) * 4, st));
        hipLaunchKernelGGL(k_root_sizes, dim3(blocks_for(n_tmp, 256)), dim3(256), 0, st, nodes, n_tmp, rank, sizes, top, root_size);
        hipLaunchKernelGGL(k_root_offsets, dim3(1), dim3(1), 0, st, root_size, n_roots, top);
        PT_BVH_TRY(hipMemsetAsync(index, 0, 4, st));
        for (size_t l = 0; l < levels.size(); ++l)
            hipLaunchKernelGGL(k_assign_index, dim3(blocks_for(levels[l].count, 256)), dim3(256), 0, st, nodes, levels[l].start,
                               levels[l].count, rank, sizes, root_size, top, index);
        PT_BVH_TRY(hipStreamSynchronize(st));
        PT_BVH_TRY(hipMalloc(&d_nodes, static_cast<size_t>(n_inner) * 64));
        hipLaunchKernelGGL(k_emit_nodes, dim3(blocks_for(n_tmp, 256)), dim3(256), 0, st, nodes, n_tmp, index,
                           static_cast<float4 *>(d_nodes));
    }
    if (hipMalloc(&d_tris, static_cast<size_t>(T) * 64) != hipSuccess) {
        (void)hipFree(d_nodes);
        *error = "device BVH build: out of device memory";
        return false;
    }
    hipLaunchKernelGGL(k_emit_triangles, dim3(blocks_for(T, 256)), dim3(256), 0, st, order, T, d_vertices, d_ids,
                       static_cast<float4 *>(d_tris));
    double cost = 0.0;
    hipError_t e = hipMemcpyAsync(&cost, d_cost, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(d_nodes);
        (void)hipFree(d_tris);
        *error = std::string("device BVH build: ") + hipGetErrorString(e);
        return false;
    }
    out->d_nodes = d_nodes;
    out->d_triangles = d_tris;
    out->n_nodes = n_inner;
    out->depth = depth;
    out->levels = static_cast<uint32_t>(levels.size());
    for (int a = 0; a < 3; ++a) {
        out->bounds_lo[a] = root.blo[a];
        out->bounds_hi[a] = root.bhi[a];
    }
    out->sah_cost = cost;
    return true;
}

}  // namespace pt
