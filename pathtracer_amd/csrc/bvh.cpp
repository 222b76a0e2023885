#include "bvh.h"
#include "tuning.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>

namespace pt {

namespace {

// Largest leaf.  (PT_BVH_LEAF: tuning experiments only, 1..7.)
const uint32_t kLeafMax = [] {
    const char *e = pt::tuning_env("PT_BVH_LEAF");
    const int v = e ? std::atoi(e) : 4;
    return static_cast<uint32_t>(v < 1 ? 1 : (v > 7 ? 7 : v));
}();
constexpr int kMaxBins = 64;
// Cost of one inner-node step in units of one triangle test.  (PT_BVH_CT: tuning experiments only.)
const double kTraversalCost = [] {
    const char *e = pt::tuning_env("PT_BVH_CT");
    return e ? std::atof(e) : 1.0;
}();

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float *p)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    double half_area() const
    {
        double dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float centroid[3];
    uint32_t id;
};

struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;  // inner when left >= 0
    uint32_t first = 0, count = 0;  // leaf range in the ordered primitive array
};

// A subtree handed to a worker thread: the primitive range, its level, and the
// node of the top tree that stands for it until the worker's nodes are spliced in.
struct Job {
    uint32_t begin, end, level;
    int32_t placeholder;
};

struct Builder {
    std::vector<Prim> prims;
    std::vector<TmpNode> tmp;
    bool median_only = false;
    int kBins = 16;  // centroid bins per axis

    // Builds the subtree over prims[begin, end) into `out` and returns its node.
    // With `jobs`, subtrees of at most job_size primitives are not built but
    // recorded (their node stays an empty placeholder).  Ranges of different
    // subtrees are disjoint, so workers may partition them concurrently.
    int32_t build(uint32_t begin, uint32_t end, uint32_t level, std::vector<TmpNode> &tmp, std::vector<Job> *jobs,
                  uint32_t job_size)
    {
        TmpNode node;
        node.box.reset();
        Box cbox;
        cbox.reset();
        for (uint32_t i = begin; i < end; ++i) {
            node.box.grow(prims[i].box);
            cbox.grow(prims[i].centroid);
        }
        const uint32_t n = end - begin;
        int32_t self = static_cast<int32_t>(tmp.size());
        tmp.push_back(node);
        // A leaf lists its triangles by index, whatever order the partitioning left them in, so that
        // the tree is a function of the triangle set alone.
        auto make_leaf = [&]() {
            std::sort(prims.begin() + begin, prims.begin() + end, [](const Prim &a, const Prim &b) { return a.id < b.id; });
            tmp[self].first = begin;
            tmp[self].count = n;
            return self;
        };
        if (n <= 1u || (n <= kLeafMax && (median_only || level >= 48))) return make_leaf();
        if (jobs && n <= job_size && n > 4u * kLeafMax) {
            jobs->push_back(Job{begin, end, level, self});
            return self;
        }

        uint32_t mid = begin + n / 2;
        bool split_found = false;
        // Past level 48 fall back to balanced splits so that recursion stays shallow
        // on adversarial input (the caller then rebuilds with median splits only).
        if (!median_only && level < 48) {
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                float extent = cbox.hi[axis] - cbox.lo[axis];
                if (!(extent > 0.f)) continue;
                Box bin_box[kMaxBins];
                uint32_t bin_n[kMaxBins];
                for (int b = 0; b < kBins; ++b) {
                    bin_box[b].reset();
                    bin_n[b] = 0;
                }
                const float k = kBins / extent;
                for (uint32_t i = begin; i < end; ++i) {
                    int b = static_cast<int>((prims[i].centroid[axis] - cbox.lo[axis]) * k);
                    b = std::max(0, std::min(kBins - 1, b));
                    bin_box[b].grow(prims[i].box);
                    bin_n[b]++;
                }
                double right_area[kMaxBins];
                uint32_t right_n[kMaxBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_n[b];
                    right_area[b] = acc.half_area();
                    right_n[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_n[b];
                    if (cnt == 0 || right_n[b + 1] == 0) continue;
                    double cost = acc.half_area() * cnt + right_area[b + 1] * right_n[b + 1];
                    if (cost < best) {
                        best = cost;
                        best_axis = axis;
                        best_bin = b;
                    }
                }
            }
            // A small group stays a leaf unless splitting it pays for the extra box test
            // (surface-area heuristic: kTraversalCost box tests against saved triangle tests).
            if (n <= kLeafMax && (best_axis < 0 || double(n) * node.box.half_area() <=
                                                       kTraversalCost * node.box.half_area() + best))
                return make_leaf();
            if (best_axis >= 0) {
                const float lo = cbox.lo[best_axis];
                const float k = kBins / (cbox.hi[best_axis] - lo);
                Prim *first = prims.data() + begin, *last = prims.data() + end;
                Prim *m = std::partition(first, last, [&](const Prim &p) {
                    int b = static_cast<int>((p.centroid[best_axis] - lo) * k);
                    b = std::max(0, std::min(kBins - 1, b));
                    return b <= best_bin;
                });
                uint32_t cut = begin + static_cast<uint32_t>(m - first);
                if (cut > begin && cut < end) {
                    mid = cut;
                    split_found = true;
                }
            }
        }
        if (!split_found) {
            // Object-median split along the widest centroid axis (or by triangle
            // index when all centroids coincide).
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
            mid = begin + n / 2;
            if (cbox.hi[axis] - cbox.lo[axis] > 0.f)
                std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                                 [axis](const Prim &a, const Prim &b) {
                                     return a.centroid[axis] < b.centroid[axis] ||
                                            (a.centroid[axis] == b.centroid[axis] && a.id < b.id);
                                 });
            else
                std::sort(prims.begin() + begin, prims.begin() + end,
                          [](const Prim &a, const Prim &b) { return a.id < b.id; });
        }
        int32_t l = build(begin, mid, level + 1, tmp, jobs, job_size);
        int32_t r = build(mid, end, level + 1, tmp, jobs, job_size);
        tmp[self].left = l;
        tmp[self].right = r;
        return self;
    }

    // The whole tree into this->tmp; returns the root.  The top of the tree is
    // built on the calling thread, the subtrees below it on worker threads, and
    // their nodes are appended in subtree order, so the result does not depend
    // on the number of threads or their timing.
    int32_t build_all()
    {
        const uint32_t n = static_cast<uint32_t>(prims.size());
        unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (const char *e = pt::tuning_env("PT_BVH_THREADS")) hw = static_cast<unsigned>(std::max(1, std::atoi(e)));
        if (n < (1u << 16) || hw < 2) return build(0, n, 0, tmp, nullptr, 0);
        std::vector<Job> jobs;
        const int32_t root = build(0, n, 0, tmp, &jobs, std::max<uint32_t>(1024, n / (8 * hw)));
        std::vector<std::vector<TmpNode>> parts(jobs.size());
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t j = next.fetch_add(1); j < jobs.size(); j = next.fetch_add(1)) {
                parts[j].reserve(2 * static_cast<size_t>(jobs[j].end - jobs[j].begin) / std::max(1u, kLeafMax / 2) + 16);
                build(jobs[j].begin, jobs[j].end, jobs[j].level, parts[j], nullptr, 0);  // its root is parts[j][0]
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < hw; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
        for (size_t j = 0; j < jobs.size(); ++j) {
            // node i >= 1 of the part becomes node offset + i - 1; node 0 replaces the placeholder
            const int32_t offset = static_cast<int32_t>(tmp.size());
            auto remap = [offset](TmpNode nd) {
                if (nd.left >= 0) {
                    nd.left += offset - 1;
                    nd.right += offset - 1;
                }
                return nd;
            };
            tmp[jobs[j].placeholder] = remap(parts[j][0]);
            for (size_t i = 1; i < parts[j].size(); ++i) tmp.push_back(remap(parts[j][i]));
            std::vector<TmpNode>().swap(parts[j]);
        }
        return root;
    }
};

int32_t leaf_code(uint32_t first, uint32_t count) { return ~static_cast<int32_t>((first << 3) | count); }

void put_boxes(BvhNode *n, const Box &b0, const Box &b1)
{
    n->q[0] = b0.lo[0]; n->q[1] = b0.lo[1]; n->q[2] = b0.lo[2];
    n->q[3] = b0.hi[0]; n->q[4] = b0.hi[1]; n->q[5] = b0.hi[2];
    n->q[6] = b1.lo[0]; n->q[7] = b1.lo[1]; n->q[8] = b1.lo[2];
    n->q[9] = b1.hi[0]; n->q[10] = b1.hi[1]; n->q[11] = b1.hi[2];
}

// Flattens the temporary tree in pre-order; returns the number of inner nodes
// on the longest path below (and including) tmp node t.
uint32_t flatten(const std::vector<TmpNode> &tmp, int32_t t, std::vector<BvhNode> *nodes, int32_t *code_out)
{
    const TmpNode &n = tmp[t];
    if (n.left < 0) {
        *code_out = leaf_code(n.first, n.count);
        return 0;
    }
    int32_t self = static_cast<int32_t>(nodes->size());
    nodes->push_back(BvhNode());
    *code_out = self;
    int32_t c0, c1;
    uint32_t d0 = flatten(tmp, n.left, nodes, &c0);
    uint32_t d1 = flatten(tmp, n.right, nodes, &c1);
    BvhNode &out = (*nodes)[self];
    std::memset(&out, 0, sizeof(out));
    put_boxes(&out, tmp[n.left].box, tmp[n.right].box);
    out.child[0] = c0;
    out.child[1] = c1;
    return 1 + std::max(d0, d1);
}

// Renumbers the inner nodes: the top of the tree breadth-first (so that every
// prefix of the node array is closed under "parent of", which lets the
// traversal kernel keep a prefix in LDS), the subtrees below it in pre-order
// (a subtree's nodes stay together in memory).
void renumber_top_breadth_first(std::vector<BvhNode> *nodes, size_t top)
{
    const size_t n = nodes->size();
    if (n <= 2) return;
    std::vector<int32_t> order;  // new index -> old index
    order.reserve(n);
    std::vector<int32_t> queue{0};
    size_t head = 0;
    while (head < queue.size() && order.size() < top) {
        const int32_t old = queue[head++];
        order.push_back(old);
        for (int c = 0; c < 2; ++c)
            if ((*nodes)[old].child[c] >= 0) queue.push_back((*nodes)[old].child[c]);
    }
    std::vector<int32_t> todo;
    for (; head < queue.size(); ++head) {
        todo.push_back(queue[head]);
        while (!todo.empty()) {
            const int32_t old = todo.back();
            todo.pop_back();
            order.push_back(old);
            for (int c = 1; c >= 0; --c)
                if ((*nodes)[old].child[c] >= 0) todo.push_back((*nodes)[old].child[c]);
        }
    }
    std::vector<int32_t> new_of_old(n);
    for (size_t i = 0; i < n; ++i) new_of_old[order[i]] = static_cast<int32_t>(i);
    std::vector<BvhNode> out(n);
    for (size_t i = 0; i < n; ++i) {
        out[i] = (*nodes)[order[i]];
        for (int c = 0; c < 2; ++c)
            if (out[i].child[c] >= 0) out[i].child[c] = new_of_old[out[i].child[c]];
    }
    nodes->swap(out);
}

}  // namespace

uint32_t bvh_leaf_max() { return kLeafMax; }
double bvh_traversal_cost() { return kTraversalCost; }

void build_bvh(const float *vertices, const uint32_t *ids, uint32_t n_triangles, uint32_t max_depth, Bvh *out, int bins)
{
    *out = Bvh();
    Builder b;
    b.kBins = bins < 2 ? 2 : (bins > kMaxBins ? kMaxBins : bins);
    b.prims.resize(n_triangles);
    Box scene;
    scene.reset();
    for (uint32_t t = 0; t < n_triangles; ++t) {
        Prim &p = b.prims[t];
        p.id = t;
        p.box.reset();
        for (int k = 0; k < 3; ++k) p.box.grow(vertices + 3 * static_cast<size_t>(ids[3 * t + k]));
        for (int a = 0; a < 3; ++a) p.centroid[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
        scene.grow(p.box);
    }
    if (n_triangles == 0)
        for (int a = 0; a < 3; ++a) scene.lo[a] = scene.hi[a] = 0.f;

    for (int attempt = 0; attempt < 2; ++attempt) {
        b.tmp.clear();
        b.tmp.reserve(static_cast<size_t>(n_triangles) + 16);
        b.median_only = (attempt == 1);
        out->nodes.clear();
        int32_t root = b.build_all();
        int32_t code;
        if (b.tmp[root].left < 0) {
            // A scene of <= 4 triangles: wrap the single leaf in a root whose
            // second child is an empty leaf.
            BvhNode n;
            std::memset(&n, 0, sizeof(n));
            put_boxes(&n, b.tmp[root].box, b.tmp[root].box);
            n.child[0] = leaf_code(b.tmp[root].first, b.tmp[root].count);
            n.child[1] = leaf_code(0, 0);
            out->nodes.push_back(n);
            out->depth = 1;
        } else {
            out->depth = flatten(b.tmp, root, &out->nodes, &code);
        }
        if (out->depth <= max_depth) break;
    }
    renumber_top_breadth_first(&out->nodes, 4096);

    double root_area = scene.half_area();
    double cost = 0.0;
    if (root_area > 0.0)
        for (const TmpNode &n : b.tmp) cost += (n.left >= 0 ? kTraversalCost : double(n.count)) * n.box.half_area() / root_area;
    out->sah_cost = cost;
    for (int a = 0; a < 3; ++a) {
        out->bounds_lo[a] = scene.lo[a];
        out->bounds_hi[a] = scene.hi[a];
    }

    out->triangles.resize(n_triangles);
    for (uint32_t i = 0; i < n_triangles; ++i) {
        const uint32_t t = b.prims[i].id;
        const float *v0 = vertices + 3 * static_cast<size_t>(ids[3 * t + 0]);
        const float *v1 = vertices + 3 * static_cast<size_t>(ids[3 * t + 1]);
        const float *v2 = vertices + 3 * static_cast<size_t>(ids[3 * t + 2]);
        BvhTriangle &r = out->triangles[i];
        std::memset(&r, 0, sizeof(r));
        for (int a = 0; a < 3; ++a) {
            r.v0[a] = v0[a];
            r.v1[a] = v1[a];
            r.v2[a] = v2[a];
        }
        // pathtracer.frag.glsl:399-404, one rounding per operation.
        const float e0x = v1[0] - v0[0], e0y = v1[1] - v0[1], e0z = v1[2] - v0[2];
        const float e1x = v2[0] - v0[0], e1y = v2[1] - v0[1], e1z = v2[2] - v0[2];
        const float cx = e0y * e1z - e0z * e1y;
        const float cy = e0z * e1x - e0x * e1z;
        const float cz = e0x * e1y - e0y * e1x;
        const float inv_len = 1.0f / std::sqrt(cx * cx + (cy * cy + cz * cz));
        r.nx = cx * inv_len;
        r.ny = cy * inv_len;
        r.nz = cz * inv_len;
        r.d0 = r.nx * v0[0] + (r.ny * v0[1] + r.nz * v0[2]);
        r.original = t;
    }
}

void build_bvh_best_of(const float *vertices, const uint32_t *ids, uint32_t n_triangles, uint32_t max_depth, Bvh *out)
{
    static const int many[] = {16, 8, 12, 24, 32, 48, 64}, few[] = {16, 8, 32};
    const int *bins = n_triangles <= 8192u ? many : few;
    const int n_bins = n_triangles <= 8192u ? 7 : 3;
    build_bvh(vertices, ids, n_triangles, max_depth, out, bins[0]);
    for (int k = 1; k < n_bins; ++k) {
        Bvh other;
        build_bvh(vertices, ids, n_triangles, max_depth, &other, bins[k]);
        if (other.depth <= max_depth && other.sah_cost < out->sah_cost) *out = std::move(other);
    }
}

void order_leaves_by_area(Bvh *bvh)
{
    struct Leaf {
        double area;
        uint32_t first, count, node, child;
    };
    std::vector<Leaf> leaves;
    for (uint32_t i = 0; i < bvh->nodes.size(); ++i)
        for (uint32_t c = 0; c < 2; ++c) {
            const int32_t code = bvh->nodes[i].child[c];
            if (code >= 0) continue;
            const uint32_t bits = static_cast<uint32_t>(~code);
            if ((bits & 7u) == 0u) continue;  // the empty second child of a one-leaf tree
            const float *q = bvh->nodes[i].q + 6 * c;  // lo xyz, hi xyz
            const double dx = double(q[3]) - q[0], dy = double(q[4]) - q[1], dz = double(q[5]) - q[2];
            leaves.push_back(Leaf{dx * dy + dy * dz + dz * dx, bits >> 3, bits & 7u, i, c});
        }
    std::stable_sort(leaves.begin(), leaves.end(), [](const Leaf &a, const Leaf &b) {
        return a.area != b.area ? a.area > b.area : a.first < b.first;
    });
    std::vector<BvhTriangle> ordered;
    ordered.reserve(bvh->triangles.size());
    for (const Leaf &l : leaves) {
        const uint32_t first = static_cast<uint32_t>(ordered.size());
        for (uint32_t k = 0; k < l.count; ++k) ordered.push_back(bvh->triangles[l.first + k]);
        bvh->nodes[l.node].child[l.child] = leaf_code(first, l.count);
    }
    if (ordered.size() == bvh->triangles.size()) bvh->triangles.swap(ordered);
}

}  // namespace pt
