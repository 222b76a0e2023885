// Host-side binned-SAH BVH over the mesh triangles, flattened into the
// node / triangle records the extend kernel reads.
#pragma once
#include <cstdint>
#include <vector>

namespace pt {

// One inner node = 64 bytes = four 16-byte lanes, holding BOTH children's
// boxes so that one fetch decides the descent:
//   q0 = (lo0.x, lo0.y, lo0.z, hi0.x)
//   q1 = (hi0.y, hi0.z, lo1.x, lo1.y)
//   q2 = (lo1.z, hi1.x, hi1.y, hi1.z)
//   q3 = (child0, child1, 0, 0) as int32 bits
// child >= 0: index of an inner node.  child < 0: a leaf, ~child =
// (first_triangle << 3) | count, count in 0..7, first_triangle indexing the
// leaf-ordered triangle records.
struct BvhNode {
    float q[12];
    int32_t child[2];
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is one 64-byte record");

// One leaf-ordered triangle = 64 bytes = four 16-byte lanes:
//   t0 = (v0.x, v0.y, v0.z, d0)      d0 = dot(n, v0)
//   t1 = (v1.x, v1.y, v1.z, n.x)     n  = normalize(cross(v1-v0, v2-v0))
//   t2 = (v2.x, v2.y, v2.z, n.y)
//   t3 = (n.z, original triangle index bits, 0, 0)
// The renderer sets kHitClassBit in the uploaded index of triangles whose
// material is not purely diffuse; the kernels mask it off wherever the index
// is compared or reported.
// n and d0 are evaluated exactly as the reference shader evaluates them per
// ray (pathtracer.frag.glsl:399-404): same operations, same order, no
// contraction, so hoisting them out of the ray loop changes no bit.
constexpr uint32_t kHitClassBit = 1u << 30;

struct BvhTriangle {
    float v0[3], d0;
    float v1[3], nx;
    float v2[3], ny;
    float nz;
    uint32_t original;
    uint32_t pad[2];
};
static_assert(sizeof(BvhTriangle) == 64, "BvhTriangle is one 64-byte record");

struct Bvh {
    std::vector<BvhNode> nodes;          // nodes[0] is the root (always an inner node)
    std::vector<BvhTriangle> triangles;  // leaf order
    uint32_t depth = 0;                  // inner nodes on the longest root-to-leaf path
    float bounds_lo[3] = {0, 0, 0};
    float bounds_hi[3] = {0, 0, 0};
    double sah_cost = 0.0;
};

// vertices: xyz triples; tri_vertex_ids: 3 vertex indices per triangle.
// max_depth bounds the traversal stack the kernel needs: a tree that comes out
// deeper is rebuilt with object-median splits.
// bins: centroid bins per axis of the SAH (2..64); the device builder (bvh_device.h) makes
// the tree of bins = 16.
void build_bvh(const float *vertices, const uint32_t *tri_vertex_ids, uint32_t n_triangles,
               uint32_t max_depth, Bvh *out, int bins = 16);

// Greedy top-down SAH is sensitive to where the bin boundaries fall (on the bundled scenes the
// total cost moves by up to 12 % between 8 and 64 bins, in no particular direction), and a
// small tree takes well under a millisecond to build: this builds the tree with several bin
// counts (seven up to 8192 triangles, three above) and keeps the one of least SAH cost.
void build_bvh_best_of(const float *vertices, const uint32_t *tri_vertex_ids, uint32_t n_triangles,
                       uint32_t max_depth, Bvh *out);

// Renumbers the leaves so that the leaf-ordered triangle array starts with the triangles of the leaves with the
// largest boxes -- the leaves a random ray is most likely to reach (the surface-area heuristic's own assumption).
// The tree is the same tree; a traversal kernel that can keep only a prefix of the triangle array in fast memory
// (LDS) then keeps the part that is fetched most.
void order_leaves_by_area(Bvh *bvh);

// The two constants of the builder's surface-area heuristic: the largest leaf
// and the cost of one inner-node step in units of one triangle test.
uint32_t bvh_leaf_max();
double bvh_traversal_cost();

}  // namespace pt
