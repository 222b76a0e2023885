// Device-side binned-SAH BVH build: the same tree as build_bvh() (bvh.h) makes
// on the host, built level by level on the GPU from triangle data that is
// already resident there.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace pt {

struct DeviceBvhParams {
    uint32_t leaf_max = 4;         // largest leaf (bvh_leaf_max())
    double traversal_cost = 1.0;   // bvh_traversal_cost()
    uint32_t max_depth = 64;       // deepest tree the traversal stack holds
    uint32_t top_breadth_first = 4096;  // inner nodes numbered breadth-first at the top of the array
};

struct DeviceBvh {
    void *d_nodes = nullptr;      // BvhNode[n_nodes]      (hipMalloc'ed; the caller frees)
    void *d_triangles = nullptr;  // BvhTriangle[n_triangles], leaf order
    uint32_t n_nodes = 0;
    uint32_t depth = 0;
    uint32_t levels = 0;          // levels of the build (root = level 0), leaves included
    float bounds_lo[3] = {0, 0, 0};
    float bounds_hi[3] = {0, 0, 0};
    double sah_cost = 0.0;
};

// d_vertices: xyz triples; d_tri_vertex_ids: 3 vertex indices per triangle; both
// in device memory, n_triangles >= 1.  Makes every split decision with the
// arithmetic of the host builder (16 centroid bins per axis, costs in double),
// so the result equals build_bvh()'s node for node and record for record.
// Returns false with *error set when the input needs what only the host builder
// has: its object-median fall-back for a tree deeper than max_depth or 48 SAH
// levels, or more than 256 triangles sharing one centroid.
bool build_bvh_device(const float *d_vertices, const uint32_t *d_tri_vertex_ids, uint32_t n_triangles,
                      const DeviceBvhParams &params, hipStream_t stream, DeviceBvh *out, std::string *error);

// The binary tree collapsed two levels at a time: one 128-byte record (a cache line) per inner node of even
// level holding the boxes and codes of the up to four nodes two levels below it.  For trees that are read from
// memory: a ray then waits for half as many dependent node fetches.  d_nodes: BvhNode[n_nodes] in device memory
// (either builder's), depth as DeviceBvh / Bvh report it.  The result is hipMalloc'ed; the caller frees it.
struct WideBvh {
    void *d_nodes = nullptr;  // n_nodes records of 8 x float4: lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] codes[4] -
    uint32_t n_nodes = 0;
    uint32_t depth = 0;       // four-wide nodes on the longest root-to-leaf path
};
bool build_wide_nodes(const void *d_nodes, uint32_t n_nodes, uint32_t depth, hipStream_t stream, WideBvh *out, std::string *error);

// Sets kHitClassBit (bvh.h) in the index of every leaf-ordered triangle record whose
// material is not purely diffuse.  d_materials: 8 floats per material (albedo rgb,
// roughness, metallic, ...); d_material_map: material index per original triangle.
void mark_triangle_classes(void *d_triangles, uint32_t n_triangles, const uint32_t *d_material_map, const float *d_materials,
                           uint32_t n_materials, hipStream_t stream);

// The per-triangle shading records (device_types.h ShadeTriangle, 80 bytes each) from the
// scene arrays as the caller's Scene lays them out: d_indices 4 floats per triangle,
// d_vertex_data 6 floats per corner.  d_out holds n_triangles records.
void build_shade_records(const float *d_vertices, const uint32_t *d_tri_vertex_ids, const float *d_indices,
                         const float *d_vertex_data, const uint32_t *d_material_map, uint32_t n_triangles, void *d_out,
                         hipStream_t stream);

}  // namespace pt
