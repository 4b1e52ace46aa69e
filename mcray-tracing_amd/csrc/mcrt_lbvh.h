// mcrt_lbvh.h -- device-side BVH construction (mcrt_lbvh.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcrt {

struct LbvhResult {
    float4 *d_nodes = nullptr;      // [n_nodes4] 128-byte BVH4 nodes (mcrt_bvh4_node), hipMalloc'ed, owned by the caller
    float4 *d_tris = nullptr;       // [n_tri][3] leaf-order triangles: v0|id, v1|mesh, v2|0
    uint32_t *d_tri_slot = nullptr; // [n_tri] triangle id -> leaf-order position
    uint32_t n_nodes4 = 0, max_stack = 0, max_depth = 0;
    float pad_abs = 0.0f;
    float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };   // bounds of the whole tree
};

// tri_dev: [n_tri][9] floats on the device (world space); mesh_dev: [n_tri] mesh index per triangle (or null = 0).
// Runs on stream st and synchronises it.  Returns an mcrt_status.
int lbvh_build(const float *tri_dev, const uint32_t *mesh_dev, uint32_t n_tri, hipStream_t st, LbvhResult *out);

// New vertex positions for an existing tree (either builder): rewrites the walk's triangle records d_recs (MCRT_TRI_PIECES x 16 bytes each) in place and refits
// every box of d_nodes bottom-up; returns the new absolute pad and the tree's bounds.  Runs on st and synchronises it.
int bvh_refit(const float *tri_dev, uint32_t n_tri, float4 *d_nodes, uint32_t n_nodes4, float4 *d_recs, hipStream_t st, float *pad_abs, float lo[3], float hi[3]);

}  // namespace mcrt
