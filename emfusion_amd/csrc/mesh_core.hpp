// mesh_core.hpp -- what the two consumers of the marching-cubes geometry share: the cube / corner /
// edge conventions of the reference (TSDF.cu:872-1096) and its vertexInterp.  lifecycle.hip streams
// the edge vertices into order statistics (updateObj), meshing.hip emits the mesh itself.
#pragma once

#include "common.hpp"

namespace emf_hip {

struct MeshSource {
    const float* tsdf;
    const float* weights;
    const uint8_t* fg;  // fgVolMask or nullptr
    I3 n;
    float voxelSize;
};

__device__ __forceinline__ V3 vertex_interp(const V3& p1, const V3& p2, float v1, float v2) {
    // TSDF.cu:909-920; the comparisons are against the double literal 0.00001
    if (static_cast<double>(fabsf(v1)) < 0.00001) return p1;
    if (static_cast<double>(fabsf(v2)) < 0.00001) return p2;
    if (static_cast<double>(fabsf(v1 - v2)) < 0.00001) return p1;
    const float mu = -v1 / (v2 - v1);
    const V3 d = v3(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);  // p1 + mu * (p2 - p1)
    return p1 + d * mu;
}

// corner i of cube (x, y, z) in the reference's numbering (TSDF.cu:896-903): x + (i ^ (i >> 1)) & 1,
// z + (i >> 1) & 1, y + (i >> 2) & 1
__device__ __forceinline__ void cube_corner(int i, int& dx, int& dy, int& dz) {
    dx = ((i & 1) ^ ((i >> 1) & 1));
    dz = (i >> 1) & 1;
    dy = (i >> 2) & 1;
}

// the corners joined by edge e (bit e of the reference's edgeTable[cls] is set iff their signs differ)
__device__ __forceinline__ unsigned active_edges(unsigned cls) {
    constexpr int e0[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
    constexpr int e1[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) m |= (((cls >> e0[e]) ^ (cls >> e1[e])) & 1u) << e;
    return m;
}

}  // namespace emf_hip
