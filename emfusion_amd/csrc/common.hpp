// common.hpp -- shared device math + host-side argument checking for the emf_hip_* entry points.
//
// The vector helpers fix the floating-point operation order the reference's kernels use
// (reference include/EMFusion/core/cuda/common.cuh:92-202: row dots summed left to right,
// component-wise true division, int3/int integer division).  Everything is compiled with
// -ffp-contract=off so hipcc never fuses a*b+c; see DESIGN.md "Numerics".
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "emf_hip.h"

namespace emf_hip {

struct V3 {
    float x, y, z;
};
struct M33 {
    V3 r0, r1, r2;
};
struct I3 {
    int x, y, z;
};

__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__host__ __device__ __forceinline__ float dot(const V3& a, const V3& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__host__ __device__ __forceinline__ V3 mul(const M33& m, const V3& v) {
    return V3{dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)};
}
__host__ __device__ __forceinline__ V3 operator+(const V3& a, const V3& b) {
    return V3{a.x + b.x, a.y + b.y, a.z + b.z};
}
__host__ __device__ __forceinline__ V3 operator*(const V3& a, float f) {
    return V3{a.x * f, a.y * f, a.z * f};
}
__host__ __device__ __forceinline__ V3 operator/(const V3& a, float f) {
    return V3{a.x / f, a.y / f, a.z / f};
}
__host__ __device__ __forceinline__ float norm(const V3& v) {
    return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
}
__host__ __device__ __forceinline__ M33 transpose(const M33& m) {
    return M33{{m.r0.x, m.r1.x, m.r2.x}, {m.r0.y, m.r1.y, m.r2.y}, {m.r0.z, m.r1.z, m.r2.z}};
}
inline M33 m33_from(const float* a) {
    return M33{{a[0], a[1], a[2]}, {a[3], a[4], a[5]}, {a[6], a[7], a[8]}};
}
inline V3 v3_from(const float* a) { return V3{a[0], a[1], a[2]}; }
inline I3 i3_from(const int32_t* a) { return I3{a[0], a[1], a[2]}; }

// (N - 1) / 2.f per axis: the voxel-space centre offset (reference TSDF.cu:345-348, 507-508)
__host__ __device__ __forceinline__ V3 half_extent(const I3& n) {
    return V3{static_cast<float>(n.x - 1) / 2.f, static_cast<float>(n.y - 1) / 2.f,
              static_cast<float>(n.z - 1) / 2.f};
}

// p / voxelSize + (N - 1) / 2.f
__device__ __forceinline__ V3 to_voxel(const V3& p, float voxelSize, const V3& half) {
    return p / voxelSize + half;
}

// v in [0, N - pad) on every axis  (pad = 1: getVolumeVals / coarse search; pad = 2: march)
__device__ __forceinline__ bool outside(const V3& v, float pad, const I3& n) {
    return v.x < 0 || v.x + pad >= static_cast<float>(n.x) || v.y < 0 ||
           v.y + pad >= static_cast<float>(n.y) || v.z < 0 || v.z + pad >= static_cast<float>(n.z);
}

// Trilinear blend in the reference's order: x pairs, then y, then z, each (1 - f) * a + f * b
// (reference TSDF.cuh:84-96).  c000..c111 indexed z,y,x.
__device__ __forceinline__ float blend8(float c000, float c001, float c010, float c011,
                                        float c100, float c101, float c110, float c111, float fx,
                                        float fy, float fz) {
    const float a0 = (1 - fx) * c000 + fx * c001;
    const float a1 = (1 - fx) * c010 + fx * c011;
    const float a2 = (1 - fx) * c100 + fx * c101;
    const float a3 = (1 - fx) * c110 + fx * c111;
    const float b0 = (1 - fy) * a0 + fy * a1;
    const float b1 = (1 - fy) * a2 + fy * a3;
    return (1 - fz) * b0 + fz * b1;
}

// Corner addressing for a sample point: base index of (lz, ly, lx) plus the three strides.
struct Cell {
    size_t base;  // ((lz * Ny) + ly) * Nx + lx
    float fx, fy, fz;
};
__device__ __forceinline__ Cell cell_of(const V3& idx, const I3& n) {
    const int lx = static_cast<int>(idx.x), ly = static_cast<int>(idx.y),
              lz = static_cast<int>(idx.z);
    Cell c;
    c.base = (static_cast<size_t>(lz) * n.y + ly) * static_cast<size_t>(n.x) + lx;
    c.fx = idx.x - static_cast<float>(lx);
    c.fy = idx.y - static_cast<float>(ly);
    c.fz = idx.z - static_cast<float>(lz);
    return c;
}

// Single-channel trilinear lookup, 8 scalar gathers.
__device__ __forceinline__ float trilinear1(const float* __restrict__ vol, const Cell& c,
                                            const I3& n) {
    const size_t sy = static_cast<size_t>(n.x), sz = static_cast<size_t>(n.x) * n.y;
    const float* p = vol + c.base;
    return blend8(p[0], p[1], p[sy], p[sy + 1], p[sz], p[sz + 1], p[sz + sy], p[sz + sy + 1],
                  c.fx, c.fy, c.fz);
}

// ---- image views ------------------------------------------------------------------------------

template <typename T>
struct Img {
    T* data;
    size_t pitch;  // bytes
    __device__ __forceinline__ T* row(int y) const {
        return (T*)((char*)data + static_cast<size_t>(y) * pitch);
    }
};

template <typename T>
inline Img<T> img(const emf_image_t* im) {
    return Img<T>{static_cast<T*>(im->data), im->pitch};
}

// ---- host-side error plumbing -----------------------------------------------------------------

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_image(const emf_image_t* im, size_t elem_bytes, const char* name);
int check_same_size(const emf_image_t* a, const emf_image_t* b, const char* an, const char* bn);
int check_res(const int32_t res[3]);
int launch_status(const char* what);

#define EMF_TRY(expr)                 \
    do {                              \
        const int emf_rc_ = (expr);   \
        if (emf_rc_ != EMF_OK) return emf_rc_; \
    } while (0)

#define EMF_REQUIRE_PTR(p)                                         \
    do {                                                           \
        if ((p) == nullptr) return fail(EMF_E_NULL, "%s: %s is NULL", __func__, #p); \
    } while (0)

inline hipStream_t as_stream(emf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned ceil_div(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

}  // namespace emf_hip
