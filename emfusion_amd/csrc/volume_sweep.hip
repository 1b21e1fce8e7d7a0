// volume_sweep.hip -- dense voxel sweeps: TSDF integration, gradient volume, foreground/background
// counts, foreground probability.  All are HBM-streaming kernels over continuous (Nz*Ny) x Nx
// volumes; each lane owns VEC consecutive x-voxels so the read-modify-write streams are 16-byte
// vector accesses along the fastest axis (1 KiB per wave instruction).
#include "device_core.hpp"

namespace emf_hip {
namespace {

constexpr int kBlock = 256;

// ---- a7: TSDF integration (reference kernel_updateTSDF, TSDF.cu:327-401) -------------------------
//
// Two kernels share the per-voxel code of device_core.hpp:
//   k_update_tsdf_tiled  Nx % 4 == 0: one workgroup per 32 x 8 x 8 voxel tile (= 32 bricks of 4^3).
//       Each lane owns 4 consecutive x voxels (16-byte accesses, a wave covers eight 128-byte row
//       segments); tiles that provably project outside the image are culled with 8 corner
//       projections instead of 2048 voxel projections; the brick uniformity flags of the tile are
//       rebuilt from the final voxel values with an LDS AND-reduction.
//   k_update_tsdf_linear any size: one lane per voxel, flags of touched bricks degrade to MIXED.

struct TileGrid {
    int ntx, nty, ntz;
};

__attribute__((amdgpu_waves_per_eu(EMF_INT_WPE, EMF_INT_WPE)))
__global__ __launch_bounds__(kBlock) void k_update_tsdf_tiled(const IntegrateGeom a, float* tsdf,
                                                              float* weights, uint8_t* bricks,
                                                              const TileGrid g) {
    __shared__ unsigned lds[32];
    const int b = blockIdx.x;
    const int tx = b % g.ntx, ty = (b / g.ntx) % g.nty, tz = b / (g.ntx * g.nty);
    integrate_tile(a, tsdf, weights, bricks, tx * kTileX, ty * kTileY, tz * kTileZ, lds);
}

// 1 / lambda per pixel: the part of the integration that depends on the pixel only.  ~55 of the
// ~200 instructions a fusing voxel costs (two divisions by fx / fy, a norm, a reciprocal); as a
// table it is one cached load.  Same expression, same flags -> same bits as the inline form.
__global__ __launch_bounds__(kBlock) void k_inv_lambda(const M33 K, Img<float> out, int w, int h) {
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
    if (x < w && y < h) out.row(y)[x] = inv_lambda_at(K, x, y);
}

__global__ __launch_bounds__(kBlock) void k_update_tsdf_linear(const IntegrateGeom a, float* tsdf,
                                                               float* weights, uint8_t* bricks) {
    const size_t total = static_cast<size_t>(a.n.x) * a.n.y * a.n.z;
    const size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / a.n.x;
    const int x = static_cast<int>(i - row * a.n.x);
    const int y = static_cast<int>(row % a.n.y), z = static_cast<int>(row / a.n.y);
    float samp = 0.f, aw = 0.f;
    const int kind = classify_voxel(a, half_extent(a.n), x, y, z, samp, aw);
    if (kind == kSkip) return;
    float wv = weights[i];
    float tv = tsdf[i];  // also for the "if unseen" branches: they store only a value that differs
    const int changed = apply_voxel(kind, samp, aw, a.maxWeight, tv, wv);
    if (changed & 1) tsdf[i] = tv;
    if (changed & 2) weights[i] = wv;
    if (bricks && (changed & 1))  // conservative: a written brick is treated as mixed
        bricks[(static_cast<size_t>(z >> kBrickShift) * bricks_along(a.n.y) + (y >> kBrickShift)) *
                   bricks_along(a.n.x) + (x >> kBrickShift)] = kBrickMixed;
}

// dilated[b] = raw[b] if the brick and all of its in-volume 26 neighbours share that non-zero
// class, else MIXED (see sample_tsdf in device_core.hpp)
__global__ __launch_bounds__(kBlock) void k_dilate_flags(const uint8_t* __restrict__ raw,
                                                         uint8_t* __restrict__ dil, int nbx,
                                                         int nby, int nbz) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nbx * nby * nbz) return;
    const int bx = i % nbx, by = (i / nbx) % nby, bz = i / (nbx * nby);
    dil[i] = dilated_flag(raw, nbx, nby, nbz, bx, by, bz);
}

__global__ __launch_bounds__(kBlock) void k_fill_bytes(uint8_t* p, size_t n, uint8_t v) {
    const size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- a8: gradient volume (reference TSDF.cpp:120-123 + kernel_computeTSDFGrads TSDF.cu:429-448) --

__global__ __launch_bounds__(kBlock) void k_tsdf_grads(const float* __restrict__ tsdf,
                                                       float* __restrict__ grads, const I3 n) {
    const size_t total = static_cast<size_t>(n.x) * n.y * n.z;
    const size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= total) return;
    const size_t nx = n.x, sz = static_cast<size_t>(n.x) * n.y;
    const size_t row = i / nx;
    const int x = static_cast<int>(i - row * nx);
    const int y = static_cast<int>(row % n.y), z = static_cast<int>(row / n.y);
    float gx = 0.f, gy = 0.f, gz = 0.f;  // last index planes stay zero (fused setTo(0))
    if (x < n.x - 1 && y < n.y - 1 && z < n.z - 1) {
        const float t = tsdf[i];
        gx = tsdf[i + 1] - t;
        gy = tsdf[i + nx] - t;
        gz = tsdf[i + sz] - t;
    }
    float* g = grads + 3 * i;
    g[0] = gx;
    g[1] = gy;
    g[2] = gz;
}

// ---- a13: fg/bg counts (reference kernel_updateFgBgProbs, ObjTSDF.cu:29-80) ----------------------

struct FgBgArgs {
    Img<const uint8_t> mask, occluded;
    int w, h;
    const float* tsdf;
    const float* weights;
    float2* fgbg;
    M33 R;
    V3 t;
    M33 K;
    I3 n;
    float voxelSize;
};

__global__ __launch_bounds__(kBlock) void k_update_fgbg(const FgBgArgs a) {
    const size_t total = static_cast<size_t>(a.n.x) * a.n.y * a.n.z;
    const size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= total) return;
    const float tv = a.tsdf[i];
    const float wv = a.weights[i];
    if (fabsf(tv) >= 1.f || wv == 0.f) return;  // only seen voxels inside the truncation band
    const size_t row = i / a.n.x;
    const int x = static_cast<int>(i - row * a.n.x);
    const int y = static_cast<int>(row % a.n.y), z = static_cast<int>(row / a.n.y);
    const V3 half = half_extent(a.n);
    const V3 pobj = v3((static_cast<float>(x) - half.x) * a.voxelSize,
                       (static_cast<float>(y) - half.y) * a.voxelSize,
                       (static_cast<float>(z) - half.z) * a.voxelSize);
    const V3 pcam = mul(a.R, pobj) + a.t;
    if (pcam.z <= 0.f) return;
    const V3 proj = mul(a.K, pcam);
    const int px = __float2int_rn(proj.x / proj.z);
    const int py = __float2int_rn(proj.y / proj.z);
    if (px < 0 || px >= a.w || py < 0 || py >= a.h) return;
    if (!a.occluded.row(py)[px]) {
        const int m = a.mask.row(py)[px] ? 1 : 0;  // mask is read as bool
        float2 c = a.fgbg[i];
        c.x = c.x + static_cast<float>(m);
        c.y = c.y + static_cast<float>(1 - m);
        a.fgbg[i] = c;
    }
}

// ---- a14: foreground probability (reference ObjTSDF::computeFgProbs, ObjTSDF.cpp:218-226) --------

__device__ __forceinline__ float fg_prob(float fg, float bg) {
    const float s = fg + bg;
    float p = (s != 0.f) ? fg / s : 0.f;  // cv::cuda::divide: x / 0 := 0
    if (p != p) p = 0.f;                  // compare(NE) + setTo(0)
    return p;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_fg_probs(const float2* __restrict__ fgbg,
                                                     float* __restrict__ probs,
                                                     uint8_t* __restrict__ volMask, size_t total) {
    const size_t g = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    const size_t i0 = g * VEC;
    if (i0 >= total) return;
    if constexpr (VEC == 4) {
        const float4 a = reinterpret_cast<const float4*>(fgbg)[2 * g];
        const float4 b = reinterpret_cast<const float4*>(fgbg)[2 * g + 1];
        float4 p;
        p.x = fg_prob(a.x, a.y);
        p.y = fg_prob(a.z, a.w);
        p.z = fg_prob(b.x, b.y);
        p.w = fg_prob(b.z, b.w);
        reinterpret_cast<float4*>(probs)[g] = p;
        uchar4 m;
        m.x = p.x > 0.5f ? 255 : 0;
        m.y = p.y > 0.5f ? 255 : 0;
        m.z = p.z > 0.5f ? 255 : 0;
        m.w = p.w > 0.5f ? 255 : 0;
        reinterpret_cast<uchar4*>(volMask)[g] = m;
    } else {
        const float2 c = fgbg[i0];
        const float p = fg_prob(c.x, c.y);
        probs[i0] = p;
        volMask[i0] = p > 0.5f ? 255 : 0;
    }
}

// ---- a11 (literal form): raycastWeights = fgVolMask ? weights : 0 (ObjTSDF.cpp:209-210) ----------

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_mask_weights(const float* __restrict__ w,
                                                         const uint8_t* __restrict__ m,
                                                         float* __restrict__ out, size_t total) {
    const size_t g = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    const size_t i0 = g * VEC;
    if (i0 >= total) return;
    if constexpr (VEC == 4) {
        const float4 wv = reinterpret_cast<const float4*>(w)[g];
        const uchar4 mv = reinterpret_cast<const uchar4*>(m)[g];
        float4 o;
        o.x = mv.x ? wv.x : 0.f;
        o.y = mv.y ? wv.y : 0.f;
        o.z = mv.z ? wv.z : 0.f;
        o.w = mv.w ? wv.w : 0.f;
        reinterpret_cast<float4*>(out)[g] = o;
    } else {
        out[i0] = m[i0] ? w[i0] : 0.f;
    }
}

inline bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

int emf_hip_updateTSDF(const emf_image_t* depth, const emf_image_t* assocWeights, float* tsdf,
                       float* weights, uint8_t* brickFlags, const float R_OC[9],
                       const float t_OC[3], const float K[9], const int32_t res[3],
                       float voxelSize, float truncdist, float maxWeight,
                       const emf_image_t* invLambda, emf_stream_t stream) {
    EMF_TRY(check_image(depth, 4, "updateTSDF: depth"));
    if (invLambda) {
        EMF_TRY(check_image(invLambda, 4, "updateTSDF: invLambda"));
        EMF_TRY(check_same_size(depth, invLambda, "depth", "invLambda"));
    }
    EMF_TRY(check_image(assocWeights, 4, "updateTSDF: assocWeights"));
    EMF_TRY(check_same_size(depth, assocWeights, "depth", "assocWeights"));
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_REQUIRE_PTR(R_OC);
    EMF_REQUIRE_PTR(t_OC);
    EMF_REQUIRE_PTR(K);
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f) || !(truncdist > 0.f))
        return fail(EMF_E_ARG, "updateTSDF: voxelSize %g / truncdist %g must be > 0", voxelSize,
                    truncdist);
    IntegrateGeom a;
    a.depth = img<const float>(depth);
    a.assoc = img<const float>(assocWeights);
    a.invLambda = invLambda ? img<const float>(invLambda) : Img<const float>{nullptr, 0};
    a.w = depth->width;
    a.h = depth->height;
    a.R = m33_from(R_OC);
    a.t = v3_from(t_OC);
    a.K = m33_from(K);
    a.pinhole = is_pinhole(a.K);
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    a.truncdist = truncdist;
    a.maxWeight = maxWeight;
    if (res[0] % 4 == 0 && aligned16(tsdf) && aligned16(weights)) {
        TileGrid g{static_cast<int>(ceil_div(res[0], kTileX)), static_cast<int>(ceil_div(res[1], kTileY)),
                   static_cast<int>(ceil_div(res[2], kTileZ))};
        hipLaunchKernelGGL(k_update_tsdf_tiled, dim3(static_cast<unsigned>(g.ntx) * g.nty * g.ntz),
                           dim3(kBlock), 0, as_stream(stream), a, tsdf, weights, brickFlags, g);
    } else {
        const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
        hipLaunchKernelGGL(k_update_tsdf_linear, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), a, tsdf, weights, brickFlags);
    }
    if (brickFlags) {  // refresh the dilated copy that the raycast consumes
        const int nbx = bricks_along(res[0]), nby = bricks_along(res[1]), nbz = bricks_along(res[2]);
        const size_t nb = static_cast<size_t>(nbx) * nby * nbz;
        hipLaunchKernelGGL(k_dilate_flags, dim3(ceil_div(nb, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), brickFlags, brickFlags + nb, nbx, nby, nbz);
    }
    return launch_status("updateTSDF");
}

int emf_hip_computeInvLambda(const float K[9], emf_image_t* invLambda, emf_stream_t stream) {
    EMF_REQUIRE_PTR(K);
    EMF_TRY(check_image(invLambda, 4, "computeInvLambda: invLambda"));
    const int w = invLambda->width, h = invLambda->height;
    hipLaunchKernelGGL(k_inv_lambda, dim3(static_cast<unsigned>(ceil_div(w, kBlock)), h),
                       dim3(kBlock), 0, as_stream(stream), m33_from(K), img<float>(invLambda), w, h);
    return launch_status("computeInvLambda");
}

int emf_hip_resetBrickFlags(uint8_t* brickFlags, const int32_t res[3], emf_stream_t stream) {
    EMF_REQUIRE_PTR(brickFlags);
    EMF_TRY(check_res(res));
    const size_t n = static_cast<size_t>(bricks_along(res[0])) * bricks_along(res[1]) *
                     bricks_along(res[2]);
    // raw flags followed by the dilated flags: a zeroed volume is deep-uniform everywhere
    hipLaunchKernelGGL(k_fill_bytes, dim3(ceil_div(2 * n, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), brickFlags, 2 * n, static_cast<uint8_t>(kBrickAllZero));
    return launch_status("resetBrickFlags");
}

int emf_hip_computeTSDFGrads(const float* tsdf, float* grads, const int32_t res[3],
                             emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(grads);
    EMF_TRY(check_res(res));
    const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
    hipLaunchKernelGGL(k_tsdf_grads, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), tsdf, grads, i3_from(res));
    return launch_status("computeTSDFGrads");
}

int emf_hip_updateFgBgProbs(const emf_image_t* mask, const emf_image_t* occluded,
                            const float* tsdf, const float* weights, float* fgBgProbs,
                            const float R_OC[9], const float t_OC[3], const float K[9],
                            const int32_t res[3], float voxelSize, emf_stream_t stream) {
    EMF_TRY(check_image(mask, 1, "updateFgBgProbs: mask"));
    EMF_TRY(check_image(occluded, 1, "updateFgBgProbs: occluded"));
    EMF_TRY(check_same_size(mask, occluded, "mask", "occluded"));
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_REQUIRE_PTR(fgBgProbs);
    EMF_REQUIRE_PTR(R_OC);
    EMF_REQUIRE_PTR(t_OC);
    EMF_REQUIRE_PTR(K);
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f)) return fail(EMF_E_ARG, "updateFgBgProbs: voxelSize must be > 0");
    FgBgArgs a;
    a.mask = img<const uint8_t>(mask);
    a.occluded = img<const uint8_t>(occluded);
    a.w = mask->width;
    a.h = mask->height;
    a.tsdf = tsdf;
    a.weights = weights;
    a.fgbg = reinterpret_cast<float2*>(fgBgProbs);
    a.R = m33_from(R_OC);
    a.t = v3_from(t_OC);
    a.K = m33_from(K);
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
    hipLaunchKernelGGL(k_update_fgbg, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), a);
    return launch_status("updateFgBgProbs");
}

int emf_hip_computeFgProbs(const float* fgBgProbs, float* fgProbs, uint8_t* fgVolMask,
                           const int32_t res[3], emf_stream_t stream) {
    EMF_REQUIRE_PTR(fgBgProbs);
    EMF_REQUIRE_PTR(fgProbs);
    EMF_REQUIRE_PTR(fgVolMask);
    EMF_TRY(check_res(res));
    const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
    if (voxels % 4 == 0 && aligned16(fgBgProbs) && aligned16(fgProbs) &&
        reinterpret_cast<uintptr_t>(fgVolMask) % 4 == 0) {
        hipLaunchKernelGGL(k_fg_probs<4>, dim3(ceil_div(voxels / 4, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), reinterpret_cast<const float2*>(fgBgProbs), fgProbs,
                           fgVolMask, voxels);
    } else {
        hipLaunchKernelGGL(k_fg_probs<1>, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), reinterpret_cast<const float2*>(fgBgProbs), fgProbs,
                           fgVolMask, voxels);
    }
    return launch_status("computeFgProbs");
}

int emf_hip_maskRaycastWeights(const float* weights, const uint8_t* fgVolMask,
                               float* raycastWeights, const int32_t res[3], emf_stream_t stream) {
    EMF_REQUIRE_PTR(weights);
    EMF_REQUIRE_PTR(fgVolMask);
    EMF_REQUIRE_PTR(raycastWeights);
    EMF_TRY(check_res(res));
    const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
    if (voxels % 4 == 0 && aligned16(weights) && aligned16(raycastWeights) &&
        reinterpret_cast<uintptr_t>(fgVolMask) % 4 == 0) {
        hipLaunchKernelGGL(k_mask_weights<4>, dim3(ceil_div(voxels / 4, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), weights, fgVolMask, raycastWeights, voxels);
    } else {
        hipLaunchKernelGGL(k_mask_weights<1>, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), weights, fgVolMask, raycastWeights, voxels);
    }
    return launch_status("maskRaycastWeights");
}

}  // extern "C"
