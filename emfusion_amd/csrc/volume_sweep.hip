// volume_sweep.hip -- dense voxel sweeps: TSDF integration, gradient volume, foreground/background
// counts, foreground probability.  All are HBM-streaming kernels over continuous (Nz*Ny) x Nx
// volumes; each lane owns VEC consecutive x-voxels so the read-modify-write streams are 16-byte
// vector accesses along the fastest axis (1 KiB per wave instruction).
#include "common.hpp"

namespace emf_hip {
namespace {

constexpr int kBlock = 256;

// ---- a7: TSDF integration (reference kernel_updateTSDF, TSDF.cu:327-401) -------------------------

struct IntegrateArgs {
    Img<const float> depth, assoc;
    int w, h;
    float* tsdf;
    float* weights;
    M33 R;  // volume -> camera
    V3 t;
    M33 K;
    I3 n;
    float voxelSize, truncdist, maxWeight;
};

enum : int { kSkip = 0, kZeroIfUnseen = 1, kNegIfUnseen = 2, kFuse = 3 };

// Geometry of one voxel: which branch of the reference kernel it takes, and for the fusing branch
// the truncated SDF sample and its association weight.
__device__ __forceinline__ int classify_voxel(const IntegrateArgs& a, const V3& half, int x, int y,
                                              int z, float& tsdfSample, float& assocW) {
    const V3 pobj = v3((static_cast<float>(x) - half.x) * a.voxelSize,
                       (static_cast<float>(y) - half.y) * a.voxelSize,
                       (static_cast<float>(z) - half.z) * a.voxelSize);
    const V3 pcam = mul(a.R, pobj) + a.t;
    if (pcam.z <= 0.f) return kZeroIfUnseen;  // TSDF.cu:351-356
    const V3 proj = mul(a.K, pcam);
    const int px = __float2int_rn(proj.x / proj.z);  // round-half-even, TSDF.cu:360-361
    const int py = __float2int_rn(proj.y / proj.z);
    if (px < 0 || px >= a.w || py < 0 || py >= a.h) return kSkip;
    const float d = a.depth.row(py)[px];
    if (d <= 0.f) return kZeroIfUnseen;  // TSDF.cu:367-372
    // lambda from the ROUNDED pixel (TSDF.cu:374-377)
    const float lambda = norm(v3((static_cast<float>(px) - a.K.r0.z) / a.K.r0.x,
                                 (static_cast<float>(py) - a.K.r1.z) / a.K.r1.y, 1.f));
    const float sdf = d - (1.f / lambda) * norm(pcam);
    if (sdf >= -a.truncdist) {
        tsdfSample = copysignf(fminf(1.f, fabsf(sdf / a.truncdist)), sdf);
        assocW = sdf < a.truncdist ? a.assoc.row(py)[px] : 1.f;  // free space fuses with 1 (Q8)
        return kFuse;
    }
    return kNegIfUnseen;  // TSDF.cu:398-400
}

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
    using type = float4;
};
template <>
struct VecT<1> {
    using type = float;
};

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_update_tsdf(const IntegrateArgs a) {
    using vec_t = typename VecT<VEC>::type;
    const size_t groupsPerRow = static_cast<size_t>(a.n.x) / VEC;
    const size_t rows = static_cast<size_t>(a.n.y) * a.n.z;
    const size_t gid = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (gid >= rows * groupsPerRow) return;
    const size_t row = gid / groupsPerRow;
    const int x0 = static_cast<int>(gid - row * groupsPerRow) * VEC;
    const int y = static_cast<int>(row % a.n.y), z = static_cast<int>(row / a.n.y);
    const V3 half = half_extent(a.n);

    int kind[VEC];
    float samp[VEC], aw[VEC];
    bool any = false, anyFuse = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        samp[e] = 0.f;
        aw[e] = 0.f;
        kind[e] = classify_voxel(a, half, x0 + e, y, z, samp[e], aw[e]);
        any |= kind[e] != kSkip;
        anyFuse |= kind[e] == kFuse;
    }
    if (!any) return;  // whole group projects outside the image: no memory touched

    const size_t base = row * a.n.x + x0;
    float wv[VEC], tv[VEC];
    {
        const vec_t wl = *reinterpret_cast<const vec_t*>(a.weights + base);
        memcpy(wv, &wl, sizeof(wl));
    }
    // Constant writes (tsdf := 0 / -1 on never-observed voxels) need no read of the old tsdf when
    // every voxel of the group takes one; otherwise the old values are loaded so the vector store
    // writes back untouched voxels bit-for-bit.
    bool allConst = true, anyConst = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const bool c = (kind[e] == kZeroIfUnseen || kind[e] == kNegIfUnseen) && wv[e] == 0;
        allConst &= c;
        anyConst |= c;
    }
    if (anyFuse || (anyConst && !allConst)) {
        const vec_t tl = *reinterpret_cast<const vec_t*>(a.tsdf + base);
        memcpy(tv, &tl, sizeof(tl));
    }
    bool wroteT = false, wroteW = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float pw = wv[e];
        if (kind[e] == kFuse) {
            const float nw = aw[e];
            if (pw + nw > 0) {  // TSDF.cu:392-397
                tv[e] = (pw * tv[e] + nw * samp[e]) / (pw + nw);
                wv[e] = fminf(pw + nw, a.maxWeight);
                wroteT = wroteW = true;
            }
        } else if (kind[e] == kZeroIfUnseen) {
            if (pw == 0) {
                tv[e] = 0.f;
                wroteT = true;
            }
        } else if (kind[e] == kNegIfUnseen) {
            if (pw == 0) {
                tv[e] = -1.f;
                wroteT = true;
            }
        }
    }
    if (wroteT) {
        vec_t o;
        memcpy(&o, tv, sizeof(o));
        *reinterpret_cast<vec_t*>(a.tsdf + base) = o;
    }
    if (wroteW) {
        vec_t o;
        memcpy(&o, wv, sizeof(o));
        *reinterpret_cast<vec_t*>(a.weights + base) = o;
    }
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
                       float* weights, const float R_OC[9], const float t_OC[3],
                       const float K[9], const int32_t res[3], float voxelSize, float truncdist,
                       float maxWeight, emf_stream_t stream) {
    EMF_TRY(check_image(depth, 4, "updateTSDF: depth"));
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
    IntegrateArgs a;
    a.depth = img<const float>(depth);
    a.assoc = img<const float>(assocWeights);
    a.w = depth->width;
    a.h = depth->height;
    a.tsdf = tsdf;
    a.weights = weights;
    a.R = m33_from(R_OC);
    a.t = v3_from(t_OC);
    a.K = m33_from(K);
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    a.truncdist = truncdist;
    a.maxWeight = maxWeight;
    const size_t voxels = static_cast<size_t>(res[0]) * res[1] * res[2];
    if (res[0] % 4 == 0 && aligned16(tsdf) && aligned16(weights)) {
        hipLaunchKernelGGL(k_update_tsdf<4>, dim3(ceil_div(voxels / 4, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), a);
    } else {
        hipLaunchKernelGGL(k_update_tsdf<1>, dim3(ceil_div(voxels, kBlock)), dim3(kBlock), 0,
                           as_stream(stream), a);
    }
    return launch_status("updateTSDF");
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
