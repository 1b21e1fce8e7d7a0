// device_core.hpp -- device functions shared by the per-volume kernels (level-1 ABI) and the
// batched, model-table-driven kernels (native path): voxel classification / fusion for TSDF
// integration, brick uniformity flags, and the ray march.
//
// Everything that decides a branch or a step is evaluated in the reference's operation order
// (see common.hpp); the performance devices used here -- lazy weight lookups, on-the-fly
// gradients, brick-flag short cuts, 8-byte pair loads -- never change a computed value.
#pragma once

#include "common.hpp"

namespace emf_hip {

// ---- brick uniformity flags ---------------------------------------------------------------------
// One byte per 4x4x4 brick of a TSDF volume: non-zero iff EVERY voxel of the brick holds exactly
// the same one of the three values the integration writes wholesale.  (4^3 rather than 8^3: on the
// bench scene 57 % of the exactly-1.0 free-space voxels sit in bricks whose 26 neighbours are
// uniform too, against 23 % for 8^3 bricks.)
// The DILATED byte additionally carries, in bits 3..4, the erosion depth D in 1..3: every brick
// within Chebyshev distance D shares the class (D = 0 is stored as the whole byte being 0).
constexpr int kBrick = 4;
constexpr int kBrickShift = 2;
constexpr int kMaxBrickDepth = 3;
enum : uint8_t { kBrickMixed = 0, kBrickAllZero = 1, kBrickAllOne = 2, kBrickAllNegOne = 4 };

__host__ __device__ __forceinline__ int bricks_along(int n) { return (n + kBrick - 1) >> kBrickShift; }

// 3-bit class of one voxel value (bit set = "equals that constant"); AND over a brick gives the flag
__device__ __forceinline__ unsigned uniform_bits(float v) {
    return (v == 0.f ? kBrickAllZero : 0u) | (v == 1.f ? kBrickAllOne : 0u) |
           (v == -1.f ? kBrickAllNegOne : 0u);
}
__device__ __forceinline__ float brick_constant(uint8_t f) {
    return f == kBrickAllZero ? 0.f : (f == kBrickAllOne ? 1.f : -1.f);
}
__host__ __device__ __forceinline__ size_t brick_count(const I3& n) {
    return static_cast<size_t>(bricks_along(n.x)) * bricks_along(n.y) * bricks_along(n.z);
}
// class | (D << 3) where D in 1..kMaxBrickDepth is the largest depth such that every in-volume
// brick within Chebyshev distance D of (bx,by,bz) has the same non-zero class; 0 if D would be 0
__device__ __forceinline__ uint8_t dilated_flag(const uint8_t* __restrict__ raw, int nbx, int nby,
                                                int nbz, int bx, int by, int bz) {
    const uint8_t c = raw[(static_cast<size_t>(bz) * nby + by) * nbx + bx];
    if (c == kBrickMixed) return kBrickMixed;
    // deep search only for free space (+1): that is where rays spend their steps; unseen (0) and
    // behind-surface (-1) space fills most of a volume and would make this pass expensive
    const int maxDepth = c == kBrickAllOne ? kMaxBrickDepth : 1;
    int depth = 0;
    for (int D = 1; D <= maxDepth; ++D) {
        // only the shell at distance exactly D is new
        bool ok = true;
        for (int dz = -D; dz <= D && ok; ++dz)
            for (int dy = -D; dy <= D && ok; ++dy) {
                const bool edge = dz == -D || dz == D || dy == -D || dy == D;
                const int z = bz + dz, y = by + dy;
                if (y < 0 || z < 0 || y >= nby || z >= nbz) continue;
                const uint8_t* row = raw + (static_cast<size_t>(z) * nby + y) * nbx;
                if (edge) {
                    for (int dx = -D; dx <= D && ok; ++dx) {
                        const int x = bx + dx;
                        if (x >= 0 && x < nbx) ok = row[x] == c;
                    }
                } else {
                    if (bx - D >= 0) ok = row[bx - D] == c;
                    if (ok && bx + D < nbx) ok = row[bx + D] == c;
                }
            }
        if (!ok) break;
        depth = D;
    }
    return depth ? static_cast<uint8_t>(c | (depth << 3)) : kBrickMixed;
}
__device__ __forceinline__ uint8_t flag_class(uint8_t f) { return f & 7u; }
__device__ __forceinline__ int flag_depth(uint8_t f) { return f >> 3; }

// ---- integration --------------------------------------------------------------------------------

struct IntegrateGeom {
    Img<const float> depth, assoc;
    Img<const float> invLambda;  // optional per-pixel 1 / lambda table (data == nullptr: inline)
    int w, h;
    M33 R;  // volume -> camera
    V3 t;
    M33 K;
    I3 n;
    float voxelSize, truncdist, maxWeight;
    bool pinhole;  // K = (fx 0 cx; 0 fy cy; 0 0 1): set by is_pinhole(K) on the host
    // float bits of the largest depth of the frame (0 bits = no valid pixel), made by the launch that
    // lists the boxes, or nullptr: lets a tile of unseen voxels that lies wholly behind everything the
    // camera sees skip the signed distance (integrate_tile)
    const unsigned* maxDepthBits = nullptr;
};

#ifndef EMF_INT_WPE
#define EMF_INT_WPE 5  // waves per SIMD the tiled integration kernels are compiled for
#endif

inline bool is_pinhole(const M33& K) {
    return K.r0.y == 0.f && K.r1.x == 0.f && K.r2.x == 0.f && K.r2.y == 0.f && K.r2.z == 1.f;
}

enum : int { kSkip = 0, kZeroIfUnseen = 1, kNegIfUnseen = 2, kFuse = 3 };

// 1 / lambda of the reference (TSDF.cu:374-377, 380): lambda = |((px - cx) / fx, (py - cy) / fy, 1)|
__device__ __forceinline__ float inv_lambda_at(const M33& K, int px, int py) {
    const float lambda = norm(v3((static_cast<float>(px) - K.r0.z) / K.r0.x,
                                 (static_cast<float>(py) - K.r1.z) / K.r1.y, 1.f));
    return 1.f / lambda;
}

// voxel centre in the camera frame (reference TSDF.cu:345-349)
__device__ __forceinline__ V3 voxel_in_camera(const IntegrateGeom& a, const V3& half, int x, int y,
                                              int z) {
    const V3 pobj = v3((static_cast<float>(x) - half.x) * a.voxelSize,
                       (static_cast<float>(y) - half.y) * a.voxelSize,
                       (static_cast<float>(z) - half.z) * a.voxelSize);
    return mul(a.R, pobj) + a.t;
}

// Which branch of reference kernel_updateTSDF (TSDF.cu:327-401) a voxel takes; for the fusing
// branch also the truncated SDF sample and its association weight.
__device__ __forceinline__ int classify_voxel(const IntegrateGeom& a, const V3& half, int x, int y,
                                              int z, float& tsdfSample, float& assocW) {
    const V3 pcam = voxel_in_camera(a, half, x, y, z);
    if (pcam.z <= 0.f) return kZeroIfUnseen;  // TSDF.cu:351-356
    const V3 proj = mul(a.K, pcam);
    const int px = __float2int_rn(proj.x / proj.z);  // round-half-even, TSDF.cu:360-361
    const int py = __float2int_rn(proj.y / proj.z);
    if (px < 0 || px >= a.w || py < 0 || py >= a.h) return kSkip;
    const float d = a.depth.row(py)[px];
    if (d <= 0.f) return kZeroIfUnseen;  // TSDF.cu:367-372
    // lambda from the ROUNDED pixel (TSDF.cu:374-377): a function of (px, py) and K alone, so the
    // caller may pass it as a table of the same floats (inv_lambda_at evaluated once per pixel)
    const float il = a.invLambda.data ? a.invLambda.row(py)[px] : inv_lambda_at(a.K, px, py);
    const float sdf = d - il * norm(pcam);
    if (sdf >= -a.truncdist) {
        tsdfSample = copysignf(fminf(1.f, fabsf(sdf / a.truncdist)), sdf);
        assocW = sdf < a.truncdist ? a.assoc.row(py)[px] : 1.f;  // free space fuses with 1 (Q8)
        return kFuse;
    }
    return kNegIfUnseen;  // TSDF.cu:398-400
}

// Apply the branch to (tsdf, weight); returns bit0 = tsdf changed, bit1 = weight changed.
__device__ __forceinline__ int apply_voxel(int kind, float samp, float aw, float maxWeight,
                                           float& tv, float& wv) {
    const float pw = wv;
    if (kind == kFuse) {
        if (pw + aw > 0) {  // TSDF.cu:392-397
            const float nt = (pw * tv + aw * samp) / (pw + aw), nw = fminf(pw + aw, maxWeight);
            // free space that has reached the weight cap re-fuses to the very same bits: no store
            const int changed = (__float_as_uint(nt) != __float_as_uint(tv) ? 1 : 0) |
                                (__float_as_uint(nw) != __float_as_uint(wv) ? 2 : 0);
            tv = nt;
            wv = nw;
            return changed;
        }
    } else if (kind == kZeroIfUnseen) {
        // "changed" only if the stored bits change: an unseen voxel holds +0 (reset, or this branch
        // earlier) or -1, and most of the volume behind the camera / behind surfaces would otherwise
        // be rewritten with the value it already has, every frame
        if (pw == 0 && __float_as_uint(tv) != 0u) {
            tv = 0.f;
            return 1;
        }
    } else if (kind == kNegIfUnseen) {
        if (pw == 0 && __float_as_uint(tv) != 0xbf800000u) {
            tv = -1.f;
            return 1;
        }
    }
    return 0;
}

// apply_voxel for a voxel whose weight is known to be 0 and whose tsdf has NOT been loaded: the branch
// either determines the new value whatever the old one was (a finite old value assumed -- the owner of
// the unseen-tile map guarantees it), or keeps the old one.  Returns true = "keeps it: load it".
//   kFuse:  (0 * old + aw * samp) / (0 + aw): 0 * old is a zero of either sign and vanishes in the sum
//           unless aw * samp is itself -0 (a denormal weight), which is left to the loaded form;
//           the new weight is min(0 + aw, maxWeight)
__device__ __forceinline__ bool apply_unseen(int kind, float samp, float aw, float maxWeight, float& tv,
                                             float& wv) {
    wv = 0.f;
    if (kind == kFuse) {
        const float prod = aw * samp;
        if (0.f + aw > 0 && __float_as_uint(prod) != 0x80000000u) {  // TSDF.cu:392-397
            tv = (0.f + prod) / (0.f + aw);
            wv = fminf(0.f + aw, maxWeight);
            return false;
        }
        return true;
    }
    if (kind == kZeroIfUnseen) {
        tv = 0.f;
        return false;
    }
    if (kind == kNegIfUnseen) {
        tv = -1.f;
        return false;
    }
    return true;  // kSkip
}

// Conservative frustum test for an axis-aligned box of voxels [x0,x1] x [y0,y1] x [z0,z1]
// (inclusive voxel indices): true only if EVERY voxel inside takes the kSkip branch, i.e.
// projects strictly in front of the camera and outside the image.  Perspective projection maps
// the box (convex, in front of the camera) into the convex hull of its projected corners, so if
// all 8 corners lie beyond the same image border by more than `margin` pixels, so does every
// voxel; the margin (1 px against a decision threshold of 0.5 px) absorbs float rounding.
__device__ __forceinline__ bool box_outside_image(const IntegrateGeom& a, const V3& half, int x0,
                                                  int y0, int z0, int x1, int y1, int z1) {
    const float margin = 1.f;
    bool left = true, right = true, top = true, bottom = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const V3 p = voxel_in_camera(a, half, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0,
                                     (k & 4) ? z1 : z0);
        if (!(p.z > 1e-3f)) return false;  // touches the camera plane: not cullable
        const V3 q = mul(a.K, p);
        const float u = q.x / q.z, v = q.y / q.z;
        left &= u < -0.5f - margin;
        right &= u > static_cast<float>(a.w) - 0.5f + margin;
        top &= v < -0.5f - margin;
        bottom &= v > static_cast<float>(a.h) - 0.5f + margin;
    }
    return left || right || top || bottom;
}

// ---- tiled integration: one workgroup (256 lanes) per 32 x 8 x 8 voxel tile -----------------------

constexpr int kTileX = 32, kTileY = 8, kTileZ = 8;

// Lane l projects corner (l & 7) of the tile; the wave votes.  Every wave of the workgroup computes
// the same votes, so the result is block-uniform.  Conservative test (see box_outside_image for the
// argument): the tile is skipped only if all 8 corner voxels are in front of the camera and beyond
// the SAME image border by more than a pixel.
__device__ __forceinline__ bool tile_culled(const IntegrateGeom& a, const V3& half, int x0, int y0,
                                            int z0) {
    const int x1 = min(x0 + kTileX, a.n.x) - 1, y1 = min(y0 + kTileY, a.n.y) - 1,
              z1 = min(z0 + kTileZ, a.n.z) - 1;
    const int k = threadIdx.x & 7;
    const V3 p = voxel_in_camera(a, half, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0, (k & 4) ? z1 : z0);
    const V3 q = mul(a.K, p);
    const float u = q.x / q.z, v = q.y / q.z;
    const float margin = 1.f;
    if (!__all(p.z > 1e-3f)) return false;  // touches the camera plane: not cullable
    return __all(u < -0.5f - margin) || __all(u > static_cast<float>(a.w) - 0.5f + margin) ||
           __all(v < -0.5f - margin) || __all(v > static_cast<float>(a.h) - 0.5f + margin);
}

// K * p for the projection.  For a pinhole matrix (fx 0 cx; 0 fy cy; 0 0 1) the general product
// (k00 x + k01 y) + k02 z adds 0 * y = +-0 to k00 x, which changes nothing unless k00 x is itself a
// zero of the other sign -- and then the following + cx z (or the comparison / division that
// consumes the value) does not see the sign.  So the short form gives the same pixel; p is finite.
__device__ __forceinline__ V3 project(const IntegrateGeom& a, const V3& p) {
    if (a.pinhole)  // kernel-argument uniform
        return v3(a.K.r0.x * p.x + a.K.r0.z * p.z, a.K.r1.y * p.y + a.K.r1.z * p.z, p.z);
    return mul(a.K, p);
}

#ifndef EMF_INT_FAST
#define EMF_INT_FAST 1  // 0: always take the IEEE division / square root sequences (A/B builds)
#endif

// round(x / z) to the pixel grid WITHOUT the IEEE division when that is provably the same integer.
// q~ = x * rcp(z): v_rcp_f32 is accurate to 1 ulp and the product rounds once, so
// |q~ - fl(x / z)| <= |q~| * 2^-22.  Unless q~ lies within twice that of a rounding tie k + 1/2 (or
// is huge / not finite, where float -> int conversion or rcp of a denormal could differ), q~ and the
// correctly rounded quotient round to the same integer.  `risky` lanes divide (about one quotient
// in 1500 on a VGA image).
__device__ __forceinline__ bool quotient_is_risky(float q) {
    const float f = q - floorf(q);
    return !(fabsf(q) < 1048576.f) || fabsf(f - 0.5f) <= fabsf(q) * 0x1p-21f;
}

// round(x / z) and round(y / z) of the projection (round-half-even, TSDF.cu:360-361).  FAST: the
// reciprocal form wherever quotient_is_risky() lets it stand, the IEEE division otherwise -- the same
// integers (premise: v_rcp_f32 within 1 ulp for every z whose reciprocal is a normal float, swept over all
// 2^32 inputs by emf_hip_sweepFastPathPremises; adversarial quotients next to every tie:
// tests/test_gpu_fast_paths.py through emf_hip_debugPixelRounding).
template <bool FAST>
__device__ __forceinline__ void round_pixel(float x, float y, float z, bool behind, int& px, int& py) {
    if (FAST) {
        const float rz = __builtin_amdgcn_rcpf(z);
        float qx = x * rz, qy = y * rz;
        if (!behind && (quotient_is_risky(qx) || quotient_is_risky(qy))) {
            qx = x / z;
            qy = y / z;
        }
        px = __float2int_rn(qx);
        py = __float2int_rn(qy);
    } else {
        px = __float2int_rn(x / z);
        py = __float2int_rn(y / z);
    }
}

// Where a voxel lands in the image: everything of classify_voxel that needs no memory.
struct VoxelShot {
    int px, py;    // rounded pixel (valid only if inImage)
    bool behind;   // p_cam.z <= 0  (TSDF.cu:351)
    bool inImage;  // in front of the camera and inside the image (TSDF.cu:362-365)
    float n2;      // |p_cam|^2 summed in norm()'s order; the sqrt is taken where it is needed
};
__device__ __forceinline__ VoxelShot shoot_voxel(const IntegrateGeom& a, const V3& half, int x,
                                                 int y, int z) {
    VoxelShot s;
    const V3 pcam = voxel_in_camera(a, half, x, y, z);
    s.behind = pcam.z <= 0.f;
    const V3 proj = project(a, pcam);
    // for `behind` voxels the quotients are never used (the reference returns before dividing)
    round_pixel<EMF_INT_FAST != 0>(proj.x, proj.y, proj.z, s.behind, s.px, s.py);
    s.inImage = !s.behind && s.px >= 0 && s.px < a.w && s.py >= 0 && s.py < a.h;
    s.n2 = pcam.x * pcam.x + pcam.y * pcam.y + pcam.z * pcam.z;
    return s;
}

// The side of the truncation band a voxel with a valid depth lies on, and its clamped sample
// (TSDF.cu:380-400): kFuse with the sample in [-1, 1] and `bandVoxel` = "sdf < truncdist: the pixel's
// association weight applies", or kNegIfUnseen.
// FAST: far from the surface only the SIDE of the band matters -- free space fuses the constant +1 with
// weight 1, voxels behind the band are left alone.  With the 1-ulp v_sqrt_f32, sdf~ = d - il * sqrt~
// differs from the reference's value by at most (|d| + |il n|) * 2^-21 (one ulp of the root, one rounding
// of the product, one of the difference, each side); outside twice that margin around +-truncdist the
// branch -- and the clamped sample -- are decided without the IEEE square root and without the division
// by truncdist.  (Premise swept over all 2^32 inputs by emf_hip_sweepFastPathPremises; distances within
// an ulp of +-truncdist: tests/test_gpu_fast_paths.py through emf_hip_debugBandDecision.)
template <bool FAST>
__device__ __forceinline__ int band_decision(float d, float il, float n2, float truncdist, float& tsdfSample,
                                             bool& bandVoxel) {
    bandVoxel = false;
    if (FAST) {
        const float t = il * __builtin_amdgcn_sqrtf(n2);
        const float approx = d - t, margin = (fabsf(d) + fabsf(t)) * 0x1p-20f;
        if (approx - margin > truncdist) {  // sdf > truncdist: |sdf / truncdist| >= 1 -> sample +1
            tsdfSample = 1.f;
            return kFuse;
        }
        if (approx + margin < -truncdist) return kNegIfUnseen;
    }
    const float sdf = d - il * sqrtf(n2);
    if (sdf >= -truncdist) {
        tsdfSample = copysignf(fminf(1.f, fabsf(sdf / truncdist)), sdf);
        bandVoxel = sdf < truncdist;
        return kFuse;
    }
    return kNegIfUnseen;
}

// The branch of kernel_updateTSDF a voxel takes, from its shot, the depth at its pixel and
// 1 / lambda (same decisions and arithmetic as classify_voxel, which the one-voxel-per-lane
// kernel uses).  For the fusing branch: the truncated SDF sample, and whether the association
// weight of the pixel applies (inside the truncation band) or the constant 1 (free space, Q8).
__device__ __forceinline__ int classify_shot(const IntegrateGeom& a, const VoxelShot& s, float d,
                                             float il, float& tsdfSample, bool& bandVoxel) {
    bandVoxel = false;
    if (s.behind) return kZeroIfUnseen;
    if (!s.inImage) return kSkip;
    if (d <= 0.f) return kZeroIfUnseen;
    return band_decision<EMF_INT_FAST != 0>(d, il, s.n2, a.truncdist, tsdfSample, bandVoxel);
}

// 16-byte volume accesses of the out-of-place sweep: streamed once, never re-read by this kernel, and
// the raycast running beside it lives on what stays in L2 -- mark them non-temporal (EMF_INT_NT=0: plain)
#ifndef EMF_INT_NT
#define EMF_INT_NT 1
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store4(float* p, const float4& v) {
    if (NT && EMF_INT_NT) {
        f4v t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
template <bool NT>
__device__ __forceinline__ float4 load4(const float* p) {
    if (NT && EMF_INT_NT) {
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *reinterpret_cast<const float4*>(p);
}

// Process the tile at voxel origin (x0, y0, z0) with the 256 lanes of the workgroup.
// lds: 32 unsigned words.  All lanes of the block must call this (it contains barriers).
//
// Each lane owns two groups of 4 consecutive x voxels (z and z + 4).  The reference kernel is a
// chain per voxel: project -> depth -> association weight -> volume read -> write.  Here
//   A. the 8 projections of the lane are computed first (pure arithmetic, shoot_voxel);
//   B. every load that does not depend on a loaded value is issued together: depth and 1 / lambda
//      at the 8 pixels, and the two 16-byte volume reads, predicated on "some voxel of the group
//      is behind the camera or lands in the image" (a property of the projection, not of depth);
//   C. the voxels are classified and fused; the association weight is read only by voxels inside
//      the truncation band (about one tile in ten touches it).
// What bounds it (scripts/probes/integrate_trace.hip): arithmetic.  ~70 VALU
// instructions per voxel -- two IEEE divisions to project, a square root, up to two more divisions
// to fuse -- put the VALU floor of the bench workload at ~0.30 ms; removing the load dependencies,
// staging the pixel window in LDS, persistent tile scheduling were all measured and gave nothing
// or lost, the 1 / lambda table and 6 waves per SIMD gave 9 %.  (The LDS window was built a second time in round 5
// for the 1280 x 960 / 1024^3 share, where the L1 tag rate reads 0.92 with raycast and sweep overlapped: depth and
// 1 / lambda over the bounding box of the tile's corner projections, 16-byte fills, ds_read in phase B, gather
// fall-back per tile and per voxel; bit-identical over 150 tests -- and slower there too, 3.59 -> 3.76 ms
// overlapped, 1.56 -> 1.74 ms alone, and 0.566 -> 0.587 ms per frame at 512^3: DESIGN.md 5.1d; in git history.)
//
// OUT = true is the out-of-place form behind emf_hip_integrateBatchedCulledOut: (tsdf, weights) are
// only READ, the integrated state goes to (tsdfOut, weightsOut), a second copy of the volume that
// already equals the first one wherever the previous integration changed nothing.  `force` (block-
// uniform; bit 0: tsdf, bit 1: weights) = "the previous integration changed that array in this
// tile": then every voxel of the array in the tile is written (integrated or copied), otherwise only
// the voxels that change now.  dirtyT / dirtyW are set when this call changes a tsdf / weight of the
// tile, i.e. they are the next call's `force`.  (Free space below the weight cap changes its weight
// every frame and its tsdf never: tracking the arrays apart halves what has to be stored there.)
// Same arithmetic, same values: only where they are stored differs.
template <bool OUT = false>
__device__ __forceinline__ void integrate_tile(const IntegrateGeom& a, float* __restrict__ tsdf,
                                               float* __restrict__ weights,
                                               uint8_t* __restrict__ bricks, int x0, int y0,
                                               int z0, unsigned* lds,
                                               float* __restrict__ tsdfOut = nullptr,
                                               float* __restrict__ weightsOut = nullptr,
                                               int force = 0, uint8_t* dirtyT = nullptr,
                                               uint8_t* dirtyW = nullptr, bool copyOnly = false,
                                               uint8_t* signPos = nullptr, uint8_t* signNeg = nullptr,
                                               uint8_t* unseen = nullptr) {
    const V3 half = half_extent(a.n);
    // copyOnly (OUT, block-uniform): the model is not integrated this frame (visibility gate closed),
    // its second copy only has to catch up
    if ((OUT && copyOnly) || tile_culled(a, half, x0, y0, z0)) {  // block-uniform: no divergent barrier
        if (OUT && force) {  // nothing to integrate, but the other copy is one integration behind here
            const int xg = threadIdx.x & 7, yy = (threadIdx.x >> 3) & 7, zs = threadIdx.x >> 6;
            const int x = x0 + 4 * xg, y = y0 + yy;
            if (x < a.n.x && y < a.n.y) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int z = z0 + zs + 4 * i;
                    if (z < a.n.z) {
                        const size_t b = (static_cast<size_t>(z) * a.n.y + y) * a.n.x + x;
                        if (force & 1) store4<OUT>(tsdfOut + b, load4<OUT>(tsdf + b));
                        if (force & 2) store4<OUT>(weightsOut + b, load4<OUT>(weights + b));
                    }
                }
            }
        }
        return;
    }
    if (!OUT) {  // in place: stores go where the loads came from
        tsdfOut = tsdf;
        weightsOut = weights;
    }
    // Unseen tile (block-uniform; emf_model_t.unseenTiles): every weight is 0, so no voxel's new value
    // depends on what the tile holds -- nothing is loaded up front, the new values are stored.
    const bool fast = unseen && !bricks && *unseen != 0;
    // Deep tile (block-uniform): an unseen tile every voxel of which lies behind the truncation band of
    // the farthest surface of the frame.  A voxel with a valid pixel then has
    //   sdf = d - |p_cam| / lambda(pix) <= dmax - z (1 - delta) < -truncdist,
    // (|p_cam| = z lambda(exact pixel); rounding to the pixel centre moves lambda by at most
    // 0.75 / min(fx, fy) =: delta; z >= the smallest z of the tile's corners, the camera z being linear
    // in the voxel index), i.e. the reference sets it to -1 (TSDF.cu:398-400) -- or to 0 on a depth
    // hole, or leaves it (pixel outside the image) -- without the distance ever mattering: no
    // 1 / lambda fetch, no square root, no band test for eight in ten tiles of the sweep.
    bool deep = false;
    if (fast && a.maxDepthBits && a.pinhole) {
        float zmin = __builtin_inff();
#pragma unroll
        for (int k = 0; k < 8; ++k)
            zmin = fminf(zmin, voxel_in_camera(a, half, x0 + ((k & 1) ? kTileX - 1 : 0), y0 + ((k & 2) ? kTileY - 1 : 0),
                                               z0 + ((k & 4) ? kTileZ - 1 : 0)).z);
        const float dmax = __uint_as_float(*a.maxDepthBits);
        const float delta = 0.75f / fminf(fabsf(a.K.r0.x), fabsf(a.K.r1.y));
        deep = zmin * (1.f - delta) * (1.f - 1e-5f) - 1e-6f > dmax + a.truncdist * (1.f + 1e-5f);  // (false for NaN / inf)
    }
    const int tid = threadIdx.x;
    const bool haveIl = a.invLambda.data != nullptr;
    bool gotWeight = false;  // a voxel of this lane has been fused into
    int anyChanged = 0;
    bool sawPos = false, sawNeg = false;  // signs among the tsdf values this lane holds at the end
    // 32 x 8 x 8 voxels = 8 x 2 x 2 bricks of 4^3: lds[bx + 8 * (by + 2 * bz)]
    if (bricks) {
        if (tid < 32) lds[tid] = 7u;
        __syncthreads();
    }
    const int xg = tid & 7, yy = (tid >> 3) & 7, zs = tid >> 6;
    const int x = x0 + 4 * xg, y = y0 + yy;
    unsigned bits[2] = {7u, 7u};  // one per z half of the tile = one per brick this lane touches
    if (x < a.n.x && y < a.n.y) {
        // ---- phase A: arithmetic only ------------------------------------------------------------
        // kept per voxel: packed pixel (py << 16 | px) and |p_cam|^2; per lane: two 8-bit masks
        unsigned pix[2][4], inMask = 0, behindMask = 0;
        float n2[2][4];
        bool live[2], touch[2];
        size_t base[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int z = z0 + zs + 4 * i;
            live[i] = z < a.n.z;
            const int zc = live[i] ? z : a.n.z - 1;
            base[i] = (static_cast<size_t>(zc) * a.n.y + y) * a.n.x + x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const VoxelShot s = shoot_voxel(a, half, x + e, y, zc);
                pix[i][e] = s.inImage ? (static_cast<unsigned>(s.py) << 16) | static_cast<unsigned>(s.px) : 0u;
                n2[i][e] = s.n2;
                if (s.inImage && live[i]) inMask |= 1u << (4 * i + e);
                if (s.behind && live[i]) behindMask |= 1u << (4 * i + e);
            }
            touch[i] = (((inMask | behindMask) >> (4 * i)) & 15u) != 0;
        }
        // ---- phase B: all independent loads in flight together -------------------------------------
        float d[2][4], il[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[i][e] = il[i][e] = 0.f;
                if (inMask & (1u << (4 * i + e))) {
                    const int px = pix[i][e] & 0xffffu, py = pix[i][e] >> 16;
                    d[i][e] = a.depth.row(py)[px];
                    if (haveIl && !deep) il[i][e] = a.invLambda.row(py)[px];
                }
            }
        float tv[2][4], wv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 tl = make_float4(0.f, 0.f, 0.f, 0.f), wl = tl;
            if (fast) {  // only a group that is merely copied needs its old values now
                if (!touch[i] && OUT && (force & 1) && live[i]) tl = load4<OUT>(tsdf + base[i]);
            } else {
                if ((bricks && live[i]) || touch[i] || (OUT && (force & 1) && live[i]))
                    tl = load4<OUT>(tsdf + base[i]);
                if (touch[i] || (OUT && (force & 2) && live[i])) wl = load4<OUT>(weights + base[i]);
            }
            tv[i][0] = tl.x; tv[i][1] = tl.y; tv[i][2] = tl.z; tv[i][3] = tl.w;
            wv[i][0] = wl.x; wv[i][1] = wl.y; wv[i][2] = wl.z; wv[i][3] = wl.w;
        }
        // ---- phase C: classify, fuse, store ---------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int changed = 0;
            if (touch[i] && fast) {
                unsigned keep = 0;  // voxels of the group that keep their old value
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    VoxelShot s;
                    s.px = pix[i][e] & 0xffffu;
                    s.py = pix[i][e] >> 16;
                    s.inImage = (inMask >> (4 * i + e)) & 1u;
                    s.behind = (behindMask >> (4 * i + e)) & 1u;
                    s.n2 = n2[i][e];
                    float samp = 0.f, aw = 1.f;
                    int kind;
                    if (deep) {  // classify_shot with "behind the band" known: TSDF.cu:351, 362, 367, 398
                        kind = s.behind ? kZeroIfUnseen : (!s.inImage ? kSkip : (d[i][e] <= 0.f ? kZeroIfUnseen : kNegIfUnseen));
                    } else {
                        const float ile = haveIl ? il[i][e] : inv_lambda_at(a.K, s.px, s.py);
                        bool band;
                        kind = classify_shot(a, s, d[i][e], ile, samp, band);
                        if (band) aw = a.assoc.row(s.py)[s.px];
                    }
                    if (apply_unseen(kind, samp, aw, a.maxWeight, tv[i][e], wv[i][e])) keep |= 1u << e;
                }
                if (keep) {  // pixels outside the image, association weight 0: rare, and only then a load
                    const float4 tl = load4<OUT>(tsdf + base[i]);
                    if (keep & 1u) tv[i][0] = tl.x;
                    if (keep & 2u) tv[i][1] = tl.y;
                    if (keep & 4u) tv[i][2] = tl.z;
                    if (keep & 8u) tv[i][3] = tl.w;
                }
                // what the tile held is not known: it counts as changed
                changed = 1 | ((wv[i][0] != 0.f || wv[i][1] != 0.f || wv[i][2] != 0.f || wv[i][3] != 0.f) ? 2 : 0);
                gotWeight = gotWeight || (changed & 2) != 0;
                anyChanged |= changed;
            } else if (touch[i]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    VoxelShot s;
                    s.px = pix[i][e] & 0xffffu;
                    s.py = pix[i][e] >> 16;
                    s.inImage = (inMask >> (4 * i + e)) & 1u;
                    s.behind = (behindMask >> (4 * i + e)) & 1u;
                    s.n2 = n2[i][e];
                    const float ile = haveIl ? il[i][e] : inv_lambda_at(a.K, s.px, s.py);
                    float samp = 0.f;
                    bool band;
                    const int kind = classify_shot(a, s, d[i][e], ile, samp, band);
                    const float aw = band ? a.assoc.row(s.py)[s.px] : 1.f;
                    changed |= apply_voxel(kind, samp, aw, a.maxWeight, tv[i][e], wv[i][e]);
                }
                gotWeight = gotWeight || (changed & 2) != 0;  // (weights only ever grow)
                anyChanged |= changed;
            }
            if (signPos && (touch[i] || (OUT && (force & 1) && live[i]))) {  // values that were loaded
                sawPos = sawPos || tv[i][0] > 0.f || tv[i][1] > 0.f || tv[i][2] > 0.f || tv[i][3] > 0.f;
                sawNeg = sawNeg || tv[i][0] < 0.f || tv[i][1] < 0.f || tv[i][2] < 0.f || tv[i][3] < 0.f;
            }
            if ((changed & 1) || (OUT && (force & 1) && live[i]))
                store4<OUT>(tsdfOut + base[i], make_float4(tv[i][0], tv[i][1], tv[i][2], tv[i][3]));
            if ((changed & 2) || (OUT && (force & 2) && live[i]))
                store4<OUT>(weightsOut + base[i], make_float4(wv[i][0], wv[i][1], wv[i][2], wv[i][3]));
            if (bricks && live[i])  // the lane's 4 voxels are one x-row of brick (xg, yy >> 2, i)
                bits[i] = uniform_bits(tv[i][0]) & uniform_bits(tv[i][1]) &
                          uniform_bits(tv[i][2]) & uniform_bits(tv[i][3]);
        }
    }
    // Sign maps (raycast far bounds, batched.hip): "this tile holds a positive / a negative tsdf".
    // Sticky and conservative: a voxel's value only ever changes in a lane that has it loaded, and
    // that lane reports the sign it leaves behind; a bit is never cleared here.
    if (signPos) {
        const bool p = __ballot(sawPos) != 0ull, n = __ballot(sawNeg) != 0ull;
        if ((tid & 63) == 0) {
            if (p) *signPos = 1;
            if (n) *signNeg = 1;
        }
    }
    // Unseen-tile map: cleared by whichever wave fuses into the tile (same value from all).  The flag is read
    // per wave at the top; should a wave read it after another one has cleared it, it merely takes the
    // loading path for its own voxels.
    if (unseen && __ballot(gotWeight) != 0ull && (tid & 63) == 0) *unseen = 0;
    if (OUT && dirtyT) {  // one store per wave that changed something (same value from all)
        const bool t = __ballot((anyChanged & 1) != 0) != 0ull, w = __ballot((anyChanged & 2) != 0) != 0ull;
        if ((tid & 63) == 0) {
            if (t) *dirtyT = 1;
            if (w) *dirtyW = 1;
        }
    }
    if (bricks) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (bits[i] != 7u) atomicAnd(&lds[xg + 8 * ((yy >> 2) + 2 * i)], bits[i]);
        __syncthreads();
        if (tid < 32) {
            const int bx = (x0 >> kBrickShift) + (tid & 7), by = (y0 >> kBrickShift) + ((tid >> 3) & 1),
                      bz = (z0 >> kBrickShift) + (tid >> 4);
            const int nbx = bricks_along(a.n.x), nby = bricks_along(a.n.y),
                      nbz = bricks_along(a.n.z);
            if (bx < nbx && by < nby && bz < nbz)
                bricks[(static_cast<size_t>(bz) * nby + by) * nbx + bx] =
                    static_cast<uint8_t>(lds[tid] == 7u ? kBrickMixed : lds[tid]);
        }
    }
}


// ---- ray march ----------------------------------------------------------------------------------

struct RayVolume {
    const float* tsdf;
    const float* grads;     // N^3 x 3 or nullptr (forward differences on the fly)
    const float* weights;
    const uint8_t* fg;      // foreground mask gating the weights, or nullptr
    const uint8_t* bricks;  // DILATED brick uniformity flags of `tsdf`, or nullptr
    bool blendFromFlags;    // answer lookups in deep-uniform bricks without gathering
    M33 R;                  // camera -> volume rotation
    V3 cam;                 // camera centre in the volume frame
    I3 n;
    float voxelSize, truncdist;
    float rcpVoxel;         // 1 / voxelSize checked by emf_hip_voxelReciprocal, or 0 = divide
};

// Host: may the march use `rcp` for this pose?  (march_wave.hpp: positions must stay below 1e30;
// with |t| <= 1e15 the slab test bounds |raylength| by ~2e15.)  NaN fails the comparison.
inline float usable_reciprocal(float rcp, const float t[3]) {
    const bool ok = fabsf(t[0]) <= 1e15f && fabsf(t[1]) <= 1e15f && fabsf(t[2]) <= 1e15f;
    return ok ? rcp : 0.f;
}
// Host: does the volume fit the 32-bit byte offsets of march_wave.hpp?
inline bool fits_offsets32(const int32_t res[3]) {
    return static_cast<unsigned long long>(res[0]) * res[1] * res[2] <= (1ull << 30);
}

struct RayHit {
    bool hit;
    float raylength;
    V3 vertex, normal;
    unsigned samples;  // main-loop samples taken (byte-model statistic)
    unsigned gathered; // ... of which read the volume (not answered by the brick flags)
    unsigned skipped;  // ... of which were fast-forwarded inside a deep-uniform brick
};

// two x-adjacent floats with one 8-byte load (only dword alignment is needed on gfx950)
struct __attribute__((packed, aligned(4))) FloatPair {
    float a, b;
};
__device__ __forceinline__ FloatPair load_pair(const float* p) {
    return *reinterpret_cast<const FloatPair*>(p);
}

__device__ __forceinline__ float trilinear_pairs(const float* __restrict__ vol, const Cell& c,
                                                 const I3& n) {
    const size_t sy = static_cast<size_t>(n.x), sz = sy * n.y;
    const float* p = vol + c.base;
    const FloatPair a = load_pair(p), b = load_pair(p + sy), d = load_pair(p + sz),
                    e = load_pair(p + sz + sy);
    return blend8(a.a, a.b, b.a, b.b, d.a, d.b, e.a, e.b, c.fx, c.fy, c.fz);
}

// weights as the march sees them: optionally gated by the foreground mask (ObjTSDF.cpp:209-210)
__device__ __forceinline__ float trilinear_weights(const RayVolume& v, const Cell& c) {
    const size_t sy = static_cast<size_t>(v.n.x), sz = sy * v.n.y;
    const float* p = v.weights + c.base;
    float w0 = p[0], w1 = p[1], w2 = p[sy], w3 = p[sy + 1], w4 = p[sz], w5 = p[sz + 1],
          w6 = p[sz + sy], w7 = p[sz + sy + 1];
    if (v.fg) {
        const uint8_t* m = v.fg + c.base;
        w0 = m[0] ? w0 : 0.f;
        w1 = m[1] ? w1 : 0.f;
        w2 = m[sy] ? w2 : 0.f;
        w3 = m[sy + 1] ? w3 : 0.f;
        w4 = m[sz] ? w4 : 0.f;
        w5 = m[sz + 1] ? w5 : 0.f;
        w6 = m[sz + sy] ? w6 : 0.f;
        w7 = m[sz + sy + 1] ? w7 : 0.f;
    }
    return blend8(w0, w1, w2, w3, w4, w5, w6, w7, c.fx, c.fy, c.fz);
}

// gradient at a hit: blend of the gradient volume, or of forward differences taken on the fly
__device__ __forceinline__ V3 gradient_at(const RayVolume& v, const Cell& c) {
    const size_t sy = static_cast<size_t>(v.n.x), sz = sy * v.n.y;
    float g[3][8];
    if (v.grads) {
        const float* p = v.grads + 3 * c.base;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t off = 3 * ((k & 1) + ((k >> 1) & 1) * sy + (k >> 2) * sz);
            g[0][k] = p[off];
            g[1][k] = p[off + 1];
            g[2][k] = p[off + 2];
        }
    } else {
        // a hit cell never touches the last index planes (the march requires v + 2 < N), so the
        // "zero on the last planes" rule of the gradient volume cannot apply here
        const float* p = v.tsdf + c.base;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t off = (k & 1) + ((k >> 1) & 1) * sy + (k >> 2) * sz;
            const float t0 = p[off];
            g[0][k] = p[off + 1] - t0;
            g[1][k] = p[off + sy] - t0;
            g[2][k] = p[off + sz] - t0;
        }
    }
    V3 r;
    r.x = blend8(g[0][0], g[0][1], g[0][2], g[0][3], g[0][4], g[0][5], g[0][6], g[0][7], c.fx,
                 c.fy, c.fz);
    r.y = blend8(g[1][0], g[1][1], g[1][2], g[1][3], g[1][4], g[1][5], g[1][6], g[1][7], c.fx,
                 c.fy, c.fz);
    r.z = blend8(g[2][0], g[2][1], g[2][2], g[2][3], g[2][4], g[2][5], g[2][6], g[2][7], c.fx,
                 c.fy, c.fz);
    return r;
}

// enterVolStep / exitVolStep (reference TSDF.cuh:31-63)
__device__ __forceinline__ float enter_step(const V3& d, const V3& c, const V3& bb) {
    const float sx = ((d.x > 0.f ? -bb.x : bb.x) - c.x) / d.x;
    const float sy = ((d.y > 0.f ? -bb.y : bb.y) - c.y) / d.y;
    const float sz = ((d.z > 0.f ? -bb.z : bb.z) - c.z) / d.z;
    return fmaxf(fmaxf(sx, sy), sz);
}
__device__ __forceinline__ float exit_step(const V3& d, const V3& c, const V3& bb) {
    const float sx = ((d.x > 0.f ? bb.x : -bb.x) - c.x) / d.x;
    const float sy = ((d.y > 0.f ? bb.y : -bb.y) - c.y) / d.y;
    const float sz = ((d.z > 0.f ? bb.z : -bb.z) - c.z) / d.z;
    return fminf(fminf(sx, sy), sz);
}

// TSDF value at a sample cell.  `v.bricks` holds DILATED uniformity flags: non-zero only if the
// brick AND all of its 26 neighbours are uniform with the same constant.  The 8 corners of a cell
// whose low corner lies in such a brick are spread over that brick and its neighbours at most, so
// all 8 voxels equal the constant and the blend is evaluated on it -- exactly the arithmetic the
// gather path performs on eight equal values -- with one byte load and no dependence on where the
// cell sits inside the brick (no straddling cases, hence no divergence between neighbouring rays).
struct BrickCache {
    int bx, by, bz;
    uint8_t flag;
};
__device__ __forceinline__ float sample_tsdf(const RayVolume& v, const V3& idx, const Cell& c,
                                             BrickCache& cache, uint8_t& uniform) {
    uniform = kBrickMixed;
    if (v.bricks) {
        const int bx = static_cast<int>(idx.x) >> kBrickShift,
                  by = static_cast<int>(idx.y) >> kBrickShift,
                  bz = static_cast<int>(idx.z) >> kBrickShift;
        if (bx != cache.bx || by != cache.by || bz != cache.bz) {
            const int nbx = bricks_along(v.n.x), nby = bricks_along(v.n.y);
            cache = BrickCache{bx, by, bz, v.bricks[(static_cast<size_t>(bz) * nby + by) * nbx + bx]};
        }
        uniform = cache.flag;  // class | depth << 3, or 0
        if (v.blendFromFlags && uniform != kBrickMixed) {
            // answer the lookup from the flag alone (saves the gather, but makes the gather of
            // mixed bricks wait for the flag byte: two dependent memory round trips)
            const float k = brick_constant(flag_class(uniform));
            return blend8(k, k, k, k, k, k, k, k, c.fx, c.fy, c.fz);
        }
    }
    // the flag byte (when its brick changed) and the 8 corners are independent loads: one round
    // trip.  In a deep-uniform brick the gathered blend IS the blend of the constant.
    return trilinear1(v.tsdf, c, v.n);
}

// How many further samples, taken `step` apart along `dir` from voxel-space position p, are
// GUARANTEED to have all 8 cell corners inside the (2D+1)^3 bricks around brick (bx, by, bz) --
// which all hold the same constant -- and to satisfy the march's p + 2 < N condition?
// Corners lx, lx + 1 lie in bricks bx - D .. bx + D  <=>  lo - 4D <= p < lo + 4 + 4D - 1.
// Conservative by far more than any accumulated rounding (0.02 voxel margin + one step held
// back): an underestimate at worst.
__device__ __forceinline__ int steps_inside_bricks(const V3& p, const V3& dir, float step,
                                                   float voxelSize, const I3& n, int bx, int by,
                                                   int bz, int depth) {
    const float eps = 0.02f;
    const float s = step / voxelSize;  // voxels per step (approximate is fine: only a bound)
    const float dv[3] = {dir.x * s, dir.y * s, dir.z * s};
    const float pp[3] = {p.x, p.y, p.z};
    const int lo[3] = {bx << kBrickShift, by << kBrickShift, bz << kBrickShift};
    const int nn[3] = {n.x, n.y, n.z};
    float k = 31.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int hi = min(lo[a] + kBrick + kBrick * depth - 1, nn[a] - 2);  // exclusive bound on p
        const int lw = max(lo[a] - kBrick * depth, 0);
        const float room = dv[a] > 0.f ? (static_cast<float>(hi) - eps) - pp[a]
                                       : pp[a] - (static_cast<float>(lw) + eps);
        const float ad = fabsf(dv[a]);
        if (ad > 1e-9f) k = fminf(k, room / ad);
        if (!(pp[a] < static_cast<float>(hi) - eps) || !(pp[a] > static_cast<float>(lw) + eps))
            k = 0.f;  // already at the margin of the guaranteed region
    }
    const int r = static_cast<int>(k) - 1;  // truncation + one step held back
    return r > 0 ? r : 0;
}

// One pixel of reference kernel_raycastTSDF (TSDF.cu:466-573).  `oldRaylength` is the incoming
// value of the raylength image (non-zero: do not search past another volume's hit).
__device__ __forceinline__ RayHit march_ray(const RayVolume& v, int x, int y, float fx, float fy,
                                            float cx, float cy, float oldRaylength) {
    RayHit out;
    out.hit = false;
    out.samples = 0;
    out.gathered = 0;
    out.skipped = 0;
    out.raylength = 0.f;
    out.vertex = v3(0.f, 0.f, 0.f);
    out.normal = v3(0.f, 0.f, 0.f);
    const V3 unproj = v3((static_cast<float>(x) - cx) / fx, (static_cast<float>(y) - cy) / fy, 1.f);
    const V3 rayv = mul(v.R, unproj);
    const V3 dir = rayv / norm(rayv);
    // (volSize - 1) / 2 is INTEGER division in the reference (TSDF.cu:490, Q2)
    const V3 bb = v3(static_cast<float>((v.n.x - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.y - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.z - 1) / 2) * v.voxelSize);
    const V3 half = half_extent(v.n);
    float raylength = enter_step(dir, v.cam, bb);
    float maxRay = exit_step(dir, v.cam, bb);
    raylength += v.voxelSize;
    maxRay -= v.voxelSize;
    if (oldRaylength != 0) maxRay = fminf(oldRaylength, maxRay);
    if (raylength >= maxRay) return out;  // ray misses the volume

    float raystep = v.truncdist;
    V3 p = to_voxel(v.cam + dir * raylength, v.voxelSize, half);
    while (outside(p, 1.f, v.n) && raylength < maxRay) {  // coarse search, TSDF.cu:509-514
        raylength += raystep;
        p = to_voxel(v.cam + dir * raylength, v.voxelSize, half);
    }
    // If the search ran out (Q4) the reference reads out of bounds and then never enters the
    // march (raylength >= maxRay): nothing is written either way.
    if (outside(p, 1.f, v.n)) return out;

    BrickCache cache{-1, -1, -1, kBrickMixed};
    uint8_t uni;
    float tsdf = sample_tsdf(v, p, cell_of(p, v.n), cache, uni);
    if (fabsf(tsdf) < 1.f) raystep = v.voxelSize;
    if (fabsf(tsdf) < .8f) raystep = 0.5f * v.voxelSize;
    const float halfVoxel = 0.5f * v.voxelSize;
    for (;;) {
        raylength += raystep;
        if (!(raylength <= maxRay)) break;
        p = to_voxel(v.cam + dir * raylength, v.voxelSize, half);
        if (outside(p, 2.f, v.n)) continue;
        ++out.samples;
        const Cell c = cell_of(p, v.n);
        const float next = sample_tsdf(v, p, c, cache, uni);
        out.gathered += (uni == kBrickMixed || !v.blendFromFlags) ? 1u : 0u;
        // zero crossing from behind: leave the volume's surface shell
        if (tsdf < 0 && next > 0 && trilinear_weights(v, c) > 0.f) break;
        if (fabsf(next) < 1.f) raystep = v.voxelSize;
        if (fabsf(next) < .8f) raystep = halfVoxel;
        if (tsdf > 0 && next < 0) {
            // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:542-543)
            const float tstar = raylength - raystep * tsdf / (next - tsdf);
            const V3 ps = to_voxel(v.cam + dir * tstar, v.voxelSize, half);
            if (outside(ps, 2.f, v.n)) continue;  // tsdf is NOT advanced here
            const Cell cs = cell_of(ps, v.n);
            if (trilinear_weights(v, cs) > 0.f) {
                const V3 g = gradient_at(v, cs);
                const M33 Rt = transpose(v.R);
                out.hit = true;
                out.raylength = tstar;
                out.vertex = mul(Rt, dir * tstar);
                out.normal = mul(Rt, g / norm(g));  // 0/0 -> NaN like the reference
                break;
            }
        }
        tsdf = next;

        // ---- fast-forward through deep-uniform bricks -------------------------------------------
        // The sample just taken has its low corner in a brick that, like every brick within
        // Chebyshev distance D of it, holds only k in {0, +1, -1}, so `tsdf` is a blend of k.
        // While the following samples keep all 8 corners inside those bricks each of them is again
        // a blend of k: no sign test above can fire
        // (tsdf and next share k's sign, or are both 0), and the step size is at its fixed point
        // -- blends of +-1 are +-1 or +-0.99999994, which set raystep to voxelSize or leave it;
        // blends of 0 are 0, which set it to voxelSize/2.  Hence, when raystep already has that
        // value, the reference loop does nothing for those samples but `raylength += raystep` and
        // the exit test, and that is all we replay (same float additions, same order).  Only the
        // LAST skipped sample's value is needed afterwards (it becomes `tsdf`) and is recomputed
        // exactly.
        // The skip is taken by the WAVE: only when every lane still marching is eligible, and for
        // a wave-uniform number of steps (the smallest budget), so neighbouring rays stay in
        // lock-step and ineligible iterations pay two scalar votes, nothing more.
        const bool eligible =
            uni != kBrickMixed &&
            raystep == (flag_class(uni) == kBrickAllZero ? halfVoxel : v.voxelSize);
        if (__all(eligible)) {
            const int budget = steps_inside_bricks(p, dir, raystep, v.voxelSize, v.n, cache.bx,
                                                   cache.by, cache.bz, flag_depth(uni));
            int steps = 0;  // largest count every active lane can afford (budget <= 31 here)
#pragma unroll
            for (int bit = 16; bit > 0; bit >>= 1)
                if (__all(budget >= steps + bit)) steps += bit;
            if (steps > 0) {
                int taken = 0;
                float r = raylength;
                for (; taken < steps; ++taken) {
                    const float rn = r + raystep;
                    if (!(rn <= maxRay)) break;
                    r = rn;
                }
                if (taken > 0) {
                    raylength = r;
                    out.samples += static_cast<unsigned>(taken);
                    out.skipped += static_cast<unsigned>(taken);
                    p = to_voxel(v.cam + dir * raylength, v.voxelSize, half);
                    const Cell cl = cell_of(p, v.n);
                    const float k = brick_constant(flag_class(uni));
                    tsdf = blend8(k, k, k, k, k, k, k, k, cl.fx, cl.fy, cl.fz);
                }
            }
        }
    }
    return out;
}

// wave-level reduction of the march statistics, one atomic pair per wave
__device__ __forceinline__ void add_ray_stats(unsigned long long* stats, unsigned samples,
                                              unsigned hits, unsigned gathered, unsigned skipped,
                                              int lane) {
    if (!stats) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        samples += __shfl_down(samples, off);
        hits += __shfl_down(hits, off);
        gathered += __shfl_down(gathered, off);
        skipped += __shfl_down(skipped, off);
    }
    if (lane == 0) {
        if (samples) atomicAdd(&stats[0], static_cast<unsigned long long>(samples));
        if (hits) atomicAdd(&stats[1], static_cast<unsigned long long>(hits));
        if (gathered) atomicAdd(&stats[2], static_cast<unsigned long long>(gathered));
        if (skipped) atomicAdd(&stats[3], static_cast<unsigned long long>(skipped));
    }
}

// ---- association likelihood (reference TSDF.cpp:125-156, ObjTSDF.cpp:181-201) -------------------

struct AssocModel {
    const float* tsdf;
    const float* fgProbs;  // nullptr for the background
    M33 R;                 // camera -> volume
    V3 t;
    I3 n;
    float voxelSize;
    float c1;  // -truncdist / sigma          (TSDF.cpp:151)
    float c2;  // 1 / (2 sigma)               (TSDF.cpp:154)
    float alpha;
    float c3;  // (1 - alpha) * uniPrior      (TSDF.cpp:133)
};

// One model's un-normalised association weight at one camera-frame point.
__device__ __forceinline__ float assoc_weight(const AssocModel& a, const V3& pc) {
    float s = 0.f, fg = 0.f;
    if (pc.z > 0) {
        const V3 v = to_voxel(mul(a.R, pc) + a.t, a.voxelSize, half_extent(a.n));
        if (!outside(v, 1.f, a.n)) {
            const Cell c = cell_of(v, a.n);
            s = trilinear1(a.tsdf, c, a.n);
            if (a.fgProbs) fg = trilinear1(a.fgProbs, c, a.n);
        }
    }
    // chain of single-operator OpenCV launches, kept as separate roundings:
    float L = fabsf(s);
    L = L * a.c1;
    L = expf(L);
    L = L * a.c2;
    if (a.fgProbs) L = L * fg;
    float wgt = L * a.alpha;
    wgt = wgt + a.c3;
    return (s == 0.f) ? 0.f : wgt;  // associationMask = (lookup == 0) -> weight 0 (Q6)
}

}  // namespace emf_hip
