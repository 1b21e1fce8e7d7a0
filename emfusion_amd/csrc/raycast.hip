// raycast.hip -- per-pixel ray march through ONE TSDF volume: the level-1 replacement of the
// reference's raycastTSDF wrapper (TSDF.cu:466-601).  The march itself lives in device_core.hpp
// (march_ray) and is shared with the batched all-models launch in batched.hip.
//
// One lane per pixel, one wave per 8x8 pixel tile (neighbouring rays walk neighbouring voxels, so
// a wave's corner gathers fall into few cache lines), 4 waves per workgroup (16x16 tile).
// The step sequence of the reference is reproduced exactly -- every `raylength += raystep` and
// every comparison is evaluated in the same order on the same single-precision values -- because
// the hit position depends on the whole history of step-size decisions.
//
// Differences from the reference kernel that do not change results:
//   * the trilinear blend of the weights volume is only evaluated when its value can matter
//     (tsdf < 0 && next > 0, and at a candidate hit): the reference gathers it on every step
//   * the gradient at a hit can be blended from forward differences of the TSDF on the fly
//     (grads == nullptr), which is what a gradient volume built by computeTSDFGrads stores
//   * the foreground mask of an object volume is applied inside the weight gather
//     (fg ? w : 0) instead of materialising ObjTSDF::raycastWeights every frame
//   * with brick uniformity flags, lookups inside uniform regions blend the constant instead of
//     gathering eight equal values
#include "march_wave.hpp"

namespace emf_hip {
namespace {

struct RaycastArgs {
    RayVolume vol;
    Img<float> ray, vert, nrm;
    Img<uint8_t> mask;
    int w, h;
    float fx, fy, cx, cy;
    unsigned long long* stats;
};

// One-wave workgroups: the march is latency-bound and a VGA frame is a single round of 4800
// waves, so what matters is that EVERY wave is resident from the start.  With 4-wave workgroups
// a CU holds 5 of them at 96 VGPRs (1280 slots for 1200 workgroups) and any imbalance of the
// dispatcher sends stragglers into a second round; single waves are placed SIMD by SIMD.
// WAVE: wave-scheduled march (march_wave.hpp) -- the default; the flag-aware per-lane march
// (device_core.hpp) is used when the caller supplies brick flags or the volume exceeds 4 GiB.

template <bool WAVE>
__global__ __launch_bounds__(64) void k_raycast(const RaycastArgs a) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 8 + (lane & 7);
    const int y = blockIdx.y * 8 + (lane >> 3);
    const bool valid = x < a.w && y < a.h;
    // non-zero incoming raylength: do not search past another volume's hit (TSDF.cu:496-500)
    const float old = valid ? a.ray.row(y)[x] : 0.f;
    RayHit r;
    if constexpr (WAVE) {
        // pixels without a hit are left untouched, as in the reference
        auto sink = [&](float raylength, const V3& vertex, const V3& normal) {
            a.ray.row(y)[x] = raylength;
            float* pv = a.vert.row(y) + 3 * x;
            float* pn = a.nrm.row(y) + 3 * x;
            pv[0] = vertex.x;
            pv[1] = vertex.y;
            pv[2] = vertex.z;
            pn[0] = normal.x;
            pn[1] = normal.y;
            pn[2] = normal.z;
            a.mask.row(y)[x] = 1;
        };
        const MarchCount c = march_wave(a.vol, valid, x, y, a.fx, a.fy, a.cx, a.cy, old, sink);
        add_ray_stats(a.stats, c.samples, c.hit ? 1u : 0u, c.samples, 0u, lane);
        return;
    } else {
        r.hit = false;
        r.samples = r.gathered = r.skipped = 0;
        if (valid) r = march_ray(a.vol, x, y, a.fx, a.fy, a.cx, a.cy, old);
    }
    if (valid && r.hit) {  // pixels without a hit are left untouched, as in the reference
        a.ray.row(y)[x] = r.raylength;
        float* pv = a.vert.row(y) + 3 * x;
        float* pn = a.nrm.row(y) + 3 * x;
        pv[0] = r.vertex.x;
        pv[1] = r.vertex.y;
        pv[2] = r.vertex.z;
        pn[0] = r.normal.x;
        pn[1] = r.normal.y;
        pn[2] = r.normal.z;
        a.mask.row(y)[x] = 1;
    }
    add_ray_stats(a.stats, r.samples, r.hit ? 1u : 0u, r.gathered, r.skipped, lane);
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" int emf_hip_raycastTSDF(const float* tsdf, const float* grads, const float* weights,
                                   const uint8_t* fgVolMask, const uint8_t* brickFlags,
                                   const emf_image_t* raylengths, const emf_image_t* vertices,
                                   const emf_image_t* normals, const emf_image_t* mask,
                                   const float R_CO[9], const float t_CO[3], const float K[9],
                                   const int32_t res[3], float voxelSize, float truncdist,
                                   float rcpVoxel, uint64_t* stats, emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_TRY(check_image(raylengths, 4, "raycastTSDF: raylengths"));
    EMF_TRY(check_image(vertices, 12, "raycastTSDF: vertices"));
    EMF_TRY(check_image(normals, 12, "raycastTSDF: normals"));
    EMF_TRY(check_image(mask, 1, "raycastTSDF: mask"));
    EMF_TRY(check_same_size(vertices, raylengths, "vertices", "raylengths"));
    EMF_TRY(check_same_size(normals, raylengths, "normals", "raylengths"));
    EMF_TRY(check_same_size(mask, raylengths, "mask", "raylengths"));
    EMF_REQUIRE_PTR(R_CO);
    EMF_REQUIRE_PTR(t_CO);
    EMF_REQUIRE_PTR(K);
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f) || !(truncdist > 0.f))
        return fail(EMF_E_ARG, "raycastTSDF: voxelSize %g / truncdist %g must be > 0", voxelSize,
                    truncdist);
    RaycastArgs a;
    a.vol.tsdf = tsdf;
    a.vol.grads = grads;
    a.vol.weights = weights;
    a.vol.fg = fgVolMask;
    a.vol.R = m33_from(R_CO);
    a.vol.cam = v3_from(t_CO);
    a.vol.n = i3_from(res);
    // the flag buffer holds the raw flags followed by the dilated flags; the march reads the latter
    a.vol.bricks = brickFlags ? brickFlags + brick_count(a.vol.n) : nullptr;
    a.vol.blendFromFlags = false;
    a.vol.voxelSize = voxelSize;
    a.vol.truncdist = truncdist;
    a.vol.rcpVoxel = usable_reciprocal(rcpVoxel, t_CO);
    a.ray = img<float>(raylengths);
    a.vert = img<float>(vertices);
    a.nrm = img<float>(normals);
    a.mask = img<uint8_t>(mask);
    a.w = raylengths->width;
    a.h = raylengths->height;
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.stats = reinterpret_cast<unsigned long long*>(stats);
    if (brickFlags || !fits_offsets32(res))
        hipLaunchKernelGGL(k_raycast<false>, dim3(ceil_div(a.w, 8), ceil_div(a.h, 8)), dim3(64), 0,
                           as_stream(stream), a);
    else
        hipLaunchKernelGGL(k_raycast<true>, dim3(ceil_div(a.w, 8), ceil_div(a.h, 8)), dim3(64), 0,
                           as_stream(stream), a);
    return launch_status("raycastTSDF");
}
