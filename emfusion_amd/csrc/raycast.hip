// raycast.hip -- per-pixel ray march through one TSDF volume (reference kernel_raycastTSDF,
// TSDF.cu:466-573; slab test TSDF.cuh:31-63; trilinear TSDF.cuh:65-97).
//
// One lane per pixel, one wave per 8x8 pixel tile (neighbouring rays walk neighbouring voxels, so
// a wave's 8-corner gathers fall into few cache lines), 4 waves per workgroup (16x16 tile).
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
#include "common.hpp"

namespace emf_hip {
namespace {

struct RaycastArgs {
    const float* tsdf;
    const float* grads;
    const float* weights;
    const uint8_t* fg;
    Img<float> ray, vert, nrm;
    Img<uint8_t> mask;
    int w, h;
    M33 R;   // camera -> volume rotation
    V3 cam;  // camera centre in the volume frame (rel_trans_CO)
    float fx, fy, cx, cy;
    I3 n;
    float voxelSize, truncdist;
    unsigned long long* stats;
};

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

// weights as the march sees them: optionally gated by the foreground mask
__device__ __forceinline__ float trilinear_w(const RaycastArgs& a, const Cell& c) {
    const size_t sy = static_cast<size_t>(a.n.x), sz = sy * a.n.y;
    const float* p = a.weights + c.base;
    float w0 = p[0], w1 = p[1], w2 = p[sy], w3 = p[sy + 1], w4 = p[sz], w5 = p[sz + 1],
          w6 = p[sz + sy], w7 = p[sz + sy + 1];
    if (a.fg) {
        const uint8_t* m = a.fg + c.base;
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
__device__ __forceinline__ V3 gradient_at(const RaycastArgs& a, const Cell& c) {
    const size_t sy = static_cast<size_t>(a.n.x), sz = sy * a.n.y;
    float g[3][8];
    if (a.grads) {
        const float* p = a.grads + 3 * c.base;
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
        const float* p = a.tsdf + c.base;
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

__global__ __launch_bounds__(256) void k_raycast(const RaycastArgs a) {
    // wave w of the block covers the 8x8 tile at (w & 1, w >> 1) of the block's 16x16 tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    unsigned nsamples = 0, nhits = 0;

    if (x < a.w && y < a.h) {
        const V3 unproj = v3((static_cast<float>(x) - a.cx) / a.fx,
                             (static_cast<float>(y) - a.cy) / a.fy, 1.f);
        const V3 rayv = mul(a.R, unproj);
        const V3 dir = rayv / norm(rayv);
        // (volSize - 1) / 2 is INTEGER division in the reference (TSDF.cu:490, Q2)
        const V3 bb = v3(static_cast<float>((a.n.x - 1) / 2) * a.voxelSize,
                         static_cast<float>((a.n.y - 1) / 2) * a.voxelSize,
                         static_cast<float>((a.n.z - 1) / 2) * a.voxelSize);
        const V3 half = half_extent(a.n);
        float raylength = enter_step(dir, a.cam, bb);
        float maxRay = exit_step(dir, a.cam, bb);
        const float old = a.ray.row(y)[x];  // non-zero: do not search past another volume's hit
        raylength += a.voxelSize;
        maxRay -= a.voxelSize;
        if (old != 0) maxRay = fminf(old, maxRay);

        if (!(raylength >= maxRay)) {
            float raystep = a.truncdist;
            V3 v = to_voxel(a.cam + dir * raylength, a.voxelSize, half);
            while (outside(v, 1.f, a.n) && raylength < maxRay) {  // coarse search, TSDF.cu:509-514
                raylength += raystep;
                v = to_voxel(a.cam + dir * raylength, a.voxelSize, half);
            }
            // If the search ran out (Q4) the reference reads out of bounds and then never enters
            // the march (raylength >= maxRay): nothing is written either way.
            if (!outside(v, 1.f, a.n)) {
                float tsdf = trilinear1(a.tsdf, cell_of(v, a.n), a.n);
                if (fabsf(tsdf) < 1.f) raystep = a.voxelSize;
                if (fabsf(tsdf) < .8f) raystep = 0.5f * a.voxelSize;
                for (;;) {
                    raylength += raystep;
                    if (!(raylength <= maxRay)) break;
                    v = to_voxel(a.cam + dir * raylength, a.voxelSize, half);
                    if (outside(v, 2.f, a.n)) continue;
                    ++nsamples;
                    const Cell c = cell_of(v, a.n);
                    const float next = trilinear1(a.tsdf, c, a.n);
                    // zero crossing from behind: leave the volume's surface shell
                    if (tsdf < 0 && next > 0 && trilinear_w(a, c) > 0.f) break;
                    if (fabsf(next) < 1.f) raystep = a.voxelSize;
                    if (fabsf(next) < .8f) raystep = 0.5f * a.voxelSize;
                    if (tsdf > 0 && next < 0) {
                        // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:542-543)
                        const float tstar = raylength - raystep * tsdf / (next - tsdf);
                        const V3 vs = to_voxel(a.cam + dir * tstar, a.voxelSize, half);
                        if (outside(vs, 2.f, a.n)) continue;  // tsdf is NOT advanced here
                        const Cell cs = cell_of(vs, a.n);
                        if (trilinear_w(a, cs) > 0.f) {
                            const V3 g = gradient_at(a, cs);
                            const M33 Rt = transpose(a.R);
                            const V3 vert = mul(Rt, dir * tstar);
                            const V3 nrm = mul(Rt, g / norm(g));  // 0/0 -> NaN like the reference
                            a.ray.row(y)[x] = tstar;
                            float* pv = a.vert.row(y) + 3 * x;
                            float* pn = a.nrm.row(y) + 3 * x;
                            pv[0] = vert.x;
                            pv[1] = vert.y;
                            pv[2] = vert.z;
                            pn[0] = nrm.x;
                            pn[1] = nrm.y;
                            pn[2] = nrm.z;
                            a.mask.row(y)[x] = 1;
                            nhits = 1;
                            break;
                        }
                    }
                    tsdf = next;
                }
            }
        }
    }

    if (a.stats) {  // wave-level reduction, one atomic pair per wave
        unsigned s = nsamples, hcount = nhits;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s += __shfl_down(s, off);
            hcount += __shfl_down(hcount, off);
        }
        // 64 lanes x < 2^26 samples cannot overflow 32 bits per wave for any supported volume
        if (lane == 0) {
            if (s) atomicAdd(&a.stats[0], static_cast<unsigned long long>(s));
            if (hcount) atomicAdd(&a.stats[1], static_cast<unsigned long long>(hcount));
        }
    }
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" int emf_hip_raycastTSDF(const float* tsdf, const float* grads, const float* weights,
                                   const uint8_t* fgVolMask, const emf_image_t* raylengths,
                                   const emf_image_t* vertices, const emf_image_t* normals,
                                   const emf_image_t* mask, const float R_CO[9],
                                   const float t_CO[3], const float K[9], const int32_t res[3],
                                   float voxelSize, float truncdist, uint64_t* stats,
                                   emf_stream_t stream) {
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
    a.tsdf = tsdf;
    a.grads = grads;
    a.weights = weights;
    a.fg = fgVolMask;
    a.ray = img<float>(raylengths);
    a.vert = img<float>(vertices);
    a.nrm = img<float>(normals);
    a.mask = img<uint8_t>(mask);
    a.w = raylengths->width;
    a.h = raylengths->height;
    a.R = m33_from(R_CO);
    a.cam = v3_from(t_CO);
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    a.truncdist = truncdist;
    a.stats = reinterpret_cast<unsigned long long*>(stats);
    hipLaunchKernelGGL(k_raycast, dim3(ceil_div(a.w, 16), ceil_div(a.h, 16)), dim3(256), 0,
                       as_stream(stream), a);
    return launch_status("raycastTSDF");
}
