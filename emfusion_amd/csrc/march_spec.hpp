// march_spec.hpp -- the ray march of device_core.hpp with SPECULATIVE BATCHING of K samples.
//
// The reference march (TSDF.cu:523-572) is a chain of dependent steps: position -> 8 gathers ->
// blend -> decisions -> next position.  On MI355X one VGA raycast is a single round of 4800
// waves, each step costs a memory round trip, and the kernel lasts as long as its longest rays:
// a few hundred image-border rays graze the boundary between seen and unseen space and take
// 500-800 half-voxel steps while the median ray takes ~240 (scripts/raycast_probe.py tail).
//
// `raystep` changes at most a handful of times along a ray, so the next K sample positions can be
// computed ahead with the SAME float additions the reference performs (r1 = r0 + step,
// r2 = r1 + step, ...), their 8 x K corner loads issued back to back (one memory round trip per
// batch instead of per step), and the samples then evaluated strictly in order.  The first
// sample that changes `raystep`, hits, or breaks ends the batch; later speculated samples are
// discarded unread.  Every value the reference would compute is computed from the same operands
// in the same order, so results are bit-identical (tests/test_gpu_parity.py runs against this).
#pragma once

#include "device_core.hpp"

namespace emf_hip {

template <int K>
__device__ __forceinline__ RayHit march_ray_spec(const RayVolume& v, int x, int y, float fx,
                                                 float fy, float cx, float cy,
                                                 float oldRaylength) {
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
    if (outside(p, 1.f, v.n)) return out;  // Q4, see march_ray

    float tsdf = trilinear1(v.tsdf, cell_of(p, v.n), v.n);
    if (fabsf(tsdf) < 1.f) raystep = v.voxelSize;
    if (fabsf(tsdf) < .8f) raystep = 0.5f * v.voxelSize;
    const float halfVoxel = 0.5f * v.voxelSize;
    const size_t sy = static_cast<size_t>(v.n.x), sz = sy * v.n.y;

    bool done = false;
    while (!done) {
        // ---- speculate: K positions with the current step, all corner loads in flight at once --
        float rr[K];
        V3 pp[K];
        Cell cc[K];
        bool inside[K];
        float c8[K][8];
        {
            float r = raylength;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                r = r + raystep;  // the reference's `raylength += raystep`, k + 1 times
                rr[k] = r;
                pp[k] = to_voxel(v.cam + dir * r, v.voxelSize, half);
                inside[k] = !outside(pp[k], 2.f, v.n);
                // samples the march will skip (outside, or beyond maxRay) load from voxel 0: the
                // values are never looked at
                const bool use = inside[k] && r <= maxRay;
                cc[k] = cell_of(use ? pp[k] : v3(0.f, 0.f, 0.f), v.n);
                const float* q = v.tsdf + cc[k].base;
                c8[k][0] = q[0];
                c8[k][1] = q[1];
                c8[k][2] = q[sy];
                c8[k][3] = q[sy + 1];
                c8[k][4] = q[sz];
                c8[k][5] = q[sz + 1];
                c8[k][6] = q[sz + sy];
                c8[k][7] = q[sz + sy + 1];
            }
        }
        // ---- evaluate in order; stop at the first sample that changes the march state ----------
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!(rr[k] <= maxRay)) {  // `while ((raylength += raystep) <= maxRaylength)` ends
                done = true;
                break;
            }
            raylength = rr[k];
            if (!inside[k]) continue;
            ++out.samples;
            ++out.gathered;
            const Cell& c = cc[k];
            const float next = blend8(c8[k][0], c8[k][1], c8[k][2], c8[k][3], c8[k][4], c8[k][5],
                                      c8[k][6], c8[k][7], c.fx, c.fy, c.fz);
            // zero crossing from behind: leave the volume's surface shell
            if (tsdf < 0 && next > 0 && trilinear_weights(v, c) > 0.f) {
                done = true;
                break;
            }
            float ns = raystep;
            if (fabsf(next) < 1.f) ns = v.voxelSize;
            if (fabsf(next) < .8f) ns = halfVoxel;
            const bool stepChanged = ns != raystep;
            raystep = ns;
            bool advance = true;
            if (tsdf > 0 && next < 0) {
                // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:542-543)
                const float tstar = raylength - raystep * tsdf / (next - tsdf);
                const V3 ps = to_voxel(v.cam + dir * tstar, v.voxelSize, half);
                if (outside(ps, 2.f, v.n)) {
                    advance = false;  // reference `continue`: tsdf is NOT advanced
                } else {
                    const Cell cs = cell_of(ps, v.n);
                    if (trilinear_weights(v, cs) > 0.f) {
                        const V3 g = gradient_at(v, cs);
                        const M33 Rt = transpose(v.R);
                        out.hit = true;
                        out.raylength = tstar;
                        out.vertex = mul(Rt, dir * tstar);
                        out.normal = mul(Rt, g / norm(g));  // 0/0 -> NaN like the reference
                        done = true;
                        break;
                    }
                }
            }
            if (advance) tsdf = next;
            if (stepChanged) break;  // later positions were speculated with the old step: redo
        }
    }
    return out;
}

}  // namespace emf_hip
