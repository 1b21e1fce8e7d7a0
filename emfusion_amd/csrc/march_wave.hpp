// march_wave.hpp -- the ray march of reference kernel_raycastTSDF (TSDF.cu:466-573) scheduled per
// WAVE instead of per lane.
//
// Why: one VGA raycast is a single round of 4800 waves and the march is a chain of dependent steps
// (~200 instructions + one gather round trip each), so the kernel lasts as long as its longest
// rays.  On the bench scene the median ray takes ~240 steps but a few hundred image-border rays
// graze the seen/unseen boundary at half-voxel steps for 500-800 steps, and object volumes leave
// most waves with a handful of rays that actually cross the box (scripts/raycast_probe.py tail).
// A lane that marches alone keeps its wave alive while 63 lanes idle.
//
// MEASURED OUTCOME: the wave-level loop below (ordinary mode only, one
// scalar ballot per iteration) is the fastest variant, 0.53 ms for the 512^3 bench background
// against 0.57-0.65 ms for per-lane loops.  The cooperative mode is exact and 20x faster on long
// plain runs, but the rays that form the tail of this workload are NOT plain -- they graze the
// seen/unseen boundary, their 0/1 blends keep crossing the 0.8 / 1.0 thresholds and flip the step
// size every few samples -- so it fires on 0.1 % of the samples while its code costs 30 % in
// registers and scheduling.  It is therefore compiled out by default (EMF_COOP_MAX_RAYS = 0) and
// kept for scenes with long uniform stretches.
//
// Two modes, chosen per wave iteration by a ballot:
//   * many rays active: every active lane takes one ordinary step (ray_step = one iteration of
//     the reference loop);
//   * at most kCoopMaxRays rays active: the wave works for ONE of them.  Lane j evaluates the
//     sample j + 1 steps ahead -- its raylength by the same j + 1 sequential float additions the
//     reference would perform -- gathers and blends it; a prefix over the ballot of in-volume
//     samples gives each lane the TSDF value that would precede it; the first sample that could
//     change the march state (step size change, sign change, end of range) is found with a
//     ballot + ffs.  All samples before it are "plain": the reference loop would only advance
//     `raylength`, count them and carry the last value in `tsdf`, which is what the leader lane
//     does in one go.  The event sample itself is then taken through ray_step on the leader, so
//     every state-changing decision is made by exactly the code of the ordinary path.
// Results are bit-identical to the per-lane march (tests/test_gpu_parity.py).
#pragma once

#include "device_core.hpp"

namespace emf_hip {

#ifndef EMF_COOP_MAX_RAYS
#define EMF_COOP_MAX_RAYS 0
#endif
constexpr int kCoopMaxRays = EMF_COOP_MAX_RAYS;

#ifndef EMF_COOP_MIN_RUN
#define EMF_COOP_MIN_RUN 24
#endif
// Cooperative mode advances ONE ray by up to 64 samples for the price of a few ordinary steps,
// whereas an ordinary iteration advances EVERY active ray by one: it pays only on long runs of
// plain samples.  A ray qualifies once its last kCoopMinRun samples were plain (image-border rays
// grazing seen/unseen space, rays crossing unseen or free space of an object volume); rays in the
// near-surface band, where the step size changes and the crossing is imminent, never do.
constexpr int kCoopMinRun = EMF_COOP_MIN_RUN;

struct RayState {
    V3 dir;
    float raylength, maxRay, raystep, tsdf;
    int plainRun;  // consecutive samples that changed nothing but raylength / tsdf
    bool active;
};

// Everything before the main loop of the reference kernel (TSDF.cu:476-521).
__device__ __forceinline__ void ray_setup(const RayVolume& v, int x, int y, float fx, float fy,
                                          float cx, float cy, float oldRaylength, RayState& r) {
    r.active = false;
    const V3 unproj = v3((static_cast<float>(x) - cx) / fx, (static_cast<float>(y) - cy) / fy, 1.f);
    const V3 rayv = mul(v.R, unproj);
    r.dir = rayv / norm(rayv);
    // (volSize - 1) / 2 is INTEGER division in the reference (TSDF.cu:490, Q2)
    const V3 bb = v3(static_cast<float>((v.n.x - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.y - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.z - 1) / 2) * v.voxelSize);
    const V3 half = half_extent(v.n);
    r.raylength = enter_step(r.dir, v.cam, bb);
    r.maxRay = exit_step(r.dir, v.cam, bb);
    r.raylength += v.voxelSize;
    r.maxRay -= v.voxelSize;
    if (oldRaylength != 0) r.maxRay = fminf(oldRaylength, r.maxRay);
    r.raystep = v.truncdist;
    r.tsdf = 0.f;
    r.plainRun = 0;
    if (r.raylength >= r.maxRay) return;  // ray misses the volume
    V3 p = to_voxel(v.cam + r.dir * r.raylength, v.voxelSize, half);
    while (outside(p, 1.f, v.n) && r.raylength < r.maxRay) {  // coarse search, TSDF.cu:509-514
        r.raylength += r.raystep;
        p = to_voxel(v.cam + r.dir * r.raylength, v.voxelSize, half);
    }
    // If the search ran out (Q4) the reference reads out of bounds and then never enters the
    // march (raylength >= maxRay): nothing is written either way.
    if (outside(p, 1.f, v.n)) return;
    r.tsdf = trilinear1(v.tsdf, cell_of(p, v.n), v.n);
    if (fabsf(r.tsdf) < 1.f) r.raystep = v.voxelSize;
    if (fabsf(r.tsdf) < .8f) r.raystep = 0.5f * v.voxelSize;
    r.active = true;
}

#ifdef EMF_X_FASTDIV
__device__ __forceinline__ float xdiv(float x, float d, float rcp) {
    const float q = x * rcp;
    return __builtin_fmaf(__builtin_fmaf(-q, d, x), rcp, q);
}
__device__ __forceinline__ V3 to_voxel_x(const V3& p, float d, float rcp, const V3& half) {
    return v3(xdiv(p.x, d, rcp) + half.x, xdiv(p.y, d, rcp) + half.y, xdiv(p.z, d, rcp) + half.z);
}
#else
__device__ __forceinline__ V3 to_voxel_x(const V3& p, float d, float, const V3& half) {
    return to_voxel(p, d, half);
}
#endif
#ifdef EMF_X_OFF32
__device__ __forceinline__ float trilinear1_x(const float* __restrict__ vol, const V3& idx, const I3& n) {
    const int lx = static_cast<int>(idx.x), ly = static_cast<int>(idx.y), lz = static_cast<int>(idx.z);
    const float fx = idx.x - static_cast<float>(lx), fy = idx.y - static_cast<float>(ly),
                fz = idx.z - static_cast<float>(lz);
    const unsigned sy = 4u * static_cast<unsigned>(n.x), sz = sy * static_cast<unsigned>(n.y);
    const unsigned o = (static_cast<unsigned>(lz) * static_cast<unsigned>(n.y) + static_cast<unsigned>(ly)) * sy + 4u * static_cast<unsigned>(lx);
    const char* b = reinterpret_cast<const char*>(vol);
    const float c0 = *reinterpret_cast<const float*>(b + o), c1 = *reinterpret_cast<const float*>(b + o + 4u);
    const unsigned o1 = o + sy, o2 = o + sz, o3 = o2 + sy;
    const float c2 = *reinterpret_cast<const float*>(b + o1), c3 = *reinterpret_cast<const float*>(b + o1 + 4u);
    const float c4 = *reinterpret_cast<const float*>(b + o2), c5 = *reinterpret_cast<const float*>(b + o2 + 4u);
    const float c6 = *reinterpret_cast<const float*>(b + o3), c7 = *reinterpret_cast<const float*>(b + o3 + 4u);
    return blend8(c0, c1, c2, c3, c4, c5, c6, c7, fx, fy, fz);
}
#endif

// One iteration of `while ((raylength += raystep) <= maxRaylength)` (TSDF.cu:523-572) for the
// calling lane.  Clears r.active when the march ends (range exhausted, back-side crossing, hit).
__device__ __forceinline__ void ray_step(const RayVolume& v, RayState& r, RayHit& out, float rcp = 0.f) {
    const V3 half = half_extent(v.n);
    r.raylength += r.raystep;
    if (!(r.raylength <= r.maxRay)) {
        r.active = false;
        return;
    }
    const V3 p = to_voxel_x(v.cam + r.dir * r.raylength, v.voxelSize, rcp, half);
    if (outside(p, 2.f, v.n)) {
        ++r.plainRun;
        return;
    }
    ++out.samples;
#ifdef EMF_X_OFF32
    const float next = trilinear1_x(v.tsdf, p, v.n);
#else
    const float next = trilinear1(v.tsdf, cell_of(p, v.n), v.n);
#endif
    // zero crossing from behind: leave the volume's surface shell
    if (r.tsdf < 0 && next > 0 && trilinear_weights(v, cell_of(p, v.n)) > 0.f) {
        r.active = false;
        return;
    }
    const float stepBefore = r.raystep;
    if (fabsf(next) < 1.f) r.raystep = v.voxelSize;
    if (fabsf(next) < .8f) r.raystep = 0.5f * v.voxelSize;
    // same classification as coop_advance: plain = nothing but raylength / tsdf moves
    const bool plain = r.raystep == stepBefore && !(r.tsdf < 0 && next > 0) && !(r.tsdf > 0 && next < 0);
    r.plainRun = plain ? r.plainRun + 1 : 0;
    if (r.tsdf > 0 && next < 0) {
        // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:542-543)
        const float tstar = r.raylength - r.raystep * r.tsdf / (next - r.tsdf);
        const V3 ps = to_voxel(v.cam + r.dir * tstar, v.voxelSize, half);
        if (outside(ps, 2.f, v.n)) return;  // reference `continue`: tsdf is NOT advanced
        const Cell cs = cell_of(ps, v.n);
        if (trilinear_weights(v, cs) > 0.f) {
            const V3 g = gradient_at(v, cs);
            const M33 Rt = transpose(v.R);
            out.hit = true;
            out.raylength = tstar;
            out.vertex = mul(Rt, r.dir * tstar);
            out.normal = mul(Rt, g / norm(g));  // 0/0 -> NaN like the reference
            r.active = false;
            return;
        }
    }
    r.tsdf = next;
}

// The whole wave works on the ray of lane `leader`: returns through `r` / `out` of that lane.
// Must be called by all 64 lanes (wave-uniform control flow).
__device__ __forceinline__ void coop_advance(const RayVolume& v, int leader, int lane, RayState& r,
                                             RayHit& out) {
    const V3 half = half_extent(v.n);
    const float r0 = __shfl(r.raylength, leader), mx = __shfl(r.maxRay, leader),
                st = __shfl(r.raystep, leader), ts = __shfl(r.tsdf, leader);
    const V3 d = v3(__shfl(r.dir.x, leader), __shfl(r.dir.y, leader), __shfl(r.dir.z, leader));
    // sample j + 1 of the leader's ray: raylength after j + 1 executions of `raylength += raystep`
    float rj = r0;
    for (int i = 0; i <= lane; ++i) rj = rj + st;
    const bool inRange = rj <= mx;
    const V3 pj = to_voxel(v.cam + d * rj, v.voxelSize, half);
    const bool ins = inRange && !outside(pj, 2.f, v.n);
    float nx = 0.f;
    if (ins) nx = trilinear1(v.tsdf, cell_of(pj, v.n), v.n);
    const unsigned long long M = __ballot(ins);
    // value the reference's `tsdf` would hold when it reaches this sample, provided every earlier
    // sample of the batch is plain: the blend of the nearest earlier in-volume sample, else `ts`
    const unsigned long long below = M & ((1ull << lane) - 1ull);
    const int prevLane = below ? 63 - __clzll(static_cast<long long>(below)) : 0;
    const float prevVal = __shfl(nx, prevLane);
    const float prev = below ? prevVal : ts;
    float ns = st;
    if (fabsf(nx) < 1.f) ns = v.voxelSize;
    if (fabsf(nx) < .8f) ns = 0.5f * v.voxelSize;
    // anything but "advance raylength / count / carry the value" is an event
    const bool event = !inRange || (ins && (ns != st || (prev < 0 && nx > 0) || (prev > 0 && nx < 0)));
    const unsigned long long E = __ballot(event);
    const int e = E ? __ffsll(static_cast<long long>(E)) - 1 : 64;  // samples [0, e) are plain
    const unsigned long long plain = e >= 64 ? ~0ull : ((1ull << e) - 1ull);
    const unsigned long long Mp = M & plain;
    const float rLast = __shfl(rj, e > 0 ? e - 1 : 0);
    const float tLast = __shfl(nx, Mp ? 63 - __clzll(static_cast<long long>(Mp)) : 0);
    if (lane == leader) {
        if (e > 0) {
            r.raylength = rLast;
            r.plainRun += e;
            const unsigned cnt = static_cast<unsigned>(__popcll(Mp));
            out.samples += cnt;
            out.gathered += cnt;
            out.skipped += cnt;  // statistic: samples consumed cooperatively
            if (Mp) r.tsdf = tLast;
        }
#ifdef EMF_COOP_DEBUG_CALLS
        out.gathered += 1000000u;  // diagnostic build only: count cooperative calls
#endif
        // the sample that may change the march state goes through the ordinary step
        if (e < 64) ray_step(v, r, out);
    }
}

// March the ray of pixel (x, y); `valid` = the pixel exists.  All 64 lanes of the wave call this.
__device__ __forceinline__ RayHit march_wave(const RayVolume& v, bool valid, int x, int y, float fx,
                                             float fy, float cx, float cy, float oldRaylength,
                                             int lane) {
    RayHit out;
    out.hit = false;
    out.samples = 0;
    out.gathered = 0;
    out.skipped = 0;
    out.raylength = 0.f;
    out.vertex = v3(0.f, 0.f, 0.f);
    out.normal = v3(0.f, 0.f, 0.f);
    RayState r;
    r.dir = v3(0.f, 0.f, 1.f);
    r.raylength = r.maxRay = r.raystep = r.tsdf = 0.f;
    r.plainRun = 0;
    r.active = false;
    if (valid) ray_setup(v, x, y, fx, fy, cx, cy, oldRaylength, r);
    const float rcp = 1.0f / v.voxelSize;
    for (;;) {
        const unsigned long long act = __ballot(r.active);
        if (act == 0) break;
        // rays that have just shown a long plain run are worth the whole wave's attention
        const unsigned long long cand =
            __popcll(act) <= kCoopMaxRays ? __ballot(r.active && r.plainRun >= kCoopMinRun) : 0ull;
        if (cand == 0) {
            if (r.active) ray_step(v, r, out, rcp);
        } else {
            coop_advance(v, __ffsll(static_cast<long long>(cand)) - 1, lane, r, out);
        }
    }
    out.gathered = out.samples;
    return out;
}

}  // namespace emf_hip
