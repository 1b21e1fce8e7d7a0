// march_wave.hpp -- the ray march of reference kernel_raycastTSDF (TSDF.cu:466-573), scheduled per
// wave and trimmed to the instructions the arithmetic needs.
//
// What bounds the raycast on MI355X (scripts/raycast_timeline.py): instructions
// per step, twice over.  A VGA raycast of the 512^3 background is 4800 waves, all resident at once.
// While they are, ~5 waves share each SIMD and a step costs ~1.2 us: the VALU is saturated.  Then
// the waves with the longest rays run on alone (image-border rays grazing the seen / unseen
// boundary: 400-1100 half-voxel steps against a median of 240) -- a lone wave issues in order, so
// a step costs (instructions x 4-5 cycles) + one memory round trip, ~0.8 us, and that tail is up
// to half the kernel.  Either way time is proportional to the instruction count of one step.
//
// The step below does the reference's arithmetic, operation for operation, in fewer instructions:
//   * loop control per wave: one ballot per iteration instead of per-lane loop bookkeeping;
//   * p / voxelSize: a constant divisor.  q = x * r, q' = fma(fma(-q, d, x), r, q) with
//     r = 1 / d equals the IEEE quotient for every float x with 1e-30 <= |x| <= 1e30 -- not
//     argued but CHECKED: emf_hip_voxelReciprocal runs every mantissa of the binade [1, 2) (both signs) and of the lowest binade of the range through both
//     forms for the given d (which decides all binades of the range: both forms commute with 2^k
//     scaling there, abi_common.hip) and hands out r only if none differs.  Outside that range: |x| < 1e-30 gives
//     |q| < 1e-24 either way and the following "+ (N - 1) / 2" absorbs it (N = 1: the sample is
//     outside the volume either way); |x| > 1e30 cannot occur because the host refuses r for poses
//     with |t| > 1e15 and the march keeps |raylength| below that.  3 instructions instead of 11;
//   * the 8 corners are read with GLOBAL loads at a 32-bit byte offset from a scalar base (the
//     model table hands out generic pointers, for which the compiler emits flat loads and 64-bit
//     address arithmetic); volumes above 4 GiB take the 64-bit march_ray instead (host decision);
//   * the bounds test is one predicate, not five nested exec-mask branches.
// Variants that were measured and dropped (kept in git history, numbers in DESIGN.md 5.3):
// speculative batching of K samples (+8 % at K = 2, register-bound beyond), a cooperative mode in
// which the 64 lanes evaluate 64 consecutive samples of one ray (exact, but the tail rays change
// their step size every few samples, so it almost never applies), brick-flag fast-forward.
#pragma once

#include "device_core.hpp"

namespace emf_hip {

typedef const __attribute__((address_space(1))) char* gchar_p;
typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef const __attribute__((address_space(1))) uint8_t* gbyte_p;
typedef float pair_f __attribute__((ext_vector_type(2), aligned(4)));  // two x-adjacent voxels
typedef const __attribute__((address_space(1))) pair_f* gpair_p;

// global (not flat) load of base[byteOff / 4]; base is wave-uniform, the offset 32 bits
__device__ __forceinline__ float gload(const float* base, unsigned byteOff) {
    return *(gfloat_p)((gchar_p)base + byteOff);
}
__device__ __forceinline__ uint8_t gload(const uint8_t* base, unsigned off) {
    return *(gbyte_p)((gchar_p)base + off);
}
// base[byteOff / 4] and its x neighbour with one 8-byte load (dword alignment suffices on gfx950)
__device__ __forceinline__ pair_f gload2(const float* base, unsigned byteOff) {
    return *(gpair_p)((gchar_p)base + byteOff);
}

// A wave-uniform pointer pinned into scalar registers.  The volume pointers come out of the device model table; the
// compiler knows they are uniform but keeps them in VGPRs and re-reads them with v_readfirstlane (+ s_nop 4) in front of
// every gather of every iteration -- and, with one SGPR pair recycled for all four row bases, splits the four gathers
// of a sample around an s_waitcnt: two memory round trips per iteration instead of one (round 5, ISA of march_quad).
__device__ __forceinline__ const float* scalar_ptr(const float* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u >> 32));
    return reinterpret_cast<const float*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// x / voxelSize: IEEE division, or its checked 3-instruction equal (rcp != 0, see the header)
__device__ __forceinline__ float div_voxel(float x, float d, float rcp) {
    if (rcp != 0.f) {  // wave-uniform
        const float q = x * rcp;
        return __builtin_fmaf(__builtin_fmaf(-q, d, x), rcp, q);
    }
    return x / d;
}
__device__ __forceinline__ V3 to_voxel(const V3& p, const RayVolume& v, const V3& half) {
    V3 q;
    if (v.rcpVoxel != 0.f) {  // wave-uniform: one branch for the three coordinates
        q = v3(div_voxel(p.x, v.voxelSize, v.rcpVoxel), div_voxel(p.y, v.voxelSize, v.rcpVoxel),
               div_voxel(p.z, v.voxelSize, v.rcpVoxel));
    } else {
        q = p / v.voxelSize;
    }
    return q + half;
}

// outside(): the same six comparisons, evaluated without short-circuit branches
__device__ __forceinline__ bool outside_flat(const V3& p, float pad, const V3& nf) {
    return (p.x < 0.f) | (p.x + pad >= nf.x) | (p.y < 0.f) | (p.y + pad >= nf.y) | (p.z < 0.f) |
           (p.z + pad >= nf.z);
}

// corner addressing with a 32-bit byte offset (volumes up to 4 GiB per channel)
struct Cell32 {
    unsigned off;  // 4 * (((lz * Ny) + ly) * Nx + lx)
    float fx, fy, fz;
};
__device__ __forceinline__ Cell32 cell32_of(const V3& idx, const I3& n) {
    const int lx = static_cast<int>(idx.x), ly = static_cast<int>(idx.y),
              lz = static_cast<int>(idx.z);
    Cell32 c;
    c.off = ((static_cast<unsigned>(lz) * static_cast<unsigned>(n.y) + static_cast<unsigned>(ly)) *
                 static_cast<unsigned>(n.x) +
             static_cast<unsigned>(lx)) *
            4u;
    c.fx = idx.x - static_cast<float>(lx);
    c.fy = idx.y - static_cast<float>(ly);
    c.fz = idx.z - static_cast<float>(lz);
    return c;
}
__device__ __forceinline__ Cell widen(const Cell32& c) {
    return Cell{static_cast<size_t>(c.off >> 2), c.fx, c.fy, c.fz};
}

__device__ __forceinline__ float trilinear1_g(const float* vol, const Cell32& c, const I3& n) {
    const unsigned sy = 4u * static_cast<unsigned>(n.x), sz = sy * static_cast<unsigned>(n.y);
    const pair_f a = gload2(vol, c.off), b = gload2(vol, c.off + sy), d = gload2(vol, c.off + sz),
                 e = gload2(vol, c.off + sz + sy);
    return blend8(a.x, a.y, b.x, b.y, d.x, d.y, e.x, e.y, c.fx, c.fy, c.fz);
}

// weights as the march sees them: optionally gated by the foreground mask (ObjTSDF.cpp:209-210)
__device__ __forceinline__ float trilinear_weights_g(const RayVolume& v, const Cell32& c) {
    const unsigned sy = 4u * static_cast<unsigned>(v.n.x), sz = sy * static_cast<unsigned>(v.n.y);
    const unsigned o[4] = {c.off, c.off + sy, c.off + sz, c.off + sz + sy};
    float w[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const pair_f p = gload2(v.weights, o[k]);
        w[2 * k] = p.x;
        w[2 * k + 1] = p.y;
    }
    if (v.fg) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[2 * k] = gload(v.fg, o[k] >> 2) ? w[2 * k] : 0.f;
            w[2 * k + 1] = gload(v.fg, (o[k] >> 2) + 1u) ? w[2 * k + 1] : 0.f;
        }
    }
    return blend8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], c.fx, c.fy, c.fz);
}

struct MarchCount {
    bool hit;
    unsigned samples;  // main-loop samples taken (byte-model statistic)
    unsigned gathered; // march_quad: samples gathered incl. the speculative ones that were dropped (= samples otherwise)
#ifdef EMF_MARCH_STAMP
    // attribution build only (scripts/raycast_attribution.py): shader clocks of the wave's loop iterations, split
    // at the moment the four corner gathers have been issued and at the moment they have all returned
    // (s_memtime; wave-uniform).  hist: iterations by the length of that wait.
    unsigned long long ckIssue, ckWait, ckRest;
    unsigned iters, hist[6];
    unsigned long long lateIssue, lateWait, lateRest;  // the same sums over iterations 256.. only (the emptying chip)
#endif
};
#ifdef EMF_MARCH_STAMP
__device__ __forceinline__ unsigned long long march_clock() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return c;
}
#endif

struct RayState {
    V3 dir;
    float raylength, maxRay, raystep, tsdf;
    bool active;
};

// Everything before the main loop of the reference kernel (TSDF.cu:476-521).
__device__ __forceinline__ void ray_setup(const RayVolume& v, const V3& half, const V3& nf, int x,
                                          int y, float fx, float fy, float cx, float cy,
                                          float oldRaylength, float cut, RayState& r) {
    r.active = false;
    const V3 unproj = v3((static_cast<float>(x) - cx) / fx, (static_cast<float>(y) - cy) / fy, 1.f);
    const V3 rayv = mul(v.R, unproj);
    r.dir = rayv / norm(rayv);
    // (volSize - 1) / 2 is INTEGER division in the reference (TSDF.cu:490, Q2)
    const V3 bb = v3(static_cast<float>((v.n.x - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.y - 1) / 2) * v.voxelSize,
                     static_cast<float>((v.n.z - 1) / 2) * v.voxelSize);
    r.raylength = enter_step(r.dir, v.cam, bb);
    r.maxRay = exit_step(r.dir, v.cam, bb);
    r.raylength += v.voxelSize;
    r.maxRay -= v.voxelSize;
    if (oldRaylength != 0) r.maxRay = fminf(oldRaylength, r.maxRay);
    r.raystep = v.truncdist;
    r.tsdf = 0.f;
    if (r.raylength >= r.maxRay) return;  // ray misses the volume
    V3 p = to_voxel(v.cam + r.dir * r.raylength, v, half);
    while (outside_flat(p, 1.f, nf) && r.raylength < r.maxRay) {  // coarse search, TSDF.cu:509-514
        r.raylength += r.raystep;
        p = to_voxel(v.cam + r.dir * r.raylength, v, half);
    }
    // If the search ran out (Q4) the reference reads out of bounds and then never enters the
    // march (raylength >= maxRay): nothing is written either way.
    if (outside_flat(p, 1.f, nf)) return;
    r.tsdf = trilinear1_g(v.tsdf, cell32_of(p, v.n), v.n);
    if (fabsf(r.tsdf) < 1.f) r.raystep = v.voxelSize;
    if (fabsf(r.tsdf) < .8f) r.raystep = 0.5f * v.voxelSize;
    r.active = true;
    // Far bound (k_far_bounds in batched.hip): past `cut` no sample can complete a hit, PROVIDED the
    // positive sample a hit compares with is the one taken an iteration earlier.  That holds once the
    // march is inside [0, N - 2)^3 -- a convex region the ray cannot re-enter, in which every iteration
    // samples; if this first sample lies in the volume's outer shell the loop may skip iterations
    // (TSDF.cu:525-528) while still comparing with it, arbitrarily far back: such a ray keeps its range.
    if (!outside_flat(p, 2.f, nf)) r.maxRay = fminf(r.maxRay, cut);
}

// ---- the march: one divergent loop per lane ---------------------------------------------
// scripts/probes/march_probe.hip: a wave that is alone on its SIMD issues ONE instruction every 8
// clocks, whatever the instruction (VALU, SALU, branch, s_nop, s_waitcnt) and however independent --
// gfx950 needs four waves per SIMD to fill the VALU.  The march is 4800 waves of ~250..600 strictly
// sequential steps, about five per SIMD at the start and ever fewer as the short rays finish: its
// duration is (steps of the longest rays) x (instructions per step) x 8 clocks, plus the memory round
// trip where the lines are cold.  So the step below is written for the instruction COUNT of its common
// path (a sample with no sign change against the previous one):
//   * the loop is the lane's own (exec-masked) loop: no `active` flag to test, set and vote on;
//   * the four x-pairs of the 8 corners are fetched at ONE 32-bit offset from four wave-uniform row
//     bases (y, y + 1 at z and z + 1) instead of four offsets from one base;
//   * offset = lz * strideZ + ly * strideY + 4 lx with two 24-bit multiply-adds (full rate);
//   * fraction = v_fract_f32 (x - floor(x): the bits of x - float(int(x)) for x >= 0);
//   * the two crossing tests (TSDF.cu:533, 541) hide behind one integer test "the sign bits of the
//     previous and the new sample differ"; only then the exact tests, the weights and a possible hit run.
// Arithmetic and its order are the reference's (TSDF.cu:523-572): same bits.
// Measured and dropped (round 5): MORE waves per SIMD.  The hit's gradient (32 loads in flight, once per ray) sets the
// kernel's register count: 86 VGPRs = 5 waves per SIMD with it, 71 = 7 with the three components taken one after the
// other in a rolled loop, 64 = 8 under a cap (4 spilled VGPRs in the rare paths).  k_raycast: 0.383 / 0.419 / 0.420 ms
// alone, 0.416 / 0.452 / 0.463 beside the sweep -- the waves a CU holds beyond 20 only take L1 and gather-path share from
// each other (LDS-capped 4 / 3 / 2 waves lost too, round 1: five is the optimum from both sides).
// Measured and dropped: touching the line the ray will want 8 / 16 half-voxel steps ahead with a fifth
// load per step (a lone wave on cold lines: 1302 -> 1224 clk / step; the full image: +19 %, the bench
// -5 %: loads return in order, so the march waits for the prefetch of the step before anyway).
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }

//   * the loop has ONE exit, at its top: a lane that is done (back-side crossing, hit) sets its range
//     to -inf and leaves at the next range test -- breaks out of nested branches cost a dozen scalar
//     instructions of exec-mask bookkeeping per iteration, in every iteration;
//   * bounds (TSDF.cu:525-528): lx = floor(x) as an integer, and 0 <= lx < Nx - 3 (one unsigned
//     comparison per axis) means 0 <= x and x + 2 < Nx whatever the rounding of x + 2; only a sample
//     in the outermost cells takes the six float comparisons;
//   * x / voxelSize by multiplication or by division is decided once per wave, not once per sample.
template <bool RCP, class Sink>
__device__ __forceinline__ void march_lane(const RayVolume& v, const V3& half, const V3& nf, RayState& r,
                                           MarchCount& out, Sink& sink) {
    const unsigned sy = 4u * static_cast<unsigned>(v.n.x), sz = sy * static_cast<unsigned>(v.n.y);
    const float* const row00 = scalar_ptr(v.tsdf);
    const float* const row01 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sy));
    const float* const row10 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sz));
    const float* const row11 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sz + sy));
    const float vs = v.voxelSize, hvs = 0.5f * v.voxelSize, rcp = v.rcpVoxel;
    // cells [0, lim) are inside whatever the rounding (lim may be <= 0 for tiny volumes: nothing is)
    const unsigned limx = static_cast<unsigned>(max(v.n.x - 3, 0)), limy = static_cast<unsigned>(max(v.n.y - 3, 0)),
                   limz = static_cast<unsigned>(max(v.n.z - 3, 0));
    const V3 dir = r.dir;
    float tmax = r.maxRay;
    float t = r.raylength, step = r.raystep, tsdf = r.tsdf;
    unsigned samples = 0;
#ifdef EMF_MARCH_STAMP
    unsigned long long ckIssue = 0, ckWait = 0, ckRest = 0, ckTop = march_clock();
    unsigned long long lateIssue = 0, lateWait = 0, lateRest = 0;
    unsigned iters = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
#endif
    for (;;) {
        t += step;
        if (!(t <= tmax)) break;
        const V3 pm = v.cam + dir * t;
        V3 p;
        if (RCP) {
            p = v3(div_voxel(pm.x, vs, rcp), div_voxel(pm.y, vs, rcp), div_voxel(pm.z, vs, rcp)) + half;
        } else {
            p = pm / vs + half;
        }
        const int lx = static_cast<int>(floorf(p.x)), ly = static_cast<int>(floorf(p.y)),
                  lz = static_cast<int>(floorf(p.z));
        bool inside = (static_cast<unsigned>(lx) < limx) & (static_cast<unsigned>(ly) < limy) &
                      (static_cast<unsigned>(lz) < limz);
        if (!inside) inside = !outside_flat(p, 2.f, nf);  // outermost cells, outside, NaN: the reference's test
        if (inside) {
            ++samples;
            const float fx = __builtin_amdgcn_fractf(p.x), fy = __builtin_amdgcn_fractf(p.y),
                        fz = __builtin_amdgcn_fractf(p.z);
            const unsigned off = mad24(static_cast<unsigned>(lz), sz,
                                       mad24(static_cast<unsigned>(ly), sy, static_cast<unsigned>(lx) << 2));
            const pair_f a = gload2(row00, off), b = gload2(row01, off), d = gload2(row10, off),
                         e = gload2(row11, off);
#ifdef EMF_MARCH_STAMP
            const unsigned long long ckSent = march_clock();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long ckBack = march_clock();
            {
                const unsigned wait = static_cast<unsigned>(ckBack - ckSent);
                ckIssue += ckSent - ckTop;
                ckWait += wait;
                if (iters >= 256u) { lateIssue += ckSent - ckTop; lateWait += wait; }
                h0 += wait < 200u; h1 += wait >= 200u && wait < 400u; h2 += wait >= 400u && wait < 700u;
                h3 += wait >= 700u && wait < 1200u; h4 += wait >= 1200u && wait < 2500u; h5 += wait >= 2500u;
                ++iters;
            }
#endif
            const float next = blend8(a.x, a.y, b.x, b.y, d.x, d.y, e.x, e.y, fx, fy, fz);
            bool advance = true;  // reference: `tsdf = next_tsdf` at the end of the iteration
            if ((__float_as_int(tsdf) ^ __float_as_int(next)) < 0) {
                // the signs differ (counting -0 as negative): the reference's two crossing tests
                const Cell32 c{off, fx, fy, fz};
                if (tsdf < 0 && next > 0 && trilinear_weights_g(v, c) > 0.f) {
                    tmax = -__builtin_inff();  // crossing from behind: `break`
                    advance = false;
                } else if (tsdf > 0 && next < 0) {
                    // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:537-543)
                    float st = step;
                    if (fabsf(next) < 1.f) st = vs;
                    if (fabsf(next) < .8f) st = hvs;
                    const float tstar = t - st * tsdf / (next - tsdf);
                    const V3 ps = to_voxel(v.cam + dir * tstar, v, half);
                    if (outside_flat(ps, 2.f, nf)) {
                        advance = false;  // reference `continue`: the step is updated, tsdf is NOT
                    } else {
                        const Cell32 cs = cell32_of(ps, v.n);
                        if (trilinear_weights_g(v, cs) > 0.f) {
                            const V3 g = gradient_at(v, widen(cs));
                            const M33 Rt = transpose(v.R);
                            out.hit = true;
                            sink(tstar, mul(Rt, dir * tstar), mul(Rt, g / norm(g)));  // 0/0 -> NaN like the reference
                            tmax = -__builtin_inff();  // `break`
                        }
                    }
                }
            }
            if (fabsf(next) < 1.f) step = vs;  // (harmless for a lane that is done)
            if (fabsf(next) < .8f) step = hvs;
            if (advance) tsdf = next;
#ifdef EMF_MARCH_STAMP
            {
                const unsigned long long ckEnd = march_clock();
                ckRest += ckEnd - ckBack;
                if (iters > 256u) lateRest += ckEnd - ckBack;
                ckTop = ckEnd;
            }
#endif
        }
    }
    out.samples = out.gathered = samples;
#ifdef EMF_MARCH_STAMP
    out.ckIssue = ckIssue; out.ckWait = ckWait; out.ckRest = ckRest; out.iters = iters;
    out.lateIssue = lateIssue; out.lateWait = lateWait; out.lateRest = lateRest;
    out.hist[0] = h0; out.hist[1] = h1; out.hist[2] = h2; out.hist[3] = h3; out.hist[4] = h4; out.hist[5] = h5;
#endif
}

// ---- the march, third form: FOUR LANES PER RAY (round 5) ----------------------------------------------
// march_lane's duration is the serial chain of its longest rays: 250-650 iterations at ~0.6 us each
// (58-65 % in-order issue at one instruction per 8-10 clocks, 35-42 % queued L1 / L2 service for the four
// corner gathers), with two waves per SIMD and the VALU 80 % idle.  Nothing that hides
// latency INSIDE a lane shortens that chain; this form trades chain length for width instead.
//
// A wave serves 16 rays.  Lane 16 j + r (row j = 0..3) holds ray r's state (t, step, tsdf, tmax) --
// replicated, every row computes it from the same inputs -- and evaluates the reference's iteration j + 1
// counted from the current state UNDER THE ASSUMPTION that iterations 1..j change nothing but the
// raylength: t_1 = t + step, t_2 = t_1 + step, ... by sequential float adds, i.e. exactly the values of
// `raylength += raystep` (TSDF.cu:523) while raystep holds.  Iteration j is TRANSPARENT iff it is skipped
// (TSDF.cu:525-528: the sample is outside, `continue`) or the step after the 0.8 / 1.0 tests
// (TSDF.cu:537-540) is the step before them and the signs of the previous and the new blend are equal (so
// neither crossing test, TSDF.cu:533 / 541, can fire); `raylength > max` is never transparent.  The first
// non-transparent row j* runs the reference's full iteration body for its sample with the blend of the last
// accepted sample before it as `tsdf`; rows behind it hold speculation on a state that did not
// materialise and are dropped.  The ray's new state -- (t_j*, step and tsdf after that body), or (t_4, step, last
// blend) when all four rows were transparent -- goes back to the four rows through two ds_bpermute.
//   Exact by construction: every accepted sample is taken at the raylength, with the step and against the
// previous blend the reference's loop would have had, in the reference's order; what differs is only that
// up to three samples per wave iteration are taken for nothing.
//   "sign of the previous blend": row j tests against the sign of the STATE's tsdf, not of row j - 1's
// blend.  The first row for which (step changes | sign differs from the state's) is the first
// non-transparent one either way: all accepted samples before it have the state's sign.
//   Rows are 16 lanes apart, not adjacent, so that the four lanes the vector L1 looks up together (one
// tag look-up per aligned group of four lanes) are four x-adjacent pixels at the SAME sample, as in
// march_lane -- a quad of consecutive samples of one ray would touch four z planes = four lines.
//   ROWS = 2 (lanes r and r + 32, 32 rays per wave) is the same scheme with two rows: half the chain for ~5 % more
// gathers, against a third of the chain for 20-60 % more with four.
template <int ROWS, bool RCP, class Sink>
__device__ __forceinline__ void march_quad(const RayVolume& v, const V3& half, const V3& nf, RayState& r,
                                           MarchCount& out, Sink& sink, int lane) {
    const unsigned sy = 4u * static_cast<unsigned>(v.n.x), sz = sy * static_cast<unsigned>(v.n.y);
    const float* const row00 = scalar_ptr(v.tsdf);
    const float* const row01 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sy));
    const float* const row10 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sz));
    const float* const row11 = scalar_ptr(reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + sz + sy));
    const float vs = v.voxelSize, hvs = 0.5f * v.voxelSize, rcp = v.rcpVoxel;
    const unsigned limx = static_cast<unsigned>(max(v.n.x - 3, 0)), limy = static_cast<unsigned>(max(v.n.y - 3, 0)),
                   limz = static_cast<unsigned>(max(v.n.z - 3, 0));
    static_assert(ROWS == 2 || ROWS == 4, "two or four lanes per ray");
    constexpr unsigned RAYS = 64u / ROWS, SHIFT = ROWS == 4 ? 4u : 5u;
    const unsigned ray = static_cast<unsigned>(lane) & (RAYS - 1u), row = static_cast<unsigned>(lane) >> SHIFT;
    const unsigned myBit = 1u << (8u * row);  // this row's bit in the per-ray byte masks below
    const V3 dir = r.dir;
    const float tmax = r.maxRay;
    float t = r.raylength, step = r.raystep, tsdf = r.tsdf;
    unsigned samples = 0, gathered = 0;
    for (;;) {
        const float t1 = t + step, t2 = t1 + step, t3 = ROWS == 4 ? t2 + step : t2, t4 = ROWS == 4 ? t3 + step : t2;
        if (!(t1 <= tmax)) break;  // (per ray: its rows leave together)
        const float tj = ROWS == 4 ? (row == 0u ? t1 : row == 1u ? t2 : row == 2u ? t3 : t4) : (row == 0u ? t1 : t2);
        const bool over = !(tj <= tmax);
        const V3 pm = v.cam + dir * tj;
        V3 p;
        if (RCP) {
            p = v3(div_voxel(pm.x, vs, rcp), div_voxel(pm.y, vs, rcp), div_voxel(pm.z, vs, rcp)) + half;
        } else {
            p = pm / vs + half;
        }
        const int lx = static_cast<int>(floorf(p.x)), ly = static_cast<int>(floorf(p.y)),
                  lz = static_cast<int>(floorf(p.z));
        bool inside = (static_cast<unsigned>(lx) < limx) & (static_cast<unsigned>(ly) < limy) &
                      (static_cast<unsigned>(lz) < limz);
        if (!inside) inside = !outside_flat(p, 2.f, nf);
        inside = inside && !over;
        float next = 0.f, nstep = step, fx = 0.f, fy = 0.f, fz = 0.f;
        unsigned off = 0u;
        bool q = over, signs = false;
        if (inside) {
            fx = __builtin_amdgcn_fractf(p.x);
            fy = __builtin_amdgcn_fractf(p.y);
            fz = __builtin_amdgcn_fractf(p.z);
            off = mad24(static_cast<unsigned>(lz), sz, mad24(static_cast<unsigned>(ly), sy, static_cast<unsigned>(lx) << 2));
            const pair_f a = gload2(row00, off), b = gload2(row01, off), d = gload2(row10, off),
                         e = gload2(row11, off);
            next = blend8(a.x, a.y, b.x, b.y, d.x, d.y, e.x, e.y, fx, fy, fz);
            if (fabsf(next) < 1.f) nstep = vs;
            if (fabsf(next) < .8f) nstep = hvs;
            signs = (__float_as_int(tsdf) ^ __float_as_int(next)) < 0;
            q = (nstep != step) | signs;
        }
        // per ray: byte k of `x` / `ia` = row k's flag
        const unsigned long long qb = __builtin_amdgcn_ballot_w64(q), ib = __builtin_amdgcn_ballot_w64(inside);
        unsigned x, iall;
        if (ROWS == 4) {  // bits ray, ray + 16, ray + 32, ray + 48 -> bytes 0..3
            const unsigned long long qm = qb >> ray, im = ib >> ray;
            x = __builtin_amdgcn_perm(static_cast<unsigned>(qm >> 32), static_cast<unsigned>(qm), 0x06040200u) & 0x01010101u;
            iall = __builtin_amdgcn_perm(static_cast<unsigned>(im >> 32), static_cast<unsigned>(im), 0x06040200u) & 0x01010101u;
        } else {  // bits ray, ray + 32 -> bytes 0, 1
            x = ((static_cast<unsigned>(qb) >> ray) & 1u) | (((static_cast<unsigned>(qb >> 32) >> ray) & 1u) << 8);
            iall = ((static_cast<unsigned>(ib) >> ray) & 1u) | (((static_cast<unsigned>(ib >> 32) >> ray) & 1u) << 8);
        }
        const unsigned acc = x ^ (x - 1u);  // rows up to and including the first non-transparent one (all: none is)
        const unsigned ia = iall & acc;     // accepted samples
        samples += __popc(ia);
        gathered += __popc(iall);
        const bool body = (x & acc & myBit) != 0u && !over;  // this lane runs the full iteration
        float outT = next, outS = over ? -1.f : nstep;       // step < 0: the ray is done
        if (__builtin_amdgcn_ballot_w64(body && signs) != 0ull) {  // (wave-uniform) somebody's crossing tests need the previous blend
            const unsigned ibf = ia & (myBit - 1u);  // accepted samples before this row
            const int src = ibf ? static_cast<int>((((31u - __clz(ibf)) >> 3) << SHIFT) | ray) : lane;
            const float got = __shfl(next, src);
            const float prev = ibf ? got : tsdf;
            if (body && signs) {
                const Cell32 c{off, fx, fy, fz};
                if (prev < 0 && next > 0 && trilinear_weights_g(v, c) > 0.f) {
                    outS = -1.f;  // crossing from behind: `break`
                } else if (prev > 0 && next < 0) {
                    // interpolated crossing; uses the UPDATED raystep (Q1, TSDF.cu:537-543)
                    const float tstar = tj - nstep * prev / (next - prev);
                    const V3 ps = to_voxel(v.cam + dir * tstar, v, half);
                    if (outside_flat(ps, 2.f, nf)) {
                        outT = prev;  // reference `continue`: the step is updated, tsdf is NOT
                    } else {
                        const Cell32 cs = cell32_of(ps, v.n);
                        if (trilinear_weights_g(v, cs) > 0.f) {
                            const V3 g = gradient_at(v, widen(cs));
                            const M33 Rt = transpose(v.R);
                            out.hit = true;
                            sink(tstar, mul(Rt, dir * tstar), mul(Rt, g / norm(g)));  // 0/0 -> NaN like the reference
                            outS = -1.f;  // `break`
                        }
                    }
                }
            }
        }
        // the ray's new state: from row j*, or from the last sampled row when all four were transparent
        const unsigned sb = x ? (x & acc) : ia;
        const int src = sb ? static_cast<int>((((31u - __clz(sb)) >> 3) << SHIFT) | ray) : lane;
        const float nt = __shfl(outT, src), ns = __shfl(outS, src);
        if (sb) {
            tsdf = nt;
            step = ns;
        }
        t = ROWS == 4 ? ((x & 1u) ? t1 : (x & 0x100u) ? t2 : (x & 0x10000u) ? t3 : t4) : ((x & 1u) ? t1 : t2);
        if (step < 0.f) break;
    }
    out.samples = row == 0u ? samples : 0u;
    out.gathered = row == 0u ? gathered : 0u;
}

// March the ray of pixel (x, y); `valid` = the pixel exists.  All 64 lanes of the wave call this.
// The volume must fit 32-bit byte offsets (Nx Ny Nz <= 2^30): the caller checks.
template <class Sink>
__device__ __forceinline__ MarchCount march_wave(const RayVolume& v, bool valid, int x, int y,
                                                 float fx, float fy, float cx, float cy,
                                                 float oldRaylength, Sink& sink,
                                                 float cut = __builtin_inff()) {
    MarchCount out;
    out.hit = false;
    out.samples = out.gathered = 0;
#ifdef EMF_MARCH_STAMP
    out.ckIssue = out.ckWait = out.ckRest = out.lateIssue = out.lateWait = out.lateRest = 0;
    out.iters = 0;
    for (int k = 0; k < 6; ++k) out.hist[k] = 0;
#endif
    const V3 half = half_extent(v.n);
    const V3 nf = v3(static_cast<float>(v.n.x), static_cast<float>(v.n.y), static_cast<float>(v.n.z));
    RayState r;
    r.dir = v3(0.f, 0.f, 1.f);
    r.raylength = r.maxRay = r.raystep = r.tsdf = 0.f;
    r.active = false;
    if (valid) ray_setup(v, half, nf, x, y, fx, fy, cx, cy, oldRaylength, cut, r);
    if (r.active) {
        if (v.rcpVoxel != 0.f)  // wave-uniform
            march_lane<true>(v, half, nf, r, out, sink);
        else
            march_lane<false>(v, half, nf, r, out, sink);
    }
    return out;
}

// The ROWS-lanes-per-ray march for the ray of pixel (x, y): lanes r, r + 64 / ROWS, ... of the wave pass the SAME
// pixel (and the same `valid`).  All 64 lanes call this.  On return `hit` is the ray's (equal in its four lanes), `samples`
// is the ray's count in row 0 and zero in the other rows.  `sink` is called by whichever row found the hit.
template <int ROWS, class Sink>
__device__ __forceinline__ MarchCount march_wave_quad(const RayVolume& v, bool valid, int x, int y, float fx, float fy,
                                                      float cx, float cy, float oldRaylength, Sink& sink, float cut,
                                                      int lane) {
    MarchCount out;
    out.hit = false;
    out.samples = out.gathered = 0;
#ifdef EMF_MARCH_STAMP
    out.ckIssue = out.ckWait = out.ckRest = out.lateIssue = out.lateWait = out.lateRest = 0;
    out.iters = 0;
    for (int k = 0; k < 6; ++k) out.hist[k] = 0;
#endif
    const V3 half = half_extent(v.n);
    const V3 nf = v3(static_cast<float>(v.n.x), static_cast<float>(v.n.y), static_cast<float>(v.n.z));
    RayState r;
    r.dir = v3(0.f, 0.f, 1.f);
    r.raylength = r.maxRay = r.raystep = r.tsdf = 0.f;
    r.active = false;
    if (valid) ray_setup(v, half, nf, x, y, fx, fy, cx, cy, oldRaylength, cut, r);  // (four times the same result)
    if (r.active) {
        if (v.rcpVoxel != 0.f)  // wave-uniform
            march_quad<ROWS, true>(v, half, nf, r, out, sink, lane);
        else
            march_quad<ROWS, false>(v, half, nf, r, out, sink, lane);
    }
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(out.hit) >> (static_cast<unsigned>(lane) & (64u / ROWS - 1u));
    out.hit = (hm & (ROWS == 4 ? 0x0001000100010001ull : 0x0000000100000001ull)) != 0ull;
    return out;
}

}  // namespace emf_hip
