// Brick probe: what does a step of the ray march cost when the volume is stored in
// bricks instead of the reference's linear (z * Ny + y, x) array -- BEFORE the product is changed.
//
// The loop below is the common path of march_lane (march_wave.hpp): t += step, position, floor, bounds,
// 8 corners, blend, sign test, step update -- identical for every layout; only the corner addressing and
// the loads differ:
//   L0  linear: 4 x 8-byte pair loads at one offset from four wave-uniform row bases (the product)
//   L1  bricks BX x BY x BZ (x fastest inside a brick, bricks in (z, y, x) order): 8 dword loads at
//       base + {0, dx} + {0, dy} + {0, dz}, the deltas switching to the brick stride on a brick face
//   L2  same bricks: 4 pair loads + 4 exec-masked dword loads for the lanes whose x pair straddles a brick
//   L3  linear volume + a per-wave LDS WINDOW of 8 x 8 x 8 voxels around the wave's rays: filled with 2 coalesced
//       16-byte loads per lane whenever half of the marching lanes have left it, placed from ONE representative lane
//       (3 v_readlane; lateral -3 .. +4, ahead along the dominant axis); corners come from LDS (4 ds_read2_b32),
//       lanes outside the window gather from memory as in L0
//   L4  the same window on a FIXED SCHEDULE along the rays' dominant axis (window k covers cells a0 + 6k .. + 7 there,
//       laterally it is centred on the tile's central ray), so that window k + 1 is known in advance and its 2 x 16
//       bytes per lane are requested right after window k has been filled and consumed at the next switch: the
//       memory round trip of a fill is off the march's critical path (probe: z-dominant rays only)
// Scenes (512^3, camera inside the front of the volume, VGA pinhole): tsdf == 0.5 (half-voxel steps, every
// second sample re-uses its cell: the old probe's regime), tsdf == 0.9 (ONE voxel per step: what 83-87 % of
// the product's samples are), and a yawed camera (rays at ~25 degrees to the z axis).
// Regimes: one lone wave (cold / warm), 16 waves on one CU, the whole image (4800 waves).
// With COUNT the kernel also counts the distinct 128-byte lines a wave touches per step (summed over its
// load instructions, and over the step as a whole).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Iemfusion_amd/csrc \
//         scripts/probes/brick_probe.hip -o build_tmp/brick_probe
#include "march_wave.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace emf_hip;

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e_));                    \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

struct Rec {
    unsigned long long c0, c1, w0, w1;
    unsigned samples;
    unsigned linesInstr, linesStep;  // COUNT builds: sums over the wave's steps
    float acc;
};

struct ProbeVol {
    const float* tsdf;
    M33 R;
    V3 cam;
    int n;
    float voxelSize, rcp;
    unsigned sy, sz;           // linear: row / plane stride in bytes
    unsigned bsx, bsy, bsz;    // bricks: byte stride between bricks along x, y, z
};

// distinct values of `line` among the active lanes
__device__ __forceinline__ unsigned distinct_lines(unsigned line) {
    unsigned long long rem = __ballot(1);
    unsigned n = 0;
    while (rem) {
        const int l = __ffsll(static_cast<long long>(rem)) - 1;
        const unsigned v = __builtin_amdgcn_readlane(line, l);
        rem &= ~__ballot(line == v);
        ++n;
    }
    return n;
}

template <int LAYOUT, int LBX, int LBY, int LBZ, bool COUNT>
__global__ __launch_bounds__(1024) void k_probe(ProbeVol v, int w, int h, float fx, float fy, float cx, float cy,
                                                int tile0, int tilesX, float tmaxAll, Rec* rec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wavesPerBlock = blockDim.x >> 6;
    const int tile = tile0 + blockIdx.x * wavesPerBlock + wave;
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    const V3 unproj = v3((static_cast<float>(px) - cx) / fx, (static_cast<float>(py) - cy) / fy, 1.f);
    const V3 rayv = mul(v.R, unproj);
    const V3 dir = rayv / norm(rayv);
    const float vs = v.voxelSize, hvs = 0.5f * vs, rcp = v.rcp;
    const float hf = static_cast<float>(v.n - 1) / 2.f;
    const V3 half = v3(hf, hf, hf);
    const unsigned lim = static_cast<unsigned>(v.n - 3);
    const float* const row00 = v.tsdf;
    const float* const row01 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + v.sy);
    const float* const row10 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + v.sz);
    const float* const row11 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tsdf) + v.sz + v.sy);
    constexpr unsigned BX = 1u << LBX, BY = 1u << LBY, BZ = 1u << LBZ;
    // L3: this wave's window (8 x 8 x 8 floats, x fastest) and its origin, wave-uniform
    extern __shared__ __attribute__((aligned(16))) float winAll[];  // 2 KB per wave of the block
    float* const win = winAll + wave * 512;
    int ox = -100000, oy = -100000, oz = -100000;
    int winCooldown = 0;
    // L4: the central ray in voxel units (wave-uniform), the schedule's start, the prefetched window
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    f4u pf0 = {0.f, 0.f, 0.f, 0.f}, pf1 = pf0;
    int kcur = -1000000, kpf = -1000000, a0 = 0;
    float repPx = 0.f, repPy = 0.f, repPz = 0.f, repDx = 0.f, repDy = 0.f, repDz = 1.f;
    if (LAYOUT == 4) {
        const V3 p0 = v3(v.cam.x / vs, v.cam.y / vs, v.cam.z / vs) + half;
        repPx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0.x), 27));
        repPy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0.y), 27));
        repPz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0.z), 27));
        repDx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.x / vs), 27));
        repDy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.y / vs), 27));
        repDz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.z / vs), 27));
        a0 = static_cast<int>(floorf(repPz)) - 1;
    }
    auto window_origin = [&](int k, int& wx, int& wy, int& wz) {  // wave-uniform
        wz = a0 + 6 * k;
        const float s = (static_cast<float>(wz) + 3.5f - repPz) / repDz;
        wx = static_cast<int>(floorf(repPx + repDx * s)) - 3;
        wy = static_cast<int>(floorf(repPy + repDy * s)) - 3;
        const int hi = v.n - 8;
        wx = min(max(wx, 0), hi); wy = min(max(wy, 0), hi); wz = min(max(wz, 0), hi);
    };
    auto window_loads = [&](int wx, int wy, int wz, f4u& r0, f4u& r1) {  // lane l: chunks l and l + 64 of 128
        const int row0 = lane >> 1, q = lane & 1;
        const unsigned g0 = mad24(static_cast<unsigned>(wz + (row0 >> 3)), v.sz, mad24(static_cast<unsigned>(wy + (row0 & 7)), v.sy, static_cast<unsigned>(wx + 4 * q) << 2));
        r0 = *(const __attribute__((address_space(1))) f4u*)((gchar_p)v.tsdf + g0);
        r1 = *(const __attribute__((address_space(1))) f4u*)((gchar_p)v.tsdf + g0 + 4u * v.sz);  // rows 32..63: four planes on
    };
    float t = vs, step = vs, tsdf = 1.f, tmax = tmaxAll, acc = 0.f;
    unsigned samples = 0, linesInstr = 0, linesStep = 0;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (;;) {
        t += step;
        if (!(t <= tmax)) break;
        const V3 pm = v.cam + dir * t;
        const V3 p = v3(div_voxel(pm.x, vs, rcp), div_voxel(pm.y, vs, rcp), div_voxel(pm.z, vs, rcp)) + half;
        const int lx = static_cast<int>(floorf(p.x)), ly = static_cast<int>(floorf(p.y)), lz = static_cast<int>(floorf(p.z));
        const bool inside = (static_cast<unsigned>(lx) < lim) & (static_cast<unsigned>(ly) < lim) & (static_cast<unsigned>(lz) < lim);
        if (inside) {
            ++samples;
            const float fxx = __builtin_amdgcn_fractf(p.x), fyy = __builtin_amdgcn_fractf(p.y), fzz = __builtin_amdgcn_fractf(p.z);
            float c[8];
            if (LAYOUT == 4) {
                bool inWin = max(max(static_cast<unsigned>(lx - ox), static_cast<unsigned>(ly - oy)), static_cast<unsigned>(lz - oz)) < 7u;
                const unsigned long long missing = __ballot(!inWin);
                if (__popcll(missing) > 32) {  // (the probe's 64 lanes all march to the end: no zombies needed here)
                    const int first = __ffsll(static_cast<long long>(missing)) - 1;
                    const int zl = __builtin_amdgcn_readlane(lz, first);
                    const int knew = (zl - a0) / 6;
                    window_origin(knew, ox, oy, oz);
                    if (knew != kpf) window_loads(ox, oy, oz, pf0, pf1);  // not what was requested ahead: fetch it now
                    const int row0 = lane >> 1, q = lane & 1;
                    *reinterpret_cast<f4u*>(win + row0 * 8 + 4 * q) = pf0;
                    *reinterpret_cast<f4u*>(win + (row0 + 32) * 8 + 4 * q) = pf1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    kcur = knew;
                    int nx, ny, nz;  // request the next window now; nobody waits for it before the next switch
                    window_origin(knew + 1, nx, ny, nz);
                    window_loads(nx, ny, nz, pf0, pf1);
                    kpf = knew + 1;
                    inWin = max(max(static_cast<unsigned>(lx - ox), static_cast<unsigned>(ly - oy)), static_cast<unsigned>(lz - oz)) < 7u;
                }
                if (inWin) {
                    const float* b = win + (((lz - oz) * 8 + (ly - oy)) * 8 + (lx - ox));
                    c[0] = b[0]; c[1] = b[1]; c[2] = b[8]; c[3] = b[9]; c[4] = b[64]; c[5] = b[65]; c[6] = b[72]; c[7] = b[73];
                } else {
                    const unsigned off = mad24(static_cast<unsigned>(lz), v.sz, mad24(static_cast<unsigned>(ly), v.sy, static_cast<unsigned>(lx) << 2));
                    const pair_f a = gload2(row00, off), b = gload2(row01, off), d = gload2(row10, off), e = gload2(row11, off);
                    c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = d.x; c[5] = d.y; c[6] = e.x; c[7] = e.y;
                }
                if (COUNT) { linesInstr += inWin ? 0u : 4u; linesStep += __popcll(__ballot(inWin)); }
            } else if (LAYOUT == 3) {
                // ---- refill policy: wave-uniform, only lanes that march take part ----
                const unsigned rx0 = static_cast<unsigned>(lx - ox), ry0 = static_cast<unsigned>(ly - oy), rz0 = static_cast<unsigned>(lz - oz);
                bool inWin = max(max(rx0, ry0), rz0) < 7u;
                const unsigned long long marching = __ballot(1), missing = __ballot(!inWin);
                if (winCooldown > 0) --winCooldown;
                if (2 * __popcll(missing) > __popcll(marching) && winCooldown == 0) {
                    // representative: the marching lane nearest the tile's centre (lane 27), else the first one
                    const int rep = (marching >> 27) & 1ull ? 27 : __ffsll(static_cast<long long>(marching)) - 1;
                    const int cx_ = __builtin_amdgcn_readlane(lx, rep), cy_ = __builtin_amdgcn_readlane(ly, rep), cz_ = __builtin_amdgcn_readlane(lz, rep);
                    const float dx_ = __builtin_amdgcn_readlane(__float_as_int(dir.x), rep) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.x), rep)) : 0.f;
                    const float dy_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.y), rep));
                    const float dz_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dir.z), rep));
                    const float ax_ = fabsf(dx_), ay_ = fabsf(dy_), az_ = fabsf(dz_);
                    int nox = cx_ - 3, noy = cy_ - 3, noz = cz_ - 3;
                    if (ax_ >= ay_ && ax_ >= az_) nox = dx_ > 0.f ? cx_ - 1 : cx_ - 5;
                    else if (ay_ >= az_) noy = dy_ > 0.f ? cy_ - 1 : cy_ - 5;
                    else noz = dz_ > 0.f ? cz_ - 1 : cz_ - 5;
                    const int hi = v.n - 8;
                    ox = min(max(nox, 0), hi); oy = min(max(noy, 0), hi); oz = min(max(noz, 0), hi);
                    // fill: 128 chunks of 4 floats over the marching lanes
                    const int nw = __popcll(marching);
                    const int me = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(marching >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(marching), 0u));
                    for (int ch = me; ch < 128; ch += nw) {
                        const int row = ch >> 1, q = ch & 1, wy = row & 7, wz = row >> 3;
                        const unsigned goff = mad24(static_cast<unsigned>(oz + wz), v.sz, mad24(static_cast<unsigned>(oy + wy), v.sy, static_cast<unsigned>(ox + 4 * q) << 2));
                            const f4u val = *(const __attribute__((address_space(1))) f4u*)((gchar_p)v.tsdf + goff);
                        float* d = win + row * 8 + 4 * q;
                        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const unsigned rx1 = static_cast<unsigned>(lx - ox), ry1 = static_cast<unsigned>(ly - oy), rz1 = static_cast<unsigned>(lz - oz);
                    inWin = max(max(rx1, ry1), rz1) < 7u;
                    if (2 * __popcll(__ballot(inWin)) < nw) winCooldown = 16;  // the tube does not fit: leave it for a while
                }
                if (inWin) {
                    const float* b = win + (((lz - oz) * 8 + (ly - oy)) * 8 + (lx - ox));
                    c[0] = b[0]; c[1] = b[1]; c[2] = b[8]; c[3] = b[9]; c[4] = b[64]; c[5] = b[65]; c[6] = b[72]; c[7] = b[73];
                } else {
                    const unsigned off = mad24(static_cast<unsigned>(lz), v.sz, mad24(static_cast<unsigned>(ly), v.sy, static_cast<unsigned>(lx) << 2));
                    const pair_f a = gload2(row00, off), b = gload2(row01, off), d = gload2(row10, off), e = gload2(row11, off);
                    c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = d.x; c[5] = d.y; c[6] = e.x; c[7] = e.y;
                }
                if (COUNT) { linesInstr += inWin ? 0u : 4u; linesStep += __popcll(__ballot(inWin)); }
            } else if (LAYOUT == 0) {
                const unsigned off = mad24(static_cast<unsigned>(lz), v.sz, mad24(static_cast<unsigned>(ly), v.sy, static_cast<unsigned>(lx) << 2));
                const pair_f a = gload2(row00, off), b = gload2(row01, off), d = gload2(row10, off), e = gload2(row11, off);
                c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = d.x; c[5] = d.y; c[6] = e.x; c[7] = e.y;
                if (COUNT) {
                    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(v.tsdf) >> 7);
                    (void)base;
                    const unsigned l0 = off >> 7, l1 = (off + v.sy) >> 7, l2 = (off + v.sz) >> 7, l3 = (off + v.sz + v.sy) >> 7;
                    linesInstr += distinct_lines(l0) + distinct_lines(l1) + distinct_lines(l2) + distinct_lines(l3);
                    // the step's line set: the four instructions touch disjoint rows (a pair may straddle: ignored)
                    linesStep += distinct_lines(l0) + distinct_lines(l1) + distinct_lines(l2) + distinct_lines(l3);
                }
            } else {
                const unsigned ux = static_cast<unsigned>(lx), uy = static_cast<unsigned>(ly), uz = static_cast<unsigned>(lz);
                // byte offset = sum of per-axis terms: in-brick index * stride + brick index * brick stride
                const unsigned ox = mad24(ux & ~(BX - 1), (v.bsx >> LBX) - 4u, ux << 2);
                const unsigned oy = mad24(uy & ~(BY - 1), (v.bsy >> LBY) - 4u * BX, uy * (4u * BX));
                const unsigned oz = mad24(uz & ~(BZ - 1), (v.bsz >> LBZ) - 4u * BX * BY, uz * (4u * BX * BY));
                const unsigned off = ox + oy + oz;
                const bool faceX = (ux & (BX - 1)) == BX - 1;
                const unsigned dx = faceX ? v.bsx - 4u * (BX - 1) : 4u;
                const unsigned dy = (uy & (BY - 1)) == BY - 1 ? v.bsy - 4u * BX * (BY - 1) : 4u * BX;
                const unsigned dz = (uz & (BZ - 1)) == BZ - 1 ? v.bsz - 4u * BX * BY * (BZ - 1) : 4u * BX * BY;
                const unsigned o00 = off, o01 = off + dy, o10 = off + dz, o11 = off + dz + dy;
                if (LAYOUT == 1) {
                    c[0] = gload(v.tsdf, o00); c[1] = gload(v.tsdf, o00 + dx);
                    c[2] = gload(v.tsdf, o01); c[3] = gload(v.tsdf, o01 + dx);
                    c[4] = gload(v.tsdf, o10); c[5] = gload(v.tsdf, o10 + dx);
                    c[6] = gload(v.tsdf, o11); c[7] = gload(v.tsdf, o11 + dx);
                } else {
                    const pair_f a = gload2(v.tsdf, o00), b = gload2(v.tsdf, o01), d = gload2(v.tsdf, o10), e = gload2(v.tsdf, o11);
                    c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = d.x; c[5] = d.y; c[6] = e.x; c[7] = e.y;
                    if (faceX) {  // the x neighbour lives in the next brick
                        c[1] = gload(v.tsdf, o00 + dx); c[3] = gload(v.tsdf, o01 + dx);
                        c[5] = gload(v.tsdf, o10 + dx); c[7] = gload(v.tsdf, o11 + dx);
                    }
                }
                if (COUNT) {
                    const unsigned o[8] = {o00, o00 + dx, o01, o01 + dx, o10, o10 + dx, o11, o11 + dx};
                    unsigned perInstr = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) perInstr += distinct_lines(o[k] >> 7);
                    linesInstr += perInstr;
                    // lines of the whole step: distinct over all 8 corners of all lanes (8 rounds over the lanes' sets)
                    unsigned stepLines = 0;
                    unsigned seen[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) seen[k] = o[k] >> 7;
                    // count line values not already counted by an earlier corner round
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        unsigned long long rem = __ballot(1);
                        while (rem) {
                            const int l = __ffsll(static_cast<long long>(rem)) - 1;
                            const unsigned val = __builtin_amdgcn_readlane(seen[k], l);
                            rem &= ~__ballot(seen[k] == val);
                            bool dup = false;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (j < k) dup = dup || __ballot(seen[j] == val) != 0ull;
                            stepLines += dup ? 0u : 1u;
                        }
                    }
                    linesStep += stepLines;
                }
            }
            const float next = blend8(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], fxx, fyy, fzz);
            if ((__float_as_int(tsdf) ^ __float_as_int(next)) < 0) {
                acc += next;  // (never taken in the probe's scenes; keeps the test in the loop)
                tmax = -__builtin_inff();
            }
            if (fabsf(next) < 1.f) step = vs;
            if (fabsf(next) < .8f) step = hvs;
            tsdf = next;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned s = samples;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = max(s, (unsigned)__shfl_xor((int)s, o));
    acc += tsdf;
    if (lane == 0) {
        Rec r;
        r.c0 = c0; r.c1 = c1; r.w0 = w0; r.w1 = w1;
        r.samples = s;
        r.linesInstr = linesInstr; r.linesStep = linesStep;
        r.acc = acc;
        rec[blockIdx.x * wavesPerBlock + wave] = r;
    }
}

// value of voxel (x, y, z): a / b alternate along x + y + z
__global__ void k_fill(float* p, int n, int layout, int lbx, int lby, int lbz, float a, float b) {
    const size_t total = (size_t)n * n * n;
    const unsigned BX = 1u << lbx, BY = 1u << lby, BZ = 1u << lbz;
    const size_t nbx = n >> lbx, nby = n >> lby;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned x = i % n, y = (i / n) % n, z = i / ((size_t)n * n);
        const float val = ((x + y + z) & 1) ? a : b;
        size_t idx = i;
        if (layout)
            idx = (((size_t)(z >> lbz) * nby + (y >> lby)) * nbx + (x >> lbx)) * (BX * BY * BZ) +
                  ((z & (BZ - 1)) * BY + (y & (BY - 1))) * BX + (x & (BX - 1));
        p[idx] = val;
    }
}

template <int LAYOUT, int LBX, int LBY, int LBZ>
static void launch(bool count, int blocks, int threads, const ProbeVol& v, int W, int H, const float K[4], int tile0, int tilesX,
                   float tmax, Rec* rec) {
    if (count)
        hipLaunchKernelGGL((k_probe<LAYOUT, LBX, LBY, LBZ, true>), dim3(blocks), dim3(threads), (threads / 64) * 2048, 0, v, W, H, K[0], K[1], K[2], K[3],
                           tile0, tilesX, tmax, rec);
    else
        hipLaunchKernelGGL((k_probe<LAYOUT, LBX, LBY, LBZ, false>), dim3(blocks), dim3(threads), (threads / 64) * 2048, 0, v, W, H, K[0], K[1], K[2], K[3],
                           tile0, tilesX, tmax, rec);
}

int main(int argc, char** argv) {
    const int N = 512;
    const size_t vox = (size_t)N * N * N;
    float *tsdf, *flush;
    Rec* rec;
    CK(hipMalloc(&tsdf, vox * 4));
    const size_t flushBytes = 3ull << 30;
    CK(hipMalloc(&flush, flushBytes));
    CK(hipMalloc(&rec, sizeof(Rec) * 8192));
    const int W = 640, H = 480;
    const float K[4] = {525.f, 525.f, 319.5f, 239.5f};
    struct Layout { const char* name; int layout, lbx, lby, lbz; };
    const Layout layouts[] = {{"L0 linear, 4 pair loads          ", 0, 0, 0, 0},
                              {"L1 bricks 4x4x4, 8 dword loads   ", 1, 2, 2, 2},
                              {"L2 bricks 4x4x4, 4 pairs + x fix ", 2, 2, 2, 2},
                              {"L1 bricks 8x2x4, 8 dword loads   ", 1, 3, 1, 2},
                              {"L2 bricks 8x2x4, 4 pairs + x fix ", 2, 3, 1, 2},
                              {"L2 bricks 8x4x4, 4 pairs + x fix ", 2, 3, 2, 2},
                              {"L3 linear + 8^3 LDS window       ", 3, 0, 0, 0},
                              {"L4 window on a schedule, prefetch", 4, 0, 0, 0}};
    struct Scene { const char* name; float a, b; float yawDeg; };
    const Scene scenes[] = {{"half-voxel steps (tsdf 0.5)", 0.5f, 0.5f, 0.f},
                            {"ONE voxel per step (0.9)   ", 0.9f, 0.9f, 0.f},
                            {"one voxel per step, yaw 25 ", 0.9f, 0.9f, 25.f}};
    struct Case { const char* name; int blocks, threads, tile0; };
    const int tilesX = W / 8;
    const Case cases[] = {{"1 lone wave (centre)", 1, 64, 30 * tilesX + 40},
                          {"1 lone wave (corner)", 1, 64, 0},
                          {"16 waves on one CU  ", 1, 1024, 30 * tilesX + 32},
                          {"whole image (4800 w)", 1200, 256, 0}};
    for (const Scene& sc : scenes)
        for (const Layout& L : layouts) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tsdf, N, L.layout >= 3 ? 0 : L.layout, L.lbx, L.lby, L.lbz, sc.a, sc.b);
            CK(hipDeviceSynchronize());
            ProbeVol v{};
            v.tsdf = tsdf;
            const float yaw = sc.yawDeg * 3.14159265f / 180.f;
            v.R = M33{{cosf(yaw), 0, sinf(yaw)}, {0, 1, 0}, {-sinf(yaw), 0, cosf(yaw)}};
            v.cam = V3{0.013f - 1.0f * sinf(yaw) * 2.f, -0.021f, -2.5f};
            v.n = N;
            v.voxelSize = 0.01f;
            v.rcp = 1.0f / v.voxelSize;
            v.sy = 4u * N; v.sz = 4u * N * N;
            const unsigned brickBytes = 4u << (L.lbx + L.lby + L.lbz);
            v.bsx = brickBytes; v.bsy = brickBytes * (N >> L.lbx); v.bsz = v.bsy * (N >> L.lby);
            const float tmax = 4.8f;  // metres: ~480 voxel steps / ~960 half-voxel steps
            for (const Case& cs : cases)
                for (int mode = 0; mode < 3; ++mode) {  // 0 cold, 1 warm, 2 warm + line counts
                    if (mode == 2 && cs.blocks > 1) continue;
                    if (mode == 0) { CK(hipMemset(flush, 0, flushBytes)); CK(hipDeviceSynchronize()); }
                    const bool count = mode == 2;
#define GO(LY, A, B, C) if (L.layout == LY && L.lbx == A && L.lby == B && L.lbz == C) launch<LY, A, B, C>(count, cs.blocks, cs.threads, v, W, H, K, cs.tile0, tilesX, tmax, rec)
                    GO(0, 0, 0, 0); GO(3, 0, 0, 0); GO(4, 0, 0, 0); GO(1, 2, 2, 2); GO(2, 2, 2, 2); GO(1, 3, 1, 2); GO(2, 3, 1, 2); GO(2, 3, 2, 2);
#undef GO
                    CK(hipDeviceSynchronize());
                    const int nw = cs.blocks * cs.threads / 64;
                    std::vector<Rec> hrec(nw);
                    CK(hipMemcpy(hrec.data(), rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
                    double clk = 0, ns = 0, li = 0, ls = 0; unsigned long long wmin = ~0ull, wmax = 0; unsigned smax = 0, smin = ~0u;
                    for (const Rec& r : hrec) {
                        clk += double(r.c1 - r.c0) / std::max(1u, r.samples);
                        ns += double(r.w1 - r.w0) * 10.0 / std::max(1u, r.samples);
                        li += double(r.linesInstr) / std::max(1u, r.samples);
                        ls += double(r.linesStep) / std::max(1u, r.samples);
                        wmin = std::min(wmin, r.w0); wmax = std::max(wmax, r.w1);
                        smax = std::max(smax, r.samples); smin = std::min(smin, r.samples);
                    }
                    if (mode < 2)
                        std::printf("%s | %s | %s %s: steps %u..%u, %5.0f clk/step, %5.0f ns/step, span %7.1f us\n", sc.name, L.name, cs.name,
                                    mode ? "warm" : "cold", smin, smax, clk / nw, ns / nw, double(wmax - wmin) * 0.01);
                    else
                        std::printf("%s | %s | %s lines: %.1f per step summed over the load instructions, %.1f distinct per step\n", sc.name,
                                    L.name, cs.name, li / nw, ls / nw);
                }
        }
    return 0;
}
