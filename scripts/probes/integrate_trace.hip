// Timeline probe for the tiled integration: runs integrate_tile (device_core.hpp, the product code)
// on a bench-like scene and records, per workgroup, start / end of the 100 MHz wall clock, the CU
// it ran on and its tile class.  Prints concurrency, per-class durations and the kernel span.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Iemfusion_amd/csrc \
//         scripts/probes/integrate_trace.hip -o build_tmp/integrate_trace
#include "device_core.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

using namespace emf_hip;

struct Rec {
    unsigned long long t0, t1;
    unsigned hw, xcc, cls, pad;
};

__global__ __launch_bounds__(256) void k_probe(const IntegrateGeom a, float* tsdf, float* weights,
                                               int ntx, int nty, Rec* rec) {
    __shared__ unsigned lds[32];
    const unsigned long long t0 = wall_clock64();
    const int b = blockIdx.x;
    const int tx = b % ntx, ty = (b / ntx) % nty, tz = b / (ntx * nty);
    const bool culledTile = tile_culled(a, half_extent(a.n), tx * kTileX, ty * kTileY, tz * kTileZ);
    integrate_tile(a, tsdf, weights, nullptr, tx * kTileX, ty * kTileY, tz * kTileZ, lds);
    __syncthreads();
    if (threadIdx.x == 0 && rec) {
        Rec r;
        r.t0 = t0;
        r.t1 = wall_clock64();
        r.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
        r.cls = culledTile ? 0 : 2;
        r.pad = 0;
        rec[b] = r;
    }
}

// Persistent variant: gridDim.x workgroups draw chunks of `chunk` consecutive tiles from a ticket
// counter (tickets[0]); tickets[1] counts finished workgroups, the last one re-arms both.
__global__ __launch_bounds__(256) void k_probe_persistent(const IntegrateGeom a, float* tsdf,
                                                          float* weights, int ntx, int nty,
                                                          int ntiles, int chunk, unsigned* tickets,
                                                          Rec* rec) {
    __shared__ unsigned lds[32];
    __shared__ unsigned s_next;
    const int nchunks = (ntiles + chunk - 1) / chunk;
    if (threadIdx.x == 0) s_next = atomicAdd(&tickets[0], 1u);
    __syncthreads();
    unsigned cur = s_next;
    while (cur < (unsigned)nchunks) {
        unsigned nxt = 0;
        if (threadIdx.x == 0) nxt = atomicAdd(&tickets[0], 1u);  // in flight while this chunk runs
        for (int b = cur * chunk; b < min((int)(cur + 1) * chunk, ntiles); ++b) {
            const unsigned long long t0 = wall_clock64();
            const int tx = b % ntx, ty = (b / ntx) % nty, tz = b / (ntx * nty);
            const bool culledTile = tile_culled(a, half_extent(a.n), tx * kTileX, ty * kTileY, tz * kTileZ);
            integrate_tile(a, tsdf, weights, nullptr, tx * kTileX, ty * kTileY, tz * kTileZ, lds);
            __syncthreads();
            if (threadIdx.x == 0 && rec) {
                Rec r;
                r.t0 = t0;
                r.t1 = wall_clock64();
                r.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
                r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
                r.cls = culledTile ? 0 : 2;
                r.pad = 0;
                rec[b] = r;
            }
        }
        if (threadIdx.x == 0) s_next = nxt;
        __syncthreads();
        cur = s_next;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
            tickets[0] = 0;
            tickets[1] = 0;
            __threadfence();
        }
    }
}

// Compacted-list variant: gridDim.x resident workgroups stride over a list of live tile ids.
__global__ __launch_bounds__(256) void k_probe_list(const IntegrateGeom a, float* tsdf,
                                                    float* weights, int ntx, int nty,
                                                    const unsigned* list, int count, Rec* rec) {
    __shared__ unsigned lds[32];
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const unsigned long long t0 = wall_clock64();
        const int b = list[i];
        const int tx = b % ntx, ty = (b / ntx) % nty, tz = b / (ntx * nty);
        const bool culledTile = tile_culled(a, half_extent(a.n), tx * kTileX, ty * kTileY, tz * kTileZ);
        integrate_tile(a, tsdf, weights, nullptr, tx * kTileX, ty * kTileY, tz * kTileZ, lds);
        __syncthreads();
        if (threadIdx.x == 0 && rec) {
            Rec r;
            r.t0 = t0;
            r.t1 = wall_clock64();
            r.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
            r.cls = culledTile ? 0 : 2;
            r.pad = 0;
            rec[b] = r;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 512;
    const int W = 640, H = 480;
    const float vox = 5.12f / N;
    std::vector<float> depth(W * H), assoc(W * H, 1.f);
    const float fx = 525.f, fy = 525.f, cx = 319.5f, cy = 239.5f;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            // plane z = 2.2 + 0.15 X - 0.1 Y with X = (x - cx) / fx * z ...
            const float rx = (x - cx) / fx, ry = (y - cy) / fy;
            float z = 2.2f / (1.f - 0.15f * rx + 0.1f * ry);
            // a sphere in front
            const float ox = 0.2f, oy = 0.1f, oz = 1.6f, rad = 0.25f;
            const float dd = rx * rx + ry * ry + 1.f, bq = rx * ox + ry * oy + oz;
            const float disc = bq * bq - dd * (ox * ox + oy * oy + oz * oz - rad * rad);
            if (disc > 0) z = fminf(z, (bq - sqrtf(disc)) / dd);
            depth[y * W + x] = ((x * 131 + y * 71) % 97 == 0) ? 0.f : z;
        }
    float *dDepth, *dAssoc, *dIl, *tsdf, *wts;
    const size_t vol = (size_t)N * N * N;
    CK(hipMalloc(&dDepth, W * H * 4)); CK(hipMalloc(&dAssoc, W * H * 4)); CK(hipMalloc(&dIl, W * H * 4));
    CK(hipMalloc(&tsdf, vol * 4)); CK(hipMalloc(&wts, vol * 4));
    CK(hipMemset(tsdf, 0, vol * 4)); CK(hipMemset(wts, 0, vol * 4));
    CK(hipMemcpy(dDepth, depth.data(), W * H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dAssoc, assoc.data(), W * H * 4, hipMemcpyHostToDevice));
    std::vector<float> il(W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float a = (x - cx) / fx, b = (y - cy) / fy;
            il[y * W + x] = 1.f / sqrtf(a * a + b * b + 1.f);
        }
    CK(hipMemcpy(dIl, il.data(), W * H * 4, hipMemcpyHostToDevice));
    IntegrateGeom g;
    g.depth = Img<const float>{dDepth, (size_t)W * 4};
    g.assoc = Img<const float>{dAssoc, (size_t)W * 4};
    g.invLambda = getenv("NO_TABLE") ? Img<const float>{nullptr, 0} : Img<const float>{dIl, (size_t)W * 4};
    g.w = W; g.h = H;
    g.R = M33{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    g.t = V3{0.f, 0.f, 2.56f};
    g.K = M33{{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    g.n = I3{N, N, N};
    g.voxelSize = vox; g.truncdist = 10 * vox; g.maxWeight = 64.f;
    const int ntx = (N + kTileX - 1) / kTileX, nty = (N + kTileY - 1) / kTileY, ntz = (N + kTileZ - 1) / kTileZ;
    const int nb = ntx * nty * ntz;
    Rec* dRec; CK(hipMalloc(&dRec, sizeof(Rec) * nb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int persistent = argc > 2 ? atoi(argv[2]) : 0;  // grid size of the persistent variant, 0 = off
    const int chunk = argc > 3 ? atoi(argv[3]) : 8;
    unsigned* tickets; CK(hipMalloc(&tickets, 8)); CK(hipMemset(tickets, 0, 8));
    for (int it = 0; it < 6; ++it) {
        g.t.x = 0.01f * it;  // the camera moves a little, as in the bench
        CK(hipEventRecord(e0));
        if (persistent)
            hipLaunchKernelGGL(k_probe_persistent, dim3(persistent), dim3(256), 0, 0, g, tsdf, wts, ntx, nty, nb, chunk, tickets, it == 5 ? dRec : nullptr);
        else
        hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, g, tsdf, wts, ntx, nty, it == 5 ? dRec : nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("launch %d: %.3f ms\n", it, ms);
    }
    std::vector<Rec> rec(nb);
    CK(hipMemcpy(rec.data(), dRec, sizeof(Rec) * nb, hipMemcpyDeviceToHost));
    if (getenv("LIST_GRID")) {
        // second experiment: host-built list of the live tiles of the last pose, strided grid
        const int G = atoi(getenv("LIST_GRID"));
        std::vector<unsigned> list;
        for (int b = 0; b < nb; ++b) if (rec[b].cls != 0) list.push_back(b);
        unsigned* dList; CK(hipMalloc(&dList, list.size() * 4));
        CK(hipMemcpy(dList, list.data(), list.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dRec, 0, sizeof(Rec) * nb));
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe_list, dim3(G), dim3(256), 0, 0, g, tsdf, wts, ntx, nty, dList, (int)list.size(), it == 3 ? dRec : nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("list launch %d (grid %d, %zu live tiles): %.3f ms\n", it, G, list.size(), ms);
        }
        std::vector<Rec> rec2(nb);
        CK(hipMemcpy(rec2.data(), dRec, sizeof(Rec) * nb, hipMemcpyDeviceToHost));
        rec.clear();
        for (auto& r : rec2) if (r.t1) rec.push_back(r);
    }
    const int nrec = (int)rec.size();
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& r : rec) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
    printf("tiles recorded %d, span %.1f us (100 MHz clock)\n", nrec, (tmax - tmin) / 100.0);
    const char* names[3] = {"culled", "staged", "unstaged"};
    for (int c = 0; c < 3; ++c) {
        std::vector<double> d;
        for (auto& r : rec) if ((int)r.cls == c) d.push_back((r.t1 - r.t0) / 100.0);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        double sum = 0; for (double v : d) sum += v;
        printf("  %-8s n=%6zu  mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f   sum %.0f us\n",
               names[c], d.size(), sum / d.size(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10],
               d[d.size() * 99 / 100], d.back(), sum);
    }
    // concurrency over time: sample 40 points
    const int S = 40;
    printf("  resident workgroups over time (all / non-culled):\n   ");
    for (int s = 0; s < S; ++s) {
        const unsigned long long t = tmin + (tmax - tmin) * (2 * s + 1) / (2 * S);
        int all = 0, nc = 0;
        for (auto& r : rec) if (r.t0 <= t && t < r.t1) { ++all; nc += r.cls != 0; }
        printf(" %d/%d", all, nc);
    }
    printf("\n");
    // per-CU busy fraction: union of intervals per (xcc, se, cu)
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> perCu;
    for (auto& r : rec) {
        const unsigned cu = (r.hw >> 8) & 0xf, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
        perCu[(r.xcc & 0xf) << 12 | se << 8 | sh << 4 | cu].push_back({r.t0, r.t1});
    }
    double busySum = 0; size_t maxWg = 0, minWg = 1u << 30;
    for (auto& kv : perCu) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        unsigned long long busy = 0, cs = v[0].first, ce = v[0].second;
        for (auto& iv : v) { if (iv.first > ce) { busy += ce - cs; cs = iv.first; ce = iv.second; } else ce = std::max(ce, iv.second); }
        busy += ce - cs;
        busySum += (double)busy / (tmax - tmin);
        maxWg = std::max(maxWg, v.size()); minWg = std::min(minWg, v.size());
    }
    printf("  distinct CUs seen %zu, mean fraction of the span with >= 1 resident workgroup %.2f, workgroups per CU min %zu max %zu\n",
           perCu.size(), busySum / perCu.size(), minWg, maxWg);
    // dispatch order vs time: start time of every 5000th block
    printf("  start offset (us) of block b:");
    for (int b = 0; b < nrec; b += nrec / 16) printf(" %d:%.0f", b, (rec[b].t0 - tmin) / 100.0);
    printf("\n");
    return 0;
}
