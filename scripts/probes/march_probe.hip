// March probe: what does ONE step of the product ray march (march_wave.hpp) cost, and why?
//   part 1: VALU issue rate of gfx950 (plain vs packed f32, 1..8 waves per SIMD)
//   part 2: the product march on a synthetic 512^3 volume whose every sample keeps the ray marching
//           at half-voxel steps (tsdf == 0.5): one lone wave, one wave per SIMD of a CU, 16 waves on
//           a CU, the whole VGA image; each cold (caches flushed) and warm (second run).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Iemfusion_amd/csrc \
//         scripts/probes/march_probe.hip -o build_tmp/march_probe
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

// ---- part 1 ---------------------------------------------------------------------------------------
template <int KIND>
__global__ void k_valu(float* out, int iters, unsigned long long* cycles) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 8 independent plain multiplies
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(m));
        } else if (KIND == 1) {  // 8 DEPENDENT plain multiplies
            asm volatile(
                "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                : "+v"(a0)
                : "v"(m));
        } else if (KIND == 2) {  // 4 independent packed multiplies (8 products)
            asm volatile(
                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                : "v"(pm));
        } else {  // 4 DEPENDENT packed multiplies
            asm volatile(
                "v_pk_mul_f32 %0, %0, %1\n s_nop 0\n v_pk_mul_f32 %0, %0, %1\n s_nop 0\n v_pk_mul_f32 %0, %0, %1\n s_nop 0\n v_pk_mul_f32 %0, %0, %1\n"
                : "+v"(p0)
                : "v"(pm));
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y +
                                                 p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- part 2 ---------------------------------------------------------------------------------------
struct Rec {
    unsigned long long c0, c1;  // shader clock (s_memtime)
    unsigned long long w0, w1;  // 100 MHz wall clock
    unsigned samples, hw;
};

__global__ __launch_bounds__(1024) void k_march(RayVolume v, int w, int h, float fx, float fy, float cx,
                                                float cy, int tile0, int tilesX, float* sinkBuf, Rec* rec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wavesPerBlock = blockDim.x >> 6;
    const int tile = tile0 + blockIdx.x * wavesPerBlock + wave;
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    const bool valid = x < w && y < h;
    float acc = 0.f;
    auto sink = [&](float raylength, const V3& vertex, const V3& normal) { acc += raylength + vertex.x + normal.x; };
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    const MarchCount c = march_wave(v, valid, x, y, fx, fy, cx, cy, 0.f, sink);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned s = c.samples;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = max(s, (unsigned)__shfl_xor((int)s, o));
    if (acc == 12345.f) sinkBuf[0] = acc;
    if (lane == 0) {
        Rec r;
        r.c0 = c0; r.c1 = c1; r.w0 = w0; r.w1 = w1;
        r.samples = s;
        r.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        rec[blockIdx.x * wavesPerBlock + wave] = r;
    }
}

// the same image with FOUR lanes per ray (march_quad): a wave = a 4x4-pixel block, four waves = the 8x8 tile `tile`
__global__ __launch_bounds__(1024) void k_march_quad(RayVolume v, int w, int h, float fx, float fy, float cx,
                                                     float cy, int tile0, int tilesX, float* sinkBuf, Rec* rec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wavesPerBlock = blockDim.x >> 6;
    const int gw = blockIdx.x * wavesPerBlock + wave;
    const int tile = tile0 + (gw >> 2), sub = gw & 3;
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    const int x = tx * 8 + (sub & 1) * 4 + (lane & 3), y = ty * 8 + (sub >> 1) * 4 + ((lane >> 2) & 3);
    const bool valid = x < w && y < h;
    float acc = 0.f;
    auto sink = [&](float raylength, const V3& vertex, const V3& normal) { acc += raylength + vertex.x + normal.x; };
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    const MarchCount c = march_wave_quad<4>(v, valid, x, y, fx, fy, cx, cy, 0.f, sink, __builtin_inff(), lane);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned s = c.samples;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = max(s, (unsigned)__shfl_xor((int)s, o));
    if (acc == 12345.f) sinkBuf[0] = acc;
    if (lane == 0) {
        Rec r;
        r.c0 = c0; r.c1 = c1; r.w0 = w0; r.w1 = w1;
        r.samples = s;
        r.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        rec[gw] = r;
    }
}

__global__ void k_fill(float* p, size_t n, float a, float b) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // a / b alternate along x + y + z so that blends vary around their mean
        const size_t x = i & 511, y = (i >> 9) & 511, z = i >> 18;
        p[i] = ((x + y + z) & 1) ? a : b;
        // scene 2 (a < 0): free space in front of a wall at z = 400: +1 up to the truncation band, a ramp through
        // zero, the product's common case (one voxel per step, nothing happens until the surface)
        if (a < 0.f) p[i] = z < 390 ? 1.f : z < 410 ? (400.f - (float)z) * 0.1f : -1.f;
    }
}

int main(int argc, char** argv) {
    // ---------------- part 1
    {
        float* out; unsigned long long* cyc;
        CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 8 * 64));
        const int iters = 4096;
        const char* names[4] = {"8 indep v_mul_f32", "8 dep   v_mul_f32", "4 indep v_pk_mul_f32", "4 dep   v_pk_mul_f32"};
        for (int kind = 0; kind < 4; ++kind)
            for (int threads : {64, 256, 512, 1024}) {
                unsigned long long h = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (kind == 0) hipLaunchKernelGGL(k_valu<0>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                    if (kind == 1) hipLaunchKernelGGL(k_valu<1>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                    if (kind == 2) hipLaunchKernelGGL(k_valu<2>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                    if (kind == 3) hipLaunchKernelGGL(k_valu<3>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
                }
                const int perIter = kind < 2 ? 8 : 4;
                std::printf("VALU %-22s threads %4d (%d waves/SIMD): %.2f clk per instruction of wave 0\n", names[kind], threads,
                            std::max(1, threads / 256), double(h) / (double(iters) * perIter));
            }
    }
    // ---------------- part 2
    const int N = 512;
    const size_t vox = (size_t)N * N * N;
    float *tsdf, *wts, *sinkBuf, *flush;
    Rec* rec;
    CK(hipMalloc(&tsdf, vox * 4)); CK(hipMalloc(&wts, vox * 4)); CK(hipMalloc(&sinkBuf, 64));
    const size_t flushBytes = 3ull << 30;
    CK(hipMalloc(&flush, flushBytes));
    CK(hipMalloc(&rec, sizeof(Rec) * 32768));
    const int W = 640, H = 480;
    const float K[4] = {525.f, 525.f, 319.5f, 239.5f};
    RayVolume v{};
    v.tsdf = tsdf; v.grads = nullptr; v.weights = wts; v.fg = nullptr; v.bricks = nullptr; v.blendFromFlags = false;
    v.R = M33{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    v.cam = V3{0.013f, -0.021f, -2.5f};
    v.n = I3{N, N, N};
    v.voxelSize = 0.01f; v.truncdist = 0.1f;
    for (int scene = 0; scene < 3; ++scene) {
        // scene 0: tsdf == 0.5 (half-voxel steps, never a crossing); scene 1: 0.7 / 0.9 checkerboard (blends
        // wander around 0.8: the step size flips between voxel and half voxel like on the frustum boundary)
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tsdf, vox, scene == 2 ? -1.f : scene == 0 ? 0.5f : 0.7f, scene == 0 ? 0.5f : 0.9f);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, wts, vox, 1.f, 1.f);
        CK(hipDeviceSynchronize());
        for (int mode = 0; mode < 3; ++mode) {  // 0: lane march, division; 1: lane march, reciprocal; 2: quad march, reciprocal
            const int rcp = mode > 0, quad = mode == 2;
            v.rcpVoxel = rcp ? 1.0f / v.voxelSize : 0.f;
            struct Case { const char* name; int blocks, threads, tile0; };
            const int tilesX = W / 8;
            const Case cases[] = {{"1 lone wave (centre tile)", 1, 64, 30 * tilesX + 40},
                                  {"1 lone wave (corner tile)", 1, 64, 0},
                                  {"4 waves, one CU (1/SIMD)", 1, 256, 30 * tilesX + 40},
                                  {"16 waves, one CU (4/SIMD)", 1, 1024, 30 * tilesX + 32},
                                  {"whole VGA image (4800 waves)", 1200, 256, 0}};
            for (const Case& cs : cases)
                for (int warm = 0; warm < 2; ++warm) {
                    if (!warm) { CK(hipMemset(flush, warm, flushBytes)); CK(hipDeviceSynchronize()); }
                    // quad: the same pixels with four times the waves (lone wave: a 4x4 block of the tile)
                    const int qblocks = cs.threads == 64 ? 1 : cs.blocks * 4;
                    if (quad)
                        hipLaunchKernelGGL(k_march_quad, dim3(qblocks), dim3(cs.threads), 0, 0, v, W, H, K[0], K[1], K[2], K[3],
                                           cs.tile0, tilesX, sinkBuf, rec);
                    else
                        hipLaunchKernelGGL(k_march, dim3(cs.blocks), dim3(cs.threads), 0, 0, v, W, H, K[0], K[1], K[2], K[3], cs.tile0,
                                           tilesX, sinkBuf, rec);
                    CK(hipDeviceSynchronize());
                    const int nw = (quad ? qblocks : cs.blocks) * cs.threads / 64;
                    std::vector<Rec> hrec(nw);
                    CK(hipMemcpy(hrec.data(), rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
                    double clkPerStep = 0, nsPerStep = 0; unsigned long long wmin = ~0ull, wmax = 0; unsigned smax = 0, smin = ~0u;
                    for (const Rec& r : hrec) {
                        clkPerStep += double(r.c1 - r.c0) / std::max(1u, r.samples);
                        nsPerStep += double(r.w1 - r.w0) * 10.0 / std::max(1u, r.samples);
                        wmin = std::min(wmin, r.w0); wmax = std::max(wmax, r.w1);
                        smax = std::max(smax, r.samples); smin = std::min(smin, r.samples);
                    }
                    std::printf("scene %d %s %-30s %s: samples/ray %u..%u, %.0f shader clk/step, %.0f ns/step, span %.1f us\n", scene,
                                quad ? "quad" : rcp ? "rcp" : "div", cs.name, warm ? "warm" : "cold", smin, smax, clkPerStep / nw, nsPerStep / nw,
                                double(wmax - wmin) * 0.01);
                }
        }
    }
    return 0;
}
