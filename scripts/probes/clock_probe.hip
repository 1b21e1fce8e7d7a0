// Shader clock of the device under different loads: cycles of a dependent VALU chain (known length)
// against the 100 MHz wall clock.  hipcc --offload-arch=gfx950 -O2 scripts/probes/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(float* out, long long* ticks, int iters) {
    float x = threadIdx.x * 1e-3f;
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);  // 64 dependent FMAs
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = w1 - w0;
        ticks[2 * blockIdx.x + 1] = c1 - c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    float* out; long long* ticks;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&ticks, 4096 * 16);
    for (int blocks : {1, 256, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, out, ticks, 20000);
            hipDeviceSynchronize();
        }
        long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        const double us = h[0] / 100.0, fmas = 20000.0 * 64;
        printf("blocks %5d: %.1f us for %.0f dependent FMAs of one wave: %.2f ns per FMA; clock64 delta %lld (%.1f per us)\n",
               blocks, us, fmas, 1e3 * us / fmas, h[1], h[1] / us);
    }
    return 0;
}
