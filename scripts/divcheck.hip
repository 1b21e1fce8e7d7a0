// Exhaustive check: is the reciprocal + FMA refinement bit-identical to IEEE x / d for EVERY float x?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/divcheck.hip -o /tmp/divcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

__device__ __forceinline__ float div_refined(float x, float d, float rcp, int rounds) {
    float q = x * rcp;
    for (int i = 0; i < rounds; ++i) {
        const float r = __builtin_fmaf(-q, d, x);
        q = __builtin_fmaf(r, rcp, q);
    }
    return q;
}

__global__ void k_check(float d, float rcp, int rounds, unsigned long long* bad, unsigned* firstBad,
                        float* minBadAbs, float* maxBadAbs, unsigned long long* zones) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
         i < (1ull << 32); i += stride) {
        const unsigned bits = (unsigned)i;
        const float x = __uint_as_float(bits);
        const float a = x / d;
        const float b = div_refined(x, d, rcp, rounds);
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        const float axx = fabsf(x);
        const int zone = !(axx >= 1e-30f) ? 0 : (axx <= 1e30f ? 1 : 2);  // tiny / working / huge
        if (!same) atomicAdd(&zones[zone], 1ull);
        if (!same) {
            ++nbad;
            atomicMin(firstBad, bits & 0x7fffffffu);
            const float ax = fabsf(x);
            if (ax == ax && ax < 3e38f) {
                atomicMin((unsigned*)minBadAbs, __float_as_uint(ax));
                atomicMax((unsigned*)maxBadAbs, __float_as_uint(ax));
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    std::vector<float> ds = {0.01f, 0.02f, 0.005f, 0.04f, 2.f * 0.3f / 128.f, 2.f * 0.15f / 128.f,
                             0.0046875f, 0.00390625f, 0.0123456f, 0.0077f, 1.f / 3.f, 0.1f};
    unsigned long long* bad; unsigned* first; float *mn, *mx;
    unsigned long long* zones; hipMalloc(&zones, 24); hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMalloc(&mn, 4); hipMalloc(&mx, 4);
    for (int rounds = 1; rounds <= 2; ++rounds)
        for (float d : ds) {
            const float rcp = 1.0f / d;
            unsigned ff = 0x7fffffffu; float big = 3.4e38f, zero = 0.f;
            hipMemset(bad, 0, 8); hipMemset(zones, 0, 24);
            hipMemcpy(first, &ff, 4, hipMemcpyHostToDevice);
            hipMemcpy(mn, &big, 4, hipMemcpyHostToDevice);
            hipMemcpy(mx, &zero, 4, hipMemcpyHostToDevice);
            k_check<<<4096, 256>>>(d, rcp, rounds, bad, first, mn, mx, zones);
            hipDeviceSynchronize();
            unsigned long long nb; float a, b;
            hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
            hipMemcpy(&a, mn, 4, hipMemcpyDeviceToHost);
            hipMemcpy(&b, mx, 4, hipMemcpyDeviceToHost);
            unsigned long long z[3];
            hipMemcpy(z, zones, 24, hipMemcpyDeviceToHost);
            printf("rounds %d d=%.9g mismatches: |x|<1e-30 (incl. -0, NaN) %llu   1e-30..1e30 %llu   >1e30 %llu\n",
                   rounds, d, z[0], z[1], z[2]);
        }
    return 0;
}
