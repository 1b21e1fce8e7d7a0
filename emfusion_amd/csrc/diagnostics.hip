// diagnostics.hip -- what the integration's two arithmetic shortcuts (device_core.hpp: round_pixel<true>,
// band_decision<true>) rest on, made checkable from the tests instead of argued:
//   * emf_hip_sweepFastPathPremises: every one of the 2^32 float bit patterns through v_rcp_f32 and
//     v_sqrt_f32 against the correctly rounded double results -- the "accurate to 1 ulp" premises;
//   * emf_hip_debugPixelRounding / emf_hip_debugBandDecision: the shortcut and the IEEE form of the very
//     device functions the integration kernels call, side by side on caller-chosen inputs (quotients
//     next to every rounding tie k + 1/2, signed distances next to +-truncdist).
// Nothing here is on the frame's path.
#include "device_core.hpp"

namespace emf_hip {
namespace {

// out[0]: z with a normal reciprocal (2^-126 <= |z| <= 2^126) whose v_rcp_f32 is off by more than 2^-23 relative
// out[1]: n >= 2^-126 (finite) whose v_sqrt_f32 is off by more than 2^-23 relative
// out[2], out[3]: the largest relative errors seen in those domains, as bits of the float
__global__ __launch_bounds__(256) void k_sweep_premises(unsigned long long* out) {
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    unsigned badRcp = 0, badSqrt = 0;
    float worstRcp = 0.f, worstSqrt = 0.f;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < (1ull << 32);
         i += stride) {
        const float z = __uint_as_float(static_cast<unsigned>(i));
        const float az = fabsf(z);
        if (az >= 0x1p-126f && az <= 0x1p126f) {  // (false for NaN)
            const double e = fabs(static_cast<double>(__builtin_amdgcn_rcpf(z)) * static_cast<double>(z) - 1.0);
            badRcp += e > 0x1p-23 ? 1u : 0u;
            worstRcp = fmaxf(worstRcp, static_cast<float>(e));
        }
        if (z >= 0x1p-126f && z < __builtin_inff()) {
            const double exact = sqrt(static_cast<double>(z));
            const double e = fabs(static_cast<double>(__builtin_amdgcn_sqrtf(z)) / exact - 1.0);
            badSqrt += e > 0x1p-23 ? 1u : 0u;
            worstSqrt = fmaxf(worstSqrt, static_cast<float>(e));
        }
    }
    if (badRcp) atomicAdd(&out[0], static_cast<unsigned long long>(badRcp));
    if (badSqrt) atomicAdd(&out[1], static_cast<unsigned long long>(badSqrt));
    atomicMax(&out[2], static_cast<unsigned long long>(__float_as_uint(worstRcp)));  // non-negative floats order as integers
    atomicMax(&out[3], static_cast<unsigned long long>(__float_as_uint(worstSqrt)));
}

__global__ void k_debug_pixel(const float* __restrict__ num, const float* __restrict__ den, int n,
                              int32_t* __restrict__ fast, int32_t* __restrict__ exact) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int fx, fy, ex, ey;
    // the second coordinate rides along with a harmless value, as in shoot_voxel both are rounded together
    round_pixel<true>(num[i], 0.f, den[i], false, fx, fy);
    round_pixel<false>(num[i], 0.f, den[i], false, ex, ey);
    fast[i] = fx;
    exact[i] = ex;
}

__global__ void k_debug_band(const float* __restrict__ d, const float* __restrict__ il, const float* __restrict__ n2, int n,
                             float truncdist, int32_t* __restrict__ kindFast, float* __restrict__ sampFast,
                             int32_t* __restrict__ kindExact, float* __restrict__ sampExact) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sf = 0.f, se = 0.f;
    bool bf = false, be = false;
    const int kf = band_decision<true>(d[i], il[i], n2[i], truncdist, sf, bf);
    const int ke = band_decision<false>(d[i], il[i], n2[i], truncdist, se, be);
    kindFast[i] = kf | (bf ? 16 : 0);
    kindExact[i] = ke | (be ? 16 : 0);
    sampFast[i] = sf;
    sampExact[i] = se;
}

}  // namespace
}  // namespace emf_hip

extern "C" {

int emf_hip_sweepFastPathPremises(unsigned long long* out4_dev, emf_stream_t stream) {
    using namespace emf_hip;
    if (!out4_dev) return fail(EMF_E_NULL, "sweepFastPathPremises: out is NULL");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const hipError_t e = hipMemsetAsync(out4_dev, 0, 4 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return fail(static_cast<int>(e), "sweepFastPathPremises: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_sweep_premises, dim3(8192), dim3(256), 0, st, out4_dev);
    return launch_status("sweepFastPathPremises");
}

int emf_hip_debugPixelRounding(const float* num_dev, const float* den_dev, int n, int32_t* fast_dev, int32_t* exact_dev,
                               emf_stream_t stream) {
    using namespace emf_hip;
    if (!num_dev || !den_dev || !fast_dev || !exact_dev) return fail(EMF_E_NULL, "debugPixelRounding: NULL buffer");
    if (n <= 0) return fail(EMF_E_SHAPE, "debugPixelRounding: n = %d", n);
    hipLaunchKernelGGL(k_debug_pixel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), num_dev,
                       den_dev, n, fast_dev, exact_dev);
    return launch_status("debugPixelRounding");
}

int emf_hip_debugBandDecision(const float* d_dev, const float* invLambda_dev, const float* n2_dev, int n, float truncdist,
                              int32_t* kindFast_dev, float* sampleFast_dev, int32_t* kindExact_dev, float* sampleExact_dev,
                              emf_stream_t stream) {
    using namespace emf_hip;
    if (!d_dev || !invLambda_dev || !n2_dev || !kindFast_dev || !sampleFast_dev || !kindExact_dev || !sampleExact_dev)
        return fail(EMF_E_NULL, "debugBandDecision: NULL buffer");
    if (n <= 0) return fail(EMF_E_SHAPE, "debugBandDecision: n = %d", n);
    if (!(truncdist > 0.f)) return fail(EMF_E_ARG, "debugBandDecision: truncdist %g", truncdist);
    hipLaunchKernelGGL(k_debug_band, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_dev,
                       invLambda_dev, n2_dev, n, truncdist, kindFast_dev, sampleFast_dev, kindExact_dev, sampleExact_dev);
    return launch_status("debugBandDecision");
}

}  // extern "C"
