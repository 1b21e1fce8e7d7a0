// abi_common.hip -- error plumbing, argument validation and device queries shared by every
// emf_hip_* entry point.
#include "common.hpp"

#include <mutex>
#include <vector>

namespace emf_hip {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_image(const emf_image_t* im, size_t elem_bytes, const char* name) {
    if (im == nullptr) return fail(EMF_E_NULL, "%s: image view is NULL", name);
    if (im->data == nullptr) return fail(EMF_E_NULL, "%s: data pointer is NULL", name);
    if (im->width <= 0 || im->height <= 0)
        return fail(EMF_E_SHAPE, "%s: bad size %d x %d", name, im->width, im->height);
    if (im->pitch < static_cast<size_t>(im->width) * elem_bytes)
        return fail(EMF_E_PITCH, "%s: pitch %zu < row bytes %zu", name, im->pitch,
                    static_cast<size_t>(im->width) * elem_bytes);
    const size_t align = elem_bytes % 4 == 0 ? 4 : 1;
    if (im->pitch % align != 0 || reinterpret_cast<uintptr_t>(im->data) % align != 0)
        return fail(EMF_E_PITCH, "%s: pitch/base not %zu-byte aligned", name, align);
    return EMF_OK;
}

int check_same_size(const emf_image_t* a, const emf_image_t* b, const char* an, const char* bn) {
    if (a->width != b->width || a->height != b->height)
        return fail(EMF_E_SHAPE, "%s is %d x %d but %s is %d x %d", an, a->width, a->height, bn,
                    b->width, b->height);
    return EMF_OK;
}

int check_res(const int32_t res[3]) {
    if (res == nullptr) return fail(EMF_E_NULL, "res is NULL");
    if (res[0] < 2 || res[1] < 2 || res[2] < 2)
        return fail(EMF_E_SHAPE, "volume resolution %d x %d x %d (each axis needs >= 2)", res[0],
                    res[1], res[2]);
    // index arithmetic is size_t on the device; keep the voxel count addressable with 4-byte elems
    const unsigned long long n =
        static_cast<unsigned long long>(res[0]) * res[1] * static_cast<unsigned long long>(res[2]);
    if (n > (1ull << 36)) return fail(EMF_E_LIMIT, "volume of %llu voxels exceeds 2^36", n);
    return EMF_OK;
}

int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return EMF_OK;
}

// Comparison of x / d with its reciprocal form (march_wave.hpp div_voxel); counts disagreements inside the range the
// march can produce, 1e-30 <= |x| <= 1e30.
//
// FULL: all 2^32 float bit patterns (2.3 ms of the whole chip) -- the form rounds 1-5 ran per distinct voxel size; kept
// as emf_hip_voxelReciprocalExhaustive for the test that the short form below gives the same verdict.
//
// Short form (default): every mantissa of the binade [1, 2) with both signs, and every mantissa of the binade that holds
// 1e-30 (positive inputs; clipped by the guard exactly as the full form clips it) -- 3 x 2^23 inputs, ~20 us.  Why that
// decides all binades of the guarded range: with r = fl(1 / d),
//     q0 = fl(x r),  t = fma(-q0, d, x),  q = fma(t, r, q0),  D = fl(x / d),
// and x' = 2^k x also inside the range,
//   * q0' = 2^k q0 and D' = 2^k D: with 1e-6 <= d <= 1e3 (rcp_check_size) both stay in [1e-33, 1e36], normal floats, and
//     round-to-nearest commutes with a power-of-two scaling of a normal result;
//   * t is EXACT: x - q0 d is a multiple of ulp(q0) ulp(d) >= 2^(e_x - 47) of magnitude <= 2^(e_x - 22), i.e. it has at
//     most 24 significant bits and, for e_x >= -100, a granularity >= 2^-147 > 2^-149 -- representable, as a subnormal
//     below 2^-126.  So t' = 2^k t with no rounding at all, PROVIDED the fma keeps subnormal results and inputs (the
//     premise k_check_reciprocal verifies on the device before it counts anything: a flushing mode fails the check);
//   * q = fl(t r + q0) is normal again, so q' = 2^k q.
// Hence q' == D' iff q == D: one binade decides.  Negating x negates q0, t, q and D alike (IEEE multiplication, fma and
// division are sign-symmetric under round-to-nearest-even), so one sign decides too -- the central binade is swept with
// both anyway.  The lowest binade of the range, where t IS subnormal and the premise is what carries the argument, is
// swept as well, so that regime is exercised, not only argued; at the upper end nothing changes (q <= 1e36 cannot
// overflow).  tests/test_gpu_parity.py compares this form's verdict with the full form's over > 1300 voxel sizes.
template <bool FULL>
__global__ __launch_bounds__(256) void k_check_reciprocal(float d, float rcp,
                                                          unsigned long long* mismatches) {
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    const unsigned long long first = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned bad = 0;
    if (!FULL && first == 0) {
        // the premise: v_fma_f32 neither flushes a subnormal result nor a subnormal input (operands formed from
        // blockDim so that nothing is folded at compile time)
        const float one = static_cast<float>(blockDim.x) * (1.f / 256.f);
        const float a = __uint_as_float(0x0d800001u) * one;   // 2^-100 (1 + 2^-23)
        const float b = __uint_as_float(0x3f800001u) * one;   // 1 + 2^-23
        const float c = __uint_as_float(0x0d800002u) * one;   // 2^-100 (1 + 2^-22)
        const float t = __builtin_fmaf(-a, b, c);             // exactly -2^-146
        bad += __float_as_uint(t) != 0x80000008u;
        const float u = __builtin_fmaf(t, 1048576.f * one, 0.f * one);  // -2^-126
        bad += __float_as_uint(u) != 0x80800000u;
    }
    const unsigned long long count = FULL ? (1ull << 32) : (3ull << 23);
    for (unsigned long long i = first; i < count; i += stride) {
        unsigned bits = static_cast<unsigned>(i);
        if (!FULL) {
            const unsigned region = static_cast<unsigned>(i >> 23);         // 0: [1, 2)   1: -[1, 2)   2: [2^-100, 2^-99)
            bits = (region == 1 ? 0x80000000u : 0u) | ((region == 2 ? 27u : 127u) << 23) | (static_cast<unsigned>(i) & 0x7fffffu);
        }
        const float x = __uint_as_float(bits);
        const float ax = fabsf(x);
        if (!(ax >= 1e-30f && ax <= 1e30f)) continue;  // also skips NaN
        const float q0 = x * rcp;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, d, x), rcp, q0);
        bad += __float_as_uint(q) != __float_as_uint(x / d);
    }
    if (bad) atomicAdd(mismatches, static_cast<unsigned long long>(bad));
}
constexpr unsigned kRcpShortBlocks = 2048;  // x 256 lanes: 48 inputs per lane

__global__ void k_check_reciprocal_begin(unsigned long long* mismatches) {
    if (threadIdx.x == 0) *mismatches = 0ull;
}

// One wave that stays resident until the host sets *release (device-visible host memory) or `ticks` of
// the 100 MHz wall clock have passed: while it runs, its stream is "not ready" -- a probe for whether a
// host call synchronised with the whole device (tests/test_gpu_dynamic_objects.py).
__global__ void k_spin_probe(const volatile uint32_t* release, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        if (__hip_atomic_load(const_cast<const uint32_t*>(release), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

// keeps its stream busy for `ticks` of the 100 MHz wall clock (Communicator.cpp: latency model)
__global__ void k_spin_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Calibration of the vector L1's gather rate (bench.py: the peak `roofline.frac` is quoted against when the L1 binds
// a kernel).  Every wave issues `iters` independent 8-byte loads -- the march's gathers are 8-byte pair loads -- from
// a footprint that stays resident in every CU's 32 KiB L1, at full occupancy; LINES = distinct 128-byte lines the 64
// lanes of one instruction touch: 64 (a line per lane), 4 (64 consecutive words) or 1 (16 words, four lanes each).
template <int LINES>
__global__ __launch_bounds__(256) void k_l1_probe(const unsigned long long* __restrict__ buf, unsigned words, int iters,
                                                  unsigned long long* __restrict__ sink) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned mask = words - 1u;  // words: a power of two
    unsigned idx = LINES == 64 ? lane * 16u : (LINES == 4 ? lane : (lane & 15u));
    idx += wave * 16u;
    const unsigned step = LINES == 4 ? 64u : 16u;
    unsigned long long acc = 0;
#pragma unroll 8
    for (int i = 0; i < iters; ++i) acc ^= buf[(idx + static_cast<unsigned>(i) * step) & mask];
    if (acc == 0x0123456789abcdefull) *sink = acc;  // (never: keeps the loads alive)
}

// device-to-device stream copy, 16 bytes per lane and iteration (the bandwidth yardstick of bench.py)
__global__ __launch_bounds__(256) void k_stream_copy(float4* __restrict__ dst,
                                                     const float4* __restrict__ src, size_t n16) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
        dst[i] = src[i];
}

}  // namespace emf_hip

extern "C" {

int emf_hip_streamCopy(void* dst, const void* src, size_t bytes, emf_stream_t stream) {
    using namespace emf_hip;
    if (!dst || !src) return fail(EMF_E_NULL, "streamCopy: NULL buffer");
    if (bytes % 16 || (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16)
        return fail(EMF_E_ARG, "streamCopy: buffers and size must be multiples of 16 bytes");
    const size_t n16 = bytes / 16;
    if (n16 == 0) return EMF_OK;
    const size_t blocks = n16 / 256 / 4 + 1;  // ~4 iterations per lane
    hipLaunchKernelGGL(k_stream_copy, dim3(static_cast<unsigned>(blocks < 65535u * 16u ? blocks : 65535u * 16u)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<float4*>(dst), static_cast<const float4*>(src), n16);
    return launch_status("streamCopy");
}

// ---- checked reciprocal: process-wide cache + asynchronous check ---------------------------------
// The check is a property of the voxel size's bit pattern alone, so its verdict is kept for the life
// of the process: a volume created with a size seen before costs a table look-up, no device work.
namespace {
struct RcpVerdict {
    uint32_t sizeBits;
    float rcp;  // 1 / voxelSize if all 2^32 inputs agreed, 0 otherwise
};
uint32_t rcp_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, sizeof(b));
    return b;
}
std::mutex g_rcpMutex;
std::vector<RcpVerdict> g_rcpVerdicts;
// The blocking form runs ONE check at a time (g_rcpCheckMutex is held from the memset to the read-back):
// concurrent first-time callers -- one host thread per rank in the rehearsal -- would otherwise clear or read
// each other's mismatch count, and a wrong verdict would be cached for the life of the process.
std::mutex g_rcpCheckMutex;
std::vector<std::pair<int, unsigned long long*>> g_rcpCounters;  // one device word per device, allocated once

bool rcp_lookup(float voxelSize, float* rcp) {
    const uint32_t bits = rcp_bits(voxelSize);
    std::lock_guard<std::mutex> lock(g_rcpMutex);
    for (const RcpVerdict& v : g_rcpVerdicts)
        if (v.sizeBits == bits) {
            *rcp = v.rcp;
            return true;
        }
    return false;
}
void rcp_remember(float voxelSize, float rcp) {
    const uint32_t bits = rcp_bits(voxelSize);
    std::lock_guard<std::mutex> lock(g_rcpMutex);
    for (const RcpVerdict& v : g_rcpVerdicts)
        if (v.sizeBits == bits) return;
    g_rcpVerdicts.push_back(RcpVerdict{bits, rcp});
}
int rcp_check_size(float voxelSize, const char* who) {
    if (!(voxelSize >= 1e-6f && voxelSize <= 1e3f))  // quotients of the checked range stay finite
        return emf_hip::fail(EMF_E_ARG, "%s: voxelSize %g outside [1e-6, 1e3]", who, voxelSize);
    return EMF_OK;
}
}  // namespace

int emf_hip_voxelReciprocal(float voxelSize, float* rcp) {
    using namespace emf_hip;
    if (!rcp) return fail(EMF_E_NULL, "voxelReciprocal: rcp is NULL");
    *rcp = 0.f;
    if (const int rc = rcp_check_size(voxelSize, "voxelReciprocal")) return rc;
    if (rcp_lookup(voxelSize, rcp)) return EMF_OK;  // seen before: no device work
    std::lock_guard<std::mutex> check(g_rcpCheckMutex);
    if (rcp_lookup(voxelSize, rcp)) return EMF_OK;  // another caller finished the same check meanwhile
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    unsigned long long* counter = nullptr;
    for (const auto& c : g_rcpCounters)
        if (c.first == dev) counter = c.second;
    if (e == hipSuccess && !counter) {
        e = hipMalloc(reinterpret_cast<void**>(&counter), sizeof(unsigned long long));
        if (e == hipSuccess) g_rcpCounters.emplace_back(dev, counter);
    }
    // its own stream: the wait below is for this check alone, not for the device
    hipStream_t st = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned long long bad = 1;
    if (e == hipSuccess) e = hipMemsetAsync(counter, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        const float r = 1.0f / voxelSize;
        hipLaunchKernelGGL(k_check_reciprocal<false>, dim3(kRcpShortBlocks), dim3(256), 0, st, voxelSize, r, counter);
        e = hipMemcpyAsync(&bad, counter, sizeof(bad), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) {
            *rcp = bad == 0 ? r : 0.f;
            rcp_remember(voxelSize, *rcp);
        }
    }
    if (st) (void)hipStreamDestroy(st);
    if (e != hipSuccess) {
        set_error("voxelReciprocal: %s", hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return EMF_OK;
}

int emf_hip_voxelReciprocalCached(float voxelSize, float* rcp) {
    using namespace emf_hip;
    if (!rcp) return fail(EMF_E_NULL, "voxelReciprocalCached: rcp is NULL");
    *rcp = 0.f;
    if (const int rc = rcp_check_size(voxelSize, "voxelReciprocalCached")) return rc;
    return rcp_lookup(voxelSize, rcp) ? EMF_OK : EMF_E_NOTREADY;
}

int emf_hip_voxelReciprocalBegin(float voxelSize, unsigned long long* mismatches, emf_stream_t stream) {
    using namespace emf_hip;
    if (!mismatches) return fail(EMF_E_NULL, "voxelReciprocalBegin: mismatches is NULL");
    if (const int rc = rcp_check_size(voxelSize, "voxelReciprocalBegin")) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_check_reciprocal_begin, dim3(1), dim3(64), 0, st, mismatches);
    hipLaunchKernelGGL(k_check_reciprocal<false>, dim3(kRcpShortBlocks), dim3(256), 0, st, voxelSize, 1.0f / voxelSize, mismatches);
    return launch_status("voxelReciprocalBegin");
}

int emf_hip_voxelReciprocalEnd(float voxelSize, unsigned long long mismatches, float* rcp) {
    using namespace emf_hip;
    if (!rcp) return fail(EMF_E_NULL, "voxelReciprocalEnd: rcp is NULL");
    *rcp = 0.f;
    if (const int rc = rcp_check_size(voxelSize, "voxelReciprocalEnd")) return rc;
    *rcp = mismatches == 0 ? 1.0f / voxelSize : 0.f;
    rcp_remember(voxelSize, *rcp);
    return EMF_OK;
}

int emf_hip_voxelReciprocalExhaustive(float voxelSize, unsigned long long* mismatches_host) {
    using namespace emf_hip;
    if (!mismatches_host) return fail(EMF_E_NULL, "voxelReciprocalExhaustive: mismatches_host is NULL");
    if (const int rc = rcp_check_size(voxelSize, "voxelReciprocalExhaustive")) return rc;
    unsigned long long* counter = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&counter), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(counter, 0, sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_check_reciprocal<true>, dim3(8192), dim3(256), 0, nullptr, voxelSize, 1.0f / voxelSize, counter);
        e = hipMemcpy(mismatches_host, counter, sizeof(unsigned long long), hipMemcpyDeviceToHost);
    }
    if (counter) (void)hipFree(counter);
    if (e != hipSuccess) {
        set_error("voxelReciprocalExhaustive: %s", hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return EMF_OK;
}

int emf_hip_l1GatherProbe(const void* buf, size_t footprintBytes, int linesPerInstruction, int iterations, int workgroups,
                          void* sink, emf_stream_t stream) {
    using namespace emf_hip;
    if (!buf || !sink) return fail(EMF_E_NULL, "l1GatherProbe: NULL buffer");
    if (footprintBytes < 8192 || (footprintBytes & (footprintBytes - 1)) || footprintBytes > (size_t(1) << 30))
        return fail(EMF_E_ARG, "l1GatherProbe: footprint %zu (a power of two, 8 KiB .. 1 GiB)", footprintBytes);
    if (iterations < 1 || workgroups < 1 || workgroups > 65535 * 16) return fail(EMF_E_ARG, "l1GatherProbe: iterations / workgroups");
    const unsigned words = static_cast<unsigned>(footprintBytes / 8);
    auto* b = static_cast<const unsigned long long*>(buf);
    auto* sk = static_cast<unsigned long long*>(sink);
    const dim3 g(static_cast<unsigned>(workgroups)), blk(256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (linesPerInstruction == 64) hipLaunchKernelGGL(k_l1_probe<64>, g, blk, 0, st, b, words, iterations, sk);
    else if (linesPerInstruction == 4) hipLaunchKernelGGL(k_l1_probe<4>, g, blk, 0, st, b, words, iterations, sk);
    else if (linesPerInstruction == 1) hipLaunchKernelGGL(k_l1_probe<1>, g, blk, 0, st, b, words, iterations, sk);
    else return fail(EMF_E_ARG, "l1GatherProbe: linesPerInstruction %d (64, 4 or 1)", linesPerInstruction);
    return launch_status("l1GatherProbe");
}

int emf_hip_spinDelay(uint32_t microseconds, emf_stream_t stream) {
    using namespace emf_hip;
    if (microseconds > 1000000u) return fail(EMF_E_ARG, "spinDelay: %u us", microseconds);
    hipLaunchKernelGGL(k_spin_delay, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<unsigned long long>(microseconds) * 100ull);
    return launch_status("spinDelay");
}

int emf_hip_spinProbe(const volatile uint32_t* release, uint32_t maxMilliseconds, emf_stream_t stream) {
    using namespace emf_hip;
    if (!release) return fail(EMF_E_NULL, "spinProbe: release is NULL");
    hipLaunchKernelGGL(k_spin_probe, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), release,
                       static_cast<unsigned long long>(maxMilliseconds) * 100000ull);
    return launch_status("spinProbe");
}

int emf_hip_abi_version(void) { return EMF_HIP_ABI_VERSION; }

const char* emf_hip_last_error_string(void) { return emf_hip::g_err; }

int emf_hip_device_info(char* name, size_t name_len, char* arch, size_t arch_len, int* num_cus) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return emf_hip::fail(EMF_E_NODEVICE, "no HIP device visible");
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return emf_hip::fail(EMF_E_NODEVICE, "cannot query HIP device");
    }
    if (name && name_len) snprintf(name, name_len, "%s", prop.name);
    if (arch && arch_len) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    if (num_cus) *num_cus = prop.multiProcessorCount;
    return EMF_OK;
}

}  // extern "C"
