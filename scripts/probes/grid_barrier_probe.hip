// What one grid-wide exchange costs INSIDE a kernel on MI355X (8 XCDs, L2s not coherent with each other) when it is
// built from write-through stores and relaxed agent-scope atomics instead of release / acquire fences -- the question
// behind a persistent Levenberg-Marquardt kernel (tracking.hip): every workgroup stores a row of partial sums, the
// last one to arrive adds them up, does the scalar step and publishes the new state, everybody reads it.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/grid_barrier_probe.hip -o /tmp/grid_barrier_probe
// Modes: 0 barrier only (ticket + generation word); 1 + rows of 30 floats, reduced by the last arriver, 128 floats
// of state published; 2 like 1 with __threadfence() release / acquire instead of the write-through stores (the
// round-1 experiment, for the record); 3 like 1, every workgroup reduces all rows itself (no second wait).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Sync {
    unsigned ticket;
    unsigned pad0[31];
    unsigned gen;
    unsigned pad1[31];
    unsigned error;
};

constexpr int kCols = 30, kState = 128;

template <class T>
__device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(1024) void k_persist(Sync* s, float* rows, float* state, float* sink, int iters, int mode,
                                                  int bodySleep, long long* stamps, int stride = 1) {
    // stride 8 (round 6): only every 8th workgroup takes part -- blocks b with b % 8 == 0 run on ONE XCD (dispatch order), so the
    // exchange stays inside one L2: the question behind a persistent kernel for OBJECT tracking stages (few rows of pixels)
    if (blockIdx.x % stride) return;
    const int bid = blockIdx.x / stride;
    __shared__ int s_last;
    __shared__ float red[kCols];
    __shared__ float st[kState];
    const int tid = threadIdx.x, nwg = gridDim.x / stride;
    const long long t0 = wall_clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // body stand-in
        for (int k = 0; k < bodySleep; ++k) __builtin_amdgcn_s_sleep(64);
        if (mode >= 1) {
            if (tid < kCols) {
                const float v = static_cast<float>(bid + it + tid);
                float* dst = rows + (static_cast<size_t>(it & 1) * kCols + tid) * nwg + bid;
                if (mode == 2) *dst = v; else st_agent(dst, v);
            }
            if (mode == 2) __threadfence();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&s->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == static_cast<unsigned>(nwg) * (it + 1) - 1u;
        }
        __syncthreads();
        const bool last = s_last != 0;
        if (mode == 3) {
            // everybody waits for the ticket to be full, then reduces all the rows itself
            if (tid == 0) {
                const long long w0 = wall_clock64();
                while (ld_agent(&s->ticket) < static_cast<unsigned>(nwg) * (it + 1)) {
                    if (wall_clock64() - w0 > 100000000ll) { st_agent(&s->error, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63;
            for (int c = wave; c < kCols; c += 16) {
                const float* col = rows + (static_cast<size_t>(it & 1) * kCols + c) * nwg;
                float a = 0.f;
                for (int i = lane; i < nwg; i += 64) a += ld_agent(col + i);
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (lane == 0) red[c] = a;
            }
            __syncthreads();
            acc += red[tid % kCols];
            continue;
        }
        if (last) {
            if (mode >= 1) {
                if (mode == 2) __threadfence();
                const int wave = tid >> 6, lane = tid & 63;
                for (int c = wave; c < kCols; c += 16) {
                    const float* col = rows + (static_cast<size_t>(it & 1) * kCols + c) * nwg;
                    float a = 0.f;
                    for (int i = lane; i < nwg; i += 64) a += mode == 2 ? col[i] : ld_agent(col + i);
                    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                    if (lane == 0) red[c] = a;
                }
                __syncthreads();
                if (tid < kState) {
                    const float v = red[tid % kCols] + tid;
                    float* dst = state + (it & 1) * kState + tid;
                    if (mode == 2) *dst = v; else st_agent(dst, v);
                }
                if (mode == 2) __threadfence();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (tid == 0) st_agent(&s->gen, static_cast<unsigned>(it + 1));
        } else if (tid == 0) {
            const long long w0 = wall_clock64();
            while (ld_agent(&s->gen) < static_cast<unsigned>(it + 1)) {
                if (wall_clock64() - w0 > 100000000ll) { st_agent(&s->error, 1u); break; }  // 1 s: never hang the box
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (mode >= 1) {
            if (mode == 2) __threadfence();
            if (tid < kState) st[tid] = mode == 2 ? state[(it & 1) * kState + tid] : ld_agent(state + (it & 1) * kState + tid);
            __syncthreads();
            acc += st[tid & (kState - 1)];
        }
    }
    const long long t1 = wall_clock64();
    if (tid == 0) stamps[bid] = t1 - t0;
    sink[bid * 1024 + tid] = acc;
}

// the same work as one launch per iteration (what the product does today), for the launch-gap figure
__global__ __launch_bounds__(1024) void k_one(float* rows, float* state, float* sink, int it, int bodySleep) {
    __shared__ float red[kCols];
    const int tid = threadIdx.x, nwg = gridDim.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int c = wave; c < kCols; c += 16) {
        const float* col = rows + (static_cast<size_t>((it + 1) & 1) * kCols + c) * nwg;
        float a = 0.f;
        for (int i = lane; i < nwg; i += 64) a += col[i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) red[c] = a;
    }
    __syncthreads();
    for (int k = 0; k < bodySleep; ++k) __builtin_amdgcn_s_sleep(64);
    if (tid < kCols) rows[(static_cast<size_t>(it & 1) * kCols + tid) * nwg + blockIdx.x] = red[tid] + blockIdx.x;
    sink[blockIdx.x * 1024 + tid] = red[tid % kCols];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    Sync* s; float *rows, *state, *sink; long long* stamps;
    hipMalloc(&s, sizeof(Sync)); hipMalloc(&rows, 2 * kCols * 1024 * 4); hipMalloc(&state, 2 * kState * 4);
    hipMalloc(&sink, 1024 * 1024 * 4); hipMalloc(&stamps, 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 2) {  // round 6: few workgroups, spread over the XCDs (stride 1) or on one XCD (stride 8)
        for (int stride : {1, 8})
            for (int nwg : {2, 4, 8, 16, 32})
                for (int body : {0, 2})
                    for (int mode : {0, 1, 3}) {
                        float best = 1e30f; unsigned err = 0;
                        for (int rep = 0; rep < 3; ++rep) {
                            hipMemset(s, 0, sizeof(Sync));
                            hipEventRecord(e0, 0);
                            hipLaunchKernelGGL(k_persist, dim3(nwg * stride), dim3(1024), 0, 0, s, rows, state, sink, iters, mode, body, stamps, stride);
                            hipEventRecord(e1, 0);
                            hipEventSynchronize(e1);
                            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
                            Sync h; hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost); err |= h.error;
                        }
                        printf("persistent stride %d nwg %2d body %d mode %d: %.2f us per iteration%s\n", stride, nwg, body, mode,
                               1e3 * best / iters, err ? "  TIMEOUT" : "");
                    }
        return 0;
    }
    for (int nwg : {64, 128, 240, 256}) {
        for (int body : {0, 8}) {
            for (int mode : {0, 1, 3, 2}) {
                if (mode == 2 && nwg != 240) continue;
                float best = 1e30f; unsigned err = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(s, 0, sizeof(Sync));
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(k_persist, dim3(nwg), dim3(1024), 0, 0, s, rows, state, sink, iters, mode, body, stamps);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
                    Sync h; hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost); err |= h.error;
                }
                std::vector<long long> st(nwg); hipMemcpy(st.data(), stamps, nwg * 8, hipMemcpyDeviceToHost);
                const long long mx = *std::max_element(st.begin(), st.end());
                printf("persistent nwg %3d body %d mode %d: %.2f us per iteration (in-kernel %.2f)%s\n", nwg, body, mode,
                       1e3 * best / iters, mx / 100.0 / iters, err ? "  TIMEOUT" : "");
            }
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                for (int it = 0; it < iters; ++it)
                    hipLaunchKernelGGL(k_one, dim3(nwg), dim3(1024), 0, 0, rows, state, sink, it, body);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("launch per iteration nwg %3d body %d: %.2f us per iteration\n", nwg, body, 1e3 * best / iters);
        }
    }
    return 0;
}
