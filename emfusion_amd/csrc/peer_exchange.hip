// peer_exchange.hip -- one-shot exchanges by direct peer writes ("Collective
// implementation": the messages of this path are 1-5 MB, so hops -- latency -- decide, not the per-link bound;
// a ring is the wrong shape).  The reference is single-GPU: nothing to cite, this is new design.
//
// Every rank owns a receive buffer of `world` slots (one per sender, twice: two parities) and `world` flag
// words; every rank can address every peer's buffer and flags (same process: plain pointers; one process per
// GPU: hipIpc mappings over xGMI, made once by the host, core/Communicator.cpp).  An exchange with sequence
// number s is three small launches on the caller's stream, no host involvement:
//   1. scatter      my contribution -> slot[me] (parity s & 1) of EVERY peer, 16-byte stores, all links at once
//   2. signal+wait  flag[me] := s on every peer (system-scope release), then spin until my own flags of all
//                   peers show >= s (system-scope acquire); bounded: a time-out sets the error word
//   3. reduce/copy  sum / min the world slots locally IN RANK ORDER -- the same bits on every rank -- or copy
//                   the sender's slot (broadcast, band gather)
// Parities: slot halves alternate with s.  A sender may overwrite parity s & 1 again in exchange s + 2 only
// after it has seen every peer's flag >= s + 1, and a peer raises that flag behind its own reduce of exchange
// s (stream order): nobody is still reading what is overwritten.  Every exchange signals and waits all-to-all,
// also a broadcast, so that this argument needs no case analysis.
// Untested on xGMI (no multi-GPU box in the build environment): exercised with one rank, with 2-4 ranks on
// threads of one process and with 2-3 ranks in separate processes over hipIpc, all sharing one GPU.
#include "common.hpp"

namespace emf_hip {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

struct PeerArgs {
    char* slots[EMF_MAX_PEERS];
    uint32_t* flags[EMF_MAX_PEERS];
    int rank, world;
    size_t slotBytes;  // one sender's slot, one parity
    uint32_t* error;
};

__device__ __forceinline__ size_t slot_offset(const PeerArgs& a, int sender, uint32_t seq) {
    return (static_cast<size_t>(seq & 1u) * a.world + sender) * a.slotBytes;
}

// src[0, bytes) -> slot[me] + dstOffset of every peer; bytes, offsets and pointers are multiples of 16
__global__ __launch_bounds__(256) void k_peer_scatter(PeerArgs a, const char* __restrict__ src, size_t bytes,
                                                      size_t dstOffset, uint32_t seq) {
    const size_t n16 = bytes / 16;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t off = slot_offset(a, a.rank, seq) + dstOffset;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const f4v v = *reinterpret_cast<const f4v*>(src + 16 * i);
        for (int p = 0; p < a.world; ++p)  // write-through: the data must be in the peer's memory, not in my L2
            __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(a.slots[p] + off + 16 * i));
    }
    __threadfence_system();
}

// one wave: lane p signals peer p, then waits for peer p's signal
__global__ void k_peer_signal_wait(PeerArgs a, uint32_t seq, unsigned long long timeoutTicks) {
    const int p = threadIdx.x;
    if (p >= a.world) return;
    __threadfence_system();  // everything this stream stored before (the scatter) is visible first
    __hip_atomic_store(a.flags[p] + a.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = a.flags[a.rank] + p;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t seen = __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (static_cast<int32_t>(seen - seq) >= 0) break;  // (wrap-around safe)
        if (wall_clock64() - t0 > timeoutTicks) {
            __hip_atomic_store(a.error, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

__device__ __forceinline__ f4v load_slot16(const char* p) {  // what a peer wrote: not through a stale cache line
    return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
}

__global__ __launch_bounds__(256) void k_peer_reduce_sum_f32(PeerArgs a, uint32_t seq, size_t count, float* __restrict__ out) {
    const size_t n4 = count / 4;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const char* base = a.slots[a.rank];
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4v acc = load_slot16(base + slot_offset(a, 0, seq) + 16 * i);
        for (int r = 1; r < a.world; ++r) {  // rank order: ((s0 + s1) + s2) + ... on every rank alike
            const f4v v = load_slot16(base + slot_offset(a, r, seq) + 16 * i);
            acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
        }
        *reinterpret_cast<f4v*>(out + 4 * i) = acc;
    }
}

__global__ __launch_bounds__(256) void k_peer_reduce_min_u64(PeerArgs a, uint32_t seq, size_t count,
                                                             unsigned long long* __restrict__ out) {
    typedef unsigned long long u2v __attribute__((ext_vector_type(2)));
    const size_t n2 = count / 2;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const char* base = a.slots[a.rank];
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += stride) {
        u2v acc = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(base + slot_offset(a, 0, seq) + 16 * i));
        for (int r = 1; r < a.world; ++r) {
            const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(base + slot_offset(a, r, seq) + 16 * i));
            acc.x = v.x < acc.x ? v.x : acc.x;
            acc.y = v.y < acc.y ? v.y : acc.y;
        }
        *reinterpret_cast<u2v*>(out + 2 * i) = acc;
    }
}

__global__ __launch_bounds__(256) void k_peer_copy_from_slot(PeerArgs a, uint32_t seq, int sender, size_t srcOffset,
                                                             char* __restrict__ dst, size_t bytes) {
    const size_t n16 = bytes / 16;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const char* src = a.slots[a.rank] + slot_offset(a, sender, seq) + srcOffset;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
        *reinterpret_cast<f4v*>(dst + 16 * i) = load_slot16(src + 16 * i);
}

int to_args(const emf_peer_t* g, PeerArgs& a, const char* who) {
    if (!g) return fail(EMF_E_NULL, "%s: peer group is NULL", who);
    if (g->world < 1 || g->world > EMF_MAX_PEERS || g->rank < 0 || g->rank >= g->world)
        return fail(EMF_E_LIMIT, "%s: rank %d of %d (at most %d peers)", who, g->rank, g->world, EMF_MAX_PEERS);
    if (!g->error || g->slotBytes == 0 || g->slotBytes % 16) return fail(EMF_E_ARG, "%s: bad slot size / error word", who);
    for (int p = 0; p < g->world; ++p) {
        if (!g->slots[p] || !g->flags[p]) return fail(EMF_E_NULL, "%s: peer %d is not mapped", who, p);
        a.slots[p] = static_cast<char*>(g->slots[p]);
        a.flags[p] = g->flags[p];
    }
    a.rank = g->rank;
    a.world = g->world;
    a.slotBytes = g->slotBytes;
    a.error = g->error;
    return EMF_OK;
}

unsigned grid_for(size_t units) {  // 16-byte units; a few per lane
    const size_t blocks = units / 256 / 4 + 1;
    return static_cast<unsigned>(blocks < 2048 ? blocks : 2048);
}

bool aligned16(const void* p, size_t a, size_t b) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && a % 16 == 0 && b % 16 == 0;
}

}  // namespace
}  // namespace emf_hip

extern "C" {

size_t emf_hip_peerBufferBytes(int world, size_t slotBytes) { return 2 * static_cast<size_t>(world) * slotBytes; }

int emf_hip_peerScatter(const emf_peer_t* group, const void* src, size_t bytes, size_t dstOffset, uint32_t seq,
                        emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = to_args(group, a, "peerScatter")) return rc;
    if (!src) return fail(EMF_E_NULL, "peerScatter: src is NULL");
    if (!aligned16(src, bytes, dstOffset) || dstOffset + bytes > a.slotBytes)
        return fail(EMF_E_ARG, "peerScatter: %zu bytes at offset %zu do not fit a %zu-byte slot in 16-byte units", bytes,
                    dstOffset, a.slotBytes);
    if (bytes == 0) return EMF_OK;
    hipLaunchKernelGGL(k_peer_scatter, dim3(grid_for(bytes / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a,
                       static_cast<const char*>(src), bytes, dstOffset, seq);
    return launch_status("peerScatter");
}

int emf_hip_peerSignalWait(const emf_peer_t* group, uint32_t seq, uint32_t timeoutMs, emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = to_args(group, a, "peerSignalWait")) return rc;
    hipLaunchKernelGGL(k_peer_signal_wait, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, seq,
                       static_cast<unsigned long long>(timeoutMs) * 100000ull);
    return launch_status("peerSignalWait");
}

int emf_hip_peerReduceSumF32(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = to_args(group, a, "peerReduceSumF32")) return rc;
    if (!out) return fail(EMF_E_NULL, "peerReduceSumF32: out is NULL");
    if (count % 4 || count * 4 > a.slotBytes || reinterpret_cast<uintptr_t>(out) % 16)
        return fail(EMF_E_ARG, "peerReduceSumF32: count %zu (multiple of 4, at most %zu)", count, a.slotBytes / 4);
    if (count == 0) return EMF_OK;
    hipLaunchKernelGGL(k_peer_reduce_sum_f32, dim3(grid_for(count / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, count, out);
    return launch_status("peerReduceSumF32");
}

int emf_hip_peerReduceMinU64(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = to_args(group, a, "peerReduceMinU64")) return rc;
    if (!out) return fail(EMF_E_NULL, "peerReduceMinU64: out is NULL");
    if (count % 2 || count * 8 > a.slotBytes || reinterpret_cast<uintptr_t>(out) % 16)
        return fail(EMF_E_ARG, "peerReduceMinU64: count %zu (even, at most %zu)", count, a.slotBytes / 8);
    if (count == 0) return EMF_OK;
    hipLaunchKernelGGL(k_peer_reduce_min_u64, dim3(grid_for(count / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, count, reinterpret_cast<unsigned long long*>(out));
    return launch_status("peerReduceMinU64");
}

int emf_hip_peerCopyFromSlot(const emf_peer_t* group, uint32_t seq, int sender, size_t srcOffset, void* dst, size_t bytes,
                             emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = to_args(group, a, "peerCopyFromSlot")) return rc;
    if (!dst) return fail(EMF_E_NULL, "peerCopyFromSlot: dst is NULL");
    if (sender < 0 || sender >= a.world || !aligned16(dst, bytes, srcOffset) || srcOffset + bytes > a.slotBytes)
        return fail(EMF_E_ARG, "peerCopyFromSlot: sender %d, %zu bytes at offset %zu", sender, bytes, srcOffset);
    if (bytes == 0) return EMF_OK;
    hipLaunchKernelGGL(k_peer_copy_from_slot, dim3(grid_for(bytes / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, sender, srcOffset, static_cast<char*>(dst), bytes);
    return launch_status("peerCopyFromSlot");
}

}  // extern "C"
