// peer_core.hpp -- device side of the direct peer-write exchanges, shared by the kernels that take part in one
// (peer_exchange.hip: the exchanges on their own; batched.hip: the E-step that scatters its partial sum;
// multigpu.hip: the hit-key merge that composites).  Protocol and parity argument: header of peer_exchange.hip.
// New design (SURVEY.md section 8e "Collective implementation"); the reference is single-GPU.
#pragma once

#include "common.hpp"

namespace emf_hip {

typedef float f4v __attribute__((ext_vector_type(4)));

struct PeerArgs {
    char* slots[EMF_MAX_PEERS];
    uint32_t* flags[EMF_MAX_PEERS];
    int rank, world;
    size_t slotBytes;  // one sender's slot, one parity
    uint32_t* error;
    unsigned long long timeoutTicks;  // of wall_clock64()
    int waitInConsumer;               // 0: the host entry has enqueued k_peer_signal_wait in front (emf_peer_t::waitInFront)
    int fences;                       // emf_peer_t::systemFences: system-scope release before the flags, acquire behind the poll
};

// word of a rank's OWN flag page that mirrors its error word on the device: consumers that did not wait
// themselves read it to skip an exchange whose wait timed out (the error word itself lives in host memory)
constexpr int kDevErrorWord = 64;

__device__ __forceinline__ size_t slot_offset(const PeerArgs& a, int sender, uint32_t seq) {
    return (static_cast<size_t>(seq & 1u) * a.world + sender) * a.slotBytes;
}

// what a peer wrote: not through a stale cache line
__device__ __forceinline__ f4v load_slot16(const char* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
}
__device__ __forceinline__ float load_slot4(const char* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const float*>(p));
}
__device__ __forceinline__ unsigned long long load_slot8(const char* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
}
__device__ __forceinline__ uint8_t load_slot1(const char* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const uint8_t*>(p));
}

// Memory ordering of an exchange, and why no kernel of it executes a system-scope fence (round 4).  Slots and flags
// live in FINE-GRAINED device memory (hipDeviceMallocFinegrained): stores to it -- nontemporal, from the producing
// kernel -- are not held back in the writer's L2, and that kernel has ENDED before the consumer starts (same
// stream), i.e. all of its memory operations are complete.  A system-scope release / __threadfence_system() on top
// adds nothing for these stores but costs a write-back of the whole XCD's L2 -- with the background's sweep
// dirtying 230 MB per frame beside it, hundreds of microseconds per fence, and the fused producers of the first
// version executed one per WAVE: the one-rank sharded frame took 1.15 ms instead of 0.57.  So: the flag is a
// RELAXED system-scope atomic store (it bypasses the caches itself), the poll a relaxed system-scope load, and one
// agent-scope acquire (an L1 invalidate) separates the poll from the slot reads, which are nontemporal loads of
// fine-grained lines nothing on this device has written.
//   That argument has only ever been tested with the ranks on ONE device (hipIpc rehearsals): no xGMI node was
// available.  So it is the protocol only where ranks share a device.  Ranks on DISTINCT devices get
// emf_peer_t::systemFences = 1 from the communicator: a system-scope release in front of every flag store and a
// system-scope acquire behind every poll, in the wait launch and in the in-consumer wait alike (6.6 us per exchange,
// measured on one device) -- until a multi-device run shows the slots never stale without them.

// Signal + wait FUSED INTO THE CONSUMING KERNEL (round 4): called by every thread of every workgroup at the
// kernel's start (it contains barriers).  The kernel's first workgroup raises this rank's flag on every peer and
// lanes 0 .. world-1 of every workgroup spin on this rank's own flag words until all peers show seq.  false: a
// peer's flag did not arrive in time; the error word is set and the caller must not consume the slots.
// (The first workgroup of a grid is dispatched first, so it is resident while the others wait for the peers,
// who do not wait for them: no circular wait whatever the grid size.)
__device__ __forceinline__ bool peer_signal_wait(const PeerArgs& a, uint32_t seq, int tid, bool firstGroup) {
    __shared__ int s_ok;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    if (tid < a.world) {
        if (firstGroup) {
            if (a.fences) {  // (wave-uniform)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(a.flags[tid] + a.rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const uint32_t* mine = a.flags[a.rank] + tid;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const uint32_t seen = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (static_cast<int32_t>(seen - seq) >= 0) break;  // (wrap-around safe)
            if (wall_clock64() - t0 > a.timeoutTicks) {
                __hip_atomic_store(a.error, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (a.fences)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return s_ok != 0;
}

// the wait of this exchange (a launch of its own) timed out: its slots hold stale data
__device__ __forceinline__ bool exchange_failed(const PeerArgs& a, uint32_t seq) {
    return __hip_atomic_load(a.flags[a.rank] + kDevErrorWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq;
}

// First statement of every consuming kernel (all threads; barriers inside): true when the peers' contributions
// of exchange seq are in this rank's slots.  By default a one-wave launch in front of the kernel has signalled and
// waited (emf_peer_t::waitInFront: measured cheaper than a grid of polling workgroups, and the only safe form when
// ranks share a GPU); with waitInFront = 0 the kernel does it itself.
__device__ __forceinline__ bool peer_arrive(const PeerArgs& a, uint32_t seq, int tid, bool firstGroup) {
    if (a.waitInConsumer) return peer_signal_wait(a, seq, tid, firstGroup);
    return !exchange_failed(a, seq);
}

// host: emf_peer_t -> PeerArgs with every field checked (peer_exchange.hip)
int peer_args(const emf_peer_t* g, PeerArgs& a, const char* who);
int peer_wait_in_front(const emf_peer_t* g, uint32_t seq, emf_stream_t stream);

}  // namespace emf_hip
