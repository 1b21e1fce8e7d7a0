// peer_exchange.hip -- one-shot exchanges by direct peer writes ("Collective
// implementation": the messages of this path are 1-5 MB, so hops -- latency -- decide, not the per-link bound;
// a ring is the wrong shape).  The reference is single-GPU: nothing to cite, this is new design.
//
// Every rank owns a receive buffer of `world` slots (one per sender, twice: two parities) and `world` flag
// words; every rank can address every peer's buffer and flags (same process: plain pointers; one process per
// GPU: hipIpc mappings over xGMI, made once by the host, core/Communicator.cpp).  An exchange with sequence
// number s is, on the caller's stream and without host involvement,
//   1. scatter      my contribution -> slot[me] (parity s & 1) of EVERY peer, all links at once
//   2. signal+wait  flag[me] := s on every peer (system-scope release), then spin until my own flags of all
//                   peers show >= s (system-scope acquire); bounded: a time-out sets the error word
//   3. reduce/copy  sum / min the world slots locally IN RANK ORDER -- the same bits on every rank -- or copy
//                   the sender's slot (broadcast, band gather)
// Round 3 ran these as three launches per exchange.  Round 4: step 2 happens at the START OF THE CONSUMING
// KERNEL (peer_core.hpp: peer_signal_wait) and step 1 inside the producing kernel where the path has one
// (k_estep stores its partial sum straight into the peers' slots, k_pack_keys_peer the hit keys and its band of
// the background raycast), so an exchange of the frame adds ONE launch -- the consumer, which also does what
// used to follow the exchange (normalisation; compositing + visibility counts).  The three-launch entries stay
// for callers that exchange a buffer of their own.
// Parities: slot halves alternate with s.  A sender may overwrite parity s & 1 again in exchange s + 2 only
// after it has seen every peer's flag >= s + 1, and a peer raises that flag behind its own reduce of exchange
// s (stream order): nobody is still reading what is overwritten.  Every exchange signals and waits all-to-all,
// also a broadcast, so that this argument needs no case analysis.
// Untested on xGMI (no multi-GPU box in the build environment): exercised with one rank, with 2-4 ranks on
// threads of one process and with 2-3 ranks in separate processes over hipIpc, all sharing one GPU.
#include "peer_core.hpp"

#include <algorithm>
#include <mutex>

namespace emf_hip {
namespace {

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
    if (a.fences) __threadfence_system();  // (ranks on distinct devices: peer_core.hpp "Memory ordering")
}

// one wave: lane p signals peer p, then waits for peer p's signal
__global__ void k_peer_signal_wait(PeerArgs a, uint32_t seq, unsigned long long timeoutTicks, int fences) {
    const int p = threadIdx.x;
    if (p >= a.world) return;
    // emf_peer_t::systemFences (wave-uniform): the belt to the braces of peer_core.hpp's "Memory ordering of an exchange"
    // -- a system-scope release in front of the flags and an acquire behind the wait.  One wave per exchange, and still
    // 6.6 us each while the background's sweep keeps the L2 dirty (0.654 -> 0.688 ms per one-rank sharded frame): on
    // for ranks on distinct devices (never validated without), off where they share one.
    if (fences) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_store(a.flags[p] + a.rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = a.flags[a.rank] + p;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t seen = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (static_cast<int32_t>(seen - seq) >= 0) break;  // (wrap-around safe)
        if (wall_clock64() - t0 > timeoutTicks) {
            __hip_atomic_store(a.error, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.flags[a.rank] + kDevErrorWord, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

__global__ __launch_bounds__(256) void k_peer_reduce_sum_f32(PeerArgs a, uint32_t seq, size_t count, float* __restrict__ out,
                                                             int waitFirst) {
    if (waitFirst ? !peer_arrive(a, seq, threadIdx.x, blockIdx.x == 0) : exchange_failed(a, seq)) return;
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
                                                             unsigned long long* __restrict__ out, int waitFirst) {
    typedef unsigned long long u2v __attribute__((ext_vector_type(2)));
    if (waitFirst ? !peer_arrive(a, seq, threadIdx.x, blockIdx.x == 0) : exchange_failed(a, seq)) return;
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
    if (exchange_failed(a, seq)) return;
    const size_t n16 = bytes / 16;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const char* src = a.slots[a.rank] + slot_offset(a, sender, seq) + srcOffset;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
        *reinterpret_cast<f4v*>(dst + 16 * i) = load_slot16(src + 16 * i);
}

// signal + wait + up to kCopyParts copies out of the senders' slots (broadcast: one part; band gathers: one per
// peer and image) in ONE launch; nparts may be 0 (the root of a broadcast only signals and waits)
constexpr int kCopyParts = 2 * EMF_MAX_PEERS;
struct CopyParts {
    int sender[kCopyParts];
    size_t srcOffset[kCopyParts], bytes[kCopyParts];
    char* dst[kCopyParts];
    int count;
};
__global__ __launch_bounds__(256) void k_peer_wait_copy(PeerArgs a, uint32_t seq, CopyParts parts) {
    if (!peer_arrive(a, seq, threadIdx.x, blockIdx.x == 0)) return;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (int k = 0; k < parts.count; ++k) {
        const char* src = a.slots[a.rank] + slot_offset(a, parts.sender[k], seq) + parts.srcOffset[k];
        char* dst = parts.dst[k];
        const size_t n16 = parts.bytes[k] / 16;
        for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
            *reinterpret_cast<f4v*>(dst + 16 * i) = load_slot16(src + 16 * i);
    }
}

// ---- the E-step's exchange, consumer side: wait, sum the ranks' object partials in rank order, normalise ----
// (reference EMFusion.cpp:653-665: norm = w_bg + sum of the objects' maps, every map divided by it, x / 0 := 0;
// the same arithmetic as k_assoc_normalize with nsum = 1 and the reduced partial as `extra`, pixel_ops.hip)
constexpr int kNormMaps = 16;
struct NormTable {
    Img<float> m[kNormMaps];
    int count;
};
// A resident grid: every workgroup polls the flags at its start (peer_core.hpp), and the polls of 1200 workgroups
// on the same uncached words cost this kernel 35 of its 40 us (measured with one rank); kPollGroups workgroups
// walk the image's 64 x 4 tiles instead.
constexpr int kPollGroups = 256;
__global__ __launch_bounds__(256) void k_peer_normalize(PeerArgs a, uint32_t seq, NormTable t, Img<float> objSum,
                                                        Img<float> norm, int w, int h) {
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (!peer_arrive(a, seq, tid, blockIdx.x == 0)) return;
    const int tilesX = (w + 63) / 64, tiles = tilesX * ((h + 3) / 4);
    const char* base = a.slots[a.rank];
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int x = (tile % tilesX) * 64 + threadIdx.x, y = (tile / tilesX) * 4 + threadIdx.y;
        if (x >= w || y >= h) continue;
        const size_t pix = static_cast<size_t>(y) * w + x;
        float e = load_slot4(base + slot_offset(a, 0, seq) + 4 * pix);
        for (int r = 1; r < a.world; ++r) e = e + load_slot4(base + slot_offset(a, r, seq) + 4 * pix);
        if (objSum.data) objSum.row(y)[x] = e;
        float v[kNormMaps];
#pragma unroll
        for (int k = 0; k < kNormMaps; ++k)
            if (k < t.count) v[k] = t.m[k].row(y)[x];
        const float s = v[0] + e;  // background (this rank's replica) + objects of all ranks
        if (norm.data) norm.row(y)[x] = s;
#pragma unroll
        for (int k = 0; k < kNormMaps; ++k)
            if (k < t.count) t.m[k].row(y)[x] = (s != 0.f) ? v[k] / s : 0.f;
    }
}

unsigned long long timeout_ticks(uint32_t timeoutMs) {  // wall_clock64() ticks; the rate is the device's, asked once
    static std::mutex m;
    static int rateKHz[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int khz = 0;
    {
        std::lock_guard<std::mutex> lock(m);
        if (dev >= 0 && dev < 64) khz = rateKHz[dev];
        if (khz <= 0) {
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
                (void)hipGetLastError();
                khz = 100000;  // 100 MHz: gfx9's constant-rate counter
            }
            if (dev >= 0 && dev < 64) rateKHz[dev] = khz;
        }
    }
    return static_cast<unsigned long long>(timeoutMs ? timeoutMs : 5000u) * static_cast<unsigned long long>(khz);
}

unsigned grid_for(size_t units) {  // 16-byte units; a few per lane
    const size_t blocks = units / 256 / 4 + 1;
    return static_cast<unsigned>(blocks < 2048 ? blocks : 2048);
}

bool aligned16(const void* p, size_t a, size_t b) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && a % 16 == 0 && b % 16 == 0;
}

}  // namespace

int peer_args(const emf_peer_t* g, PeerArgs& a, const char* who) {
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
    a.timeoutTicks = timeout_ticks(g->timeoutMs);
    a.waitInConsumer = g->waitInFront ? 0 : 1;
    a.fences = g->systemFences ? 1 : 0;
    return EMF_OK;
}

// ranks sharing a GPU: the exchange's signal + wait as a one-wave launch in front of its consumer
int peer_wait_in_front(const emf_peer_t* g, uint32_t seq, emf_stream_t stream) {
    if (!g || !g->waitInFront) return EMF_OK;
    return emf_hip_peerSignalWait(g, seq, 0, stream);
}

}  // namespace emf_hip

extern "C" {

size_t emf_hip_peerBufferBytes(int world, size_t slotBytes) { return 2 * static_cast<size_t>(world) * slotBytes; }

int emf_hip_peerScatter(const emf_peer_t* group, const void* src, size_t bytes, size_t dstOffset, uint32_t seq,
                        emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, "peerScatter")) return rc;
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
    if (const int rc = peer_args(group, a, "peerSignalWait")) return rc;
    hipLaunchKernelGGL(k_peer_signal_wait, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, seq,
                       timeoutMs ? timeout_ticks(timeoutMs) : a.timeoutTicks, group->systemFences ? 1 : 0);
    return launch_status("peerSignalWait");
}

namespace {
int reduce_sum(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream, int waitFirst,
               const char* fn) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, fn)) return rc;
    if (!out) return fail(EMF_E_NULL, "%s: out is NULL", fn);
    if (count % 4 || count * 4 > a.slotBytes || reinterpret_cast<uintptr_t>(out) % 16)
        return fail(EMF_E_ARG, "%s: count %zu (multiple of 4, at most %zu)", fn, count, a.slotBytes / 4);
    if (count == 0 && !waitFirst) return EMF_OK;
    if (waitFirst) EMF_TRY(peer_wait_in_front(group, seq, stream));
    hipLaunchKernelGGL(k_peer_reduce_sum_f32, dim3(grid_for(count / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, count, out, waitFirst);
    return launch_status(fn);
}
int reduce_min(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream, int waitFirst,
               const char* fn) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, fn)) return rc;
    if (!out) return fail(EMF_E_NULL, "%s: out is NULL", fn);
    if (count % 2 || count * 8 > a.slotBytes || reinterpret_cast<uintptr_t>(out) % 16)
        return fail(EMF_E_ARG, "%s: count %zu (even, at most %zu)", fn, count, a.slotBytes / 8);
    if (count == 0 && !waitFirst) return EMF_OK;
    if (waitFirst) EMF_TRY(peer_wait_in_front(group, seq, stream));
    hipLaunchKernelGGL(k_peer_reduce_min_u64, dim3(grid_for(count / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, count, reinterpret_cast<unsigned long long*>(out), waitFirst);
    return launch_status(fn);
}
}  // namespace

int emf_hip_peerReduceSumF32(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream) {
    return reduce_sum(group, seq, count, out, stream, 0, "peerReduceSumF32");
}
int emf_hip_peerReduceMinU64(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream) {
    return reduce_min(group, seq, count, out, stream, 0, "peerReduceMinU64");
}
int emf_hip_peerWaitReduceSumF32(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream) {
    return reduce_sum(group, seq, count, out, stream, 1, "peerWaitReduceSumF32");
}
int emf_hip_peerWaitReduceMinU64(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream) {
    return reduce_min(group, seq, count, out, stream, 1, "peerWaitReduceMinU64");
}

int emf_hip_peerCopyFromSlot(const emf_peer_t* group, uint32_t seq, int sender, size_t srcOffset, void* dst, size_t bytes,
                             emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, "peerCopyFromSlot")) return rc;
    if (!dst) return fail(EMF_E_NULL, "peerCopyFromSlot: dst is NULL");
    if (sender < 0 || sender >= a.world || !aligned16(dst, bytes, srcOffset) || srcOffset + bytes > a.slotBytes)
        return fail(EMF_E_ARG, "peerCopyFromSlot: sender %d, %zu bytes at offset %zu", sender, bytes, srcOffset);
    if (bytes == 0) return EMF_OK;
    hipLaunchKernelGGL(k_peer_copy_from_slot, dim3(grid_for(bytes / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, seq, sender, srcOffset, static_cast<char*>(dst), bytes);
    return launch_status("peerCopyFromSlot");
}

int emf_hip_peerWaitCopyFromSlots(const emf_peer_t* group, uint32_t seq, int nparts, const int32_t* senders_host,
                                  const size_t* srcOffsets_host, void* const* dsts_host, const size_t* bytes_host,
                                  emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, "peerWaitCopyFromSlots")) return rc;
    if (nparts < 0 || nparts > kCopyParts) return fail(EMF_E_LIMIT, "peerWaitCopyFromSlots: %d parts (at most %d)", nparts, kCopyParts);
    if (nparts > 0 && (!senders_host || !srcOffsets_host || !dsts_host || !bytes_host))
        return fail(EMF_E_NULL, "peerWaitCopyFromSlots: NULL part table");
    CopyParts parts;
    parts.count = 0;
    size_t most = 0;
    for (int k = 0; k < nparts; ++k) {
        if (bytes_host[k] == 0) continue;
        if (!dsts_host[k]) return fail(EMF_E_NULL, "peerWaitCopyFromSlots: dst %d is NULL", k);
        if (senders_host[k] < 0 || senders_host[k] >= a.world || !aligned16(dsts_host[k], bytes_host[k], srcOffsets_host[k]) ||
            srcOffsets_host[k] + bytes_host[k] > a.slotBytes)
            return fail(EMF_E_ARG, "peerWaitCopyFromSlots: part %d: sender %d, %zu bytes at offset %zu", k, senders_host[k],
                        bytes_host[k], srcOffsets_host[k]);
        const int j = parts.count++;
        parts.sender[j] = senders_host[k];
        parts.srcOffset[j] = srcOffsets_host[k];
        parts.bytes[j] = bytes_host[k];
        parts.dst[j] = static_cast<char*>(dsts_host[k]);
        most = bytes_host[k] > most ? bytes_host[k] : most;
    }
    EMF_TRY(peer_wait_in_front(group, seq, stream));
    if (parts.count == 0 && group->waitInFront) return EMF_OK;  // (a broadcast's root: the wait launch was all of it)
    hipLaunchKernelGGL(k_peer_wait_copy, dim3(grid_for(most / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, seq,
                       parts);
    return launch_status("peerWaitCopyFromSlots");
}

int emf_hip_peerNormalizeAssociation(const emf_peer_t* group, uint32_t seq, const emf_image_t* maps_host, int nmaps,
                                     const emf_image_t* objSum, const emf_image_t* norm, emf_stream_t stream) {
    using namespace emf_hip;
    PeerArgs a;
    if (const int rc = peer_args(group, a, "peerNormalizeAssociation")) return rc;
    EMF_REQUIRE_PTR(maps_host);
    if (nmaps < 1 || nmaps > kNormMaps)
        return fail(EMF_E_LIMIT, "peerNormalizeAssociation: %d maps (1..%d; more: peerWaitReduceSumF32 + normalizeAssociation)",
                    nmaps, kNormMaps);
    NormTable t;
    t.count = nmaps;
    for (int k = 0; k < nmaps; ++k) {
        EMF_TRY(check_image(&maps_host[k], 4, "peerNormalizeAssociation: map"));
        EMF_TRY(check_same_size(&maps_host[k], &maps_host[0], "map", "map 0"));
        t.m[k] = img<float>(&maps_host[k]);
    }
    const int w = maps_host[0].width, h = maps_host[0].height;
    if (static_cast<size_t>(w) * h * 4 > a.slotBytes)
        return fail(EMF_E_ARG, "peerNormalizeAssociation: %d x %d floats exceed the %zu-byte slot", w, h, a.slotBytes);
    Img<float> so{nullptr, 0}, no{nullptr, 0};
    if (objSum) {
        EMF_TRY(check_image(objSum, 4, "peerNormalizeAssociation: objSum"));
        EMF_TRY(check_same_size(objSum, &maps_host[0], "objSum", "map 0"));
        so = img<float>(objSum);
    }
    if (norm) {
        EMF_TRY(check_image(norm, 4, "peerNormalizeAssociation: norm"));
        EMF_TRY(check_same_size(norm, &maps_host[0], "norm", "map 0"));
        no = img<float>(norm);
    }
    EMF_TRY(peer_wait_in_front(group, seq, stream));
    // a polling grid stays small (kPollGroups); behind the one-wave wait launch every tile gets its workgroup
    const unsigned tiles = ceil_div(w, 64) * ceil_div(h, 4);
    hipLaunchKernelGGL(k_peer_normalize, dim3(group->waitInFront ? tiles : std::min<unsigned>(kPollGroups, tiles)), dim3(64, 4), 0,
                       reinterpret_cast<hipStream_t>(stream), a, seq, t, so, no, w, h);
    return launch_status("peerNormalizeAssociation");
}

}  // extern "C"
