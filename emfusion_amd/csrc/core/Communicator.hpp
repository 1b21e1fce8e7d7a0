// Communicator.hpp -- the two cross-GPU exchanges of the sharded path.
//
// The reference is single-GPU and has no communication layer; this interface is new design.
// Object volumes are sharded round-robin over ranks (one process per GPU), the background is
// replicated.  Per E-step ONE all-reduce(sum, f32, W*H) combines the ranks' object association
// partials into the per-pixel normaliser; per raycast ONE all-reduce(min, u64, W*H) merges the
// nearest-hit keys for compositing.  Both are enqueued on a HIP stream (RCCL over xGMI), never
// synchronising the host.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "types.hpp"

namespace emf {

class Communicator {
public:
    virtual ~Communicator() = default;
    virtual int rank() const = 0;
    virtual int size() const = 0;
    virtual void allReduceSumF32(float* dev, size_t count, Stream& stream) = 0;
    virtual void allReduceMinU64(uint64_t* dev, size_t count, Stream& stream) = 0;
    virtual void broadcast(void* dev, size_t bytes, int root, Stream& stream) = 0;
    /**
     * In place gather of row bands of an image: rank r owns rows [r * bandRows, (r + 1) * bandRows)
     * clipped to totalRows; afterwards every rank holds all rows.  One group of per-band
     * broadcasts (bands may be ragged or empty, which an all-gather's equal counts cannot express).
     */
    virtual void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows,
                                Stream& stream) = 0;
    /**
     * Collectives issued between groupStart() and groupEnd() (same stream) may be fused into ONE
     * launch by the transport (ncclGroupStart / ncclGroupEnd): the nearest-hit merge and the two band
     * gathers of a raycast travel as one exchange.  Backends without such a notion run them one by one.
     */
    virtual void groupStart() {}
    virtual void groupEnd() {}
    /** Collectives this communicator has issued (groups count once): what a frame costs in launches. */
    virtual uint64_t exchangesIssued() const { return 0; }
    /**
     * Direct peer-write transports only (else nullptr): the group as the kernels see it.  With it the sharded path
     * FUSES its exchanges into the kernels around them (emf_hip_estepBatchedPeer -> emf_hip_peerNormalizeAssociation,
     * emf_hip_packHitKeysPeer -> emf_hip_compositeFromKeysPeer): the producer stores into the peers' slots, the
     * consumer signals, waits and reduces -- one extra launch per exchange instead of four.
     * beginPeerExchange() hands out the sequence number of the next exchange, which the caller then enqueues on
     * `stream` with those entries (every rank must issue the same sequence of exchanges, fused or not).
     */
    virtual const emf_peer_t* peerGroup() const { return nullptr; }
    virtual uint32_t beginPeerExchange(Stream& stream) {
        (void)stream;
        throw HipError("Communicator::beginPeerExchange: not a direct peer-write transport", EMF_E_ARG);
    }
    /** Throws (EMF_E_PEER_TIMEOUT) if an exchange enqueued earlier has timed out on the device; call after a
     *  synchronisation.  Transports that report failures at the call itself do nothing. */
    virtual void check() {}
    /**
     * What the transport itself reports about this rank -- not what the launcher asked for: a JSON object
     * {"transport", "ranks" (ncclCommCount for RCCL), "rank", "device" (ordinal the communicator is bound to),
     * "pci_bus_id", "version"}.  bench.py gathers one per rank into the multi-GPU line (`rccl`), so that the first run
     * on a real node shows whether N ranks really sat on N distinct devices.
     */
    virtual std::string describe() const;
};
/** {"device": current HIP device ordinal, "pci_bus_id": "..."} fields of describe(), shared by the transports. */
std::string describeCurrentDevice();

/** Rank that owns an object volume: round-robin by (1-based) object id. */
inline int ownerOf(int objectId, int worldSize) { return (objectId - 1) % worldSize; }

/** Rows per rank of the background-raycast band split: whole 16-row tiles, ceil(tiles / world). */
inline int bgBandRows(int height, int worldSize) {
    const int tiles = (height + 15) / 16;
    return ((tiles + worldSize - 1) / worldSize) * 16;
}

constexpr size_t kRcclUniqueIdBytes = 128;

/** Fills `out` (kRcclUniqueIdBytes) with a fresh ncclUniqueId; call on rank 0 and distribute. */
void rcclGetUniqueId(void* out);
/** ncclCommInitRank on the current device.  Throws HipError on failure. */
std::shared_ptr<Communicator> makeRcclCommunicator(const void* uniqueId, int rank, int worldSize);

/**
 * Rehearsal backend: `worldSize` communicators of ONE process that share a GPU, one per host thread.
 * RCCL refuses two ranks on one device, so the sharded pipeline cannot be run with world > 1 on a
 * single-GPU box through it; this group lets N EMFusion instances on N threads go through exactly the
 * code path of an N-GPU run (ownership, band split, the sequence of collectives -- a mismatch shows
 * as a time-out instead of a hang).  Collectives are staged through host memory and synchronise the
 * calling stream: a test vehicle, not a transport.
 */
std::vector<std::shared_ptr<Communicator>> makeLocalCommunicators(int worldSize);

/**
 * Rehearsal backend for one PROCESS per rank: every collective is staged through host memory and
 * handed to caller-supplied functions (the Python harness passes torch.distributed / gloo).  With it
 * `torchrun --nproc-per-node N bench.py --gpus N --comm gloo` runs the whole multi-rank job on a box
 * with fewer than N GPUs.  Callbacks return 0 on success.
 */
struct HostStagedCallbacks {
    int rank = 0, world = 1;
    int (*allReduceSumF32)(void* user, float* host, size_t count) = nullptr;
    int (*allReduceMinU64)(void* user, uint64_t* host, size_t count) = nullptr;
    int (*broadcast)(void* user, void* host, size_t bytes, int root) = nullptr;
    void* user = nullptr;
};
std::shared_ptr<Communicator> makeHostStagedCommunicator(const HostStagedCallbacks& cb);

/**
 * Direct peer-write exchanges (SURVEY.md section 8e "Collective implementation"; kernels and protocol in
 * csrc/peer_exchange.hip): every rank stores its contribution into a slot of every peer's receive buffer,
 * raises a flag there, waits for the peers' flags and reduces the slots locally in rank order -- three small
 * launches per exchange on the caller's stream, no library collective, the same bits on every rank.
 * `slotBytes` bounds one message (the largest of the path: W*H*8 for the hit keys; multiple of 16).
 * Messages must be multiples of 16 bytes (every image of the path at the usual sizes is).
 *   makePeerCommunicatorsLocal : `worldSize` ranks of ONE process sharing a GPU (threads), plain pointers;
 *   makePeerCommunicator       : one process per rank; the receive buffers are mapped into every peer with
 *                                hipIpcGetMemHandle / hipIpcOpenMemHandle, the 128 handle bytes per rank
 *                                travel through the caller's all-gather (e.g. torch.distributed over gloo).
 * UNTESTED ON xGMI: no multi-GPU box exists in the build environment.  Exercised with 1 rank, with 2-4 ranks
 * on threads and with 2-3 processes over hipIpc, all on one GPU (tests/test_gpu_peer_exchange.py).
 */
std::vector<std::shared_ptr<Communicator>> makePeerCommunicatorsLocal(int worldSize, size_t slotBytes);
struct PeerBootstrap {
    int rank = 0, world = 1;
    /** all[r * bytes .. (r + 1) * bytes) := rank r's `mine`; 0 on success. */
    int (*allGather)(void* user, const void* mine, size_t bytes, void* all) = nullptr;
    void* user = nullptr;
};
std::shared_ptr<Communicator> makePeerCommunicator(const PeerBootstrap& boot, size_t slotBytes);

/**
 * Latency model for single-GPU measurements of the exchange path: every exchange of `inner` (a group
 * counts once) is preceded, on its stream, by a kernel that keeps the stream busy for `microseconds` --
 * what a small-message collective over xGMI costs whatever its size (20-40 us).  Around a 1-rank RCCL
 * communicator (EMF_FORCE_SHARDED=1) this shows how much of the per-frame exchange time the schedule
 * hides: tests/test_gpu_rehearsal.py.
 */
std::shared_ptr<Communicator> makeDelayedCommunicator(std::shared_ptr<Communicator> inner, int microseconds);

}  // namespace emf
