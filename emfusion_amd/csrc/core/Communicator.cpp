// Communicator.cpp -- RCCL implementation of the cross-GPU exchanges.
#include "Communicator.hpp"

#include <algorithm>

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace emf {

std::string describeCurrentDevice() {
    int dev = -1;
    char bus[32] = {0};
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, static_cast<int>(sizeof(bus)) - 1, dev) != hipSuccess)
        (void)hipGetLastError();
    return "\"device\": " + std::to_string(dev) + ", \"pci_bus_id\": \"" + bus + "\"";
}

std::string Communicator::describe() const {
    return "{\"transport\": \"host-staged / in-process (rehearsal)\", \"ranks\": " + std::to_string(size()) + ", \"rank\": " +
           std::to_string(rank()) + ", " + describeCurrentDevice() + ", \"version\": null}";
}

namespace {

void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess)
        throw HipError(std::string(what) + ": " + ncclGetErrorString(r), static_cast<int>(r));
}

class RcclCommunicator final : public Communicator {
public:
    RcclCommunicator(const void* uniqueId, int rank, int world) : rank_(rank), world_(world) {
        static_assert(sizeof(ncclUniqueId) == kRcclUniqueIdBytes, "ncclUniqueId size");
        ncclUniqueId id;
        std::memcpy(&id, uniqueId, sizeof(id));
        ncclCheck(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
    }
    ~RcclCommunicator() override {
        if (comm_) ncclCommDestroy(comm_);
    }
    int rank() const override { return rank_; }
    int size() const override { return world_; }
    void allReduceSumF32(float* dev, size_t count, Stream& s) override {
        ncclCheck(ncclAllReduce(dev, dev, count, ncclFloat32, ncclSum, comm_, s.get()),
                  "ncclAllReduce(sum,f32)");
    }
    void allReduceMinU64(uint64_t* dev, size_t count, Stream& s) override {
        ncclCheck(ncclAllReduce(dev, dev, count, ncclUint64, ncclMin, comm_, s.get()),
                  "ncclAllReduce(min,u64)");
    }
    void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows,
                        Stream& s) override {
        ncclCheck(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < world_; ++r) {
            const int r0 = r * bandRows;
            const int n = std::min(bandRows, totalRows - r0);
            if (n <= 0) break;
            char* p = static_cast<char*>(dev) + static_cast<size_t>(r0) * bytesPerRow;
            ncclCheck(ncclBroadcast(p, p, static_cast<size_t>(n) * bytesPerRow, ncclUint8, r, comm_, s.get()),
                      "ncclBroadcast(band)");
        }
        ncclCheck(ncclGroupEnd(), "ncclGroupEnd");
    }
    void broadcast(void* dev, size_t bytes, int root, Stream& s) override {
        ncclCheck(ncclBroadcast(dev, dev, bytes, ncclUint8, root, comm_, s.get()),
                  "ncclBroadcast");
    }
    void groupStart() override { ncclCheck(ncclGroupStart(), "ncclGroupStart"); }
    void groupEnd() override { ncclCheck(ncclGroupEnd(), "ncclGroupEnd"); }
    std::string describe() const override {
        // asked of the communicator, not remembered from the constructor's arguments
        int count = -1, user = -1, dev = -1, version = 0;
        (void)ncclCommCount(comm_, &count);
        (void)ncclCommUserRank(comm_, &user);
        (void)ncclCommCuDevice(comm_, &dev);
        (void)ncclGetVersion(&version);
        char bus[32] = {0};
        if (dev < 0 || hipDeviceGetPCIBusId(bus, static_cast<int>(sizeof(bus)) - 1, dev) != hipSuccess) (void)hipGetLastError();
        return "{\"transport\": \"rccl\", \"ranks\": " + std::to_string(count) + ", \"rank\": " + std::to_string(user) +
               ", \"device\": " + std::to_string(dev) + ", \"pci_bus_id\": \"" + bus + "\", \"version\": " +
               std::to_string(version) + "}";
    }

private:
    ncclComm_t comm_ = nullptr;
    int rank_, world_;
};

// ---- direct peer-write exchanges (see Communicator.hpp, csrc/peer_exchange.hip) -----------------------
// What one rank owns: its receive buffer, its flag words (device memory the peers write into) and an error
// word in host memory the waiting kernel can reach.
struct PeerMemory {
    void* rx = nullptr;
    uint32_t* flags = nullptr;
    uint32_t* error = nullptr;  // pinned host word
    size_t rxBytes = 0;
    PeerMemory(int world, size_t slotBytes) : rxBytes(emf_hip_peerBufferBytes(world, slotBytes)) {
        // fine-grained: stores of another device must become visible without a cache flush of this one, and
        // coarse-grained memory gives no such promise -- so no silent fall-back for more than one rank
        void* f = nullptr;
        if (world > 1) {
            hipCheck(hipExtMallocWithFlags(&rx, rxBytes, hipDeviceMallocFinegrained),
                     "hipExtMallocWithFlags(fine-grained peer receive buffer)");
            hipCheck(hipExtMallocWithFlags(&f, 4096, hipDeviceMallocFinegrained), "hipExtMallocWithFlags(fine-grained peer flags)");
        } else {
            if (hipExtMallocWithFlags(&rx, rxBytes, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                hipCheck(hipMalloc(&rx, rxBytes), "hipMalloc(peer receive buffer)");
            }
            if (hipExtMallocWithFlags(&f, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                hipCheck(hipMalloc(&f, 4096), "hipMalloc(peer flags)");
            }
        }
        flags = static_cast<uint32_t*>(f);
        hipCheck(hipMemset(flags, 0, 4096), "hipMemset(peer flags)");
        hipCheck(hipHostMalloc(reinterpret_cast<void**>(&error), 64, hipHostMallocCoherent | hipHostMallocMapped),
                 "hipHostMalloc(peer error word)");
        *error = 0;
        hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    ~PeerMemory() {
        if (rx) (void)hipFree(rx);
        if (flags) (void)hipFree(flags);
        if (error) (void)hipHostFree(error);
    }
    PeerMemory(const PeerMemory&) = delete;
    PeerMemory& operator=(const PeerMemory&) = delete;
};

class PeerCommunicator final : public Communicator {
public:
    // `own` is this rank's memory; `keep` holds whatever else must outlive the communicator (the group's
    // memories in the in-process form).  slots / flags of all ranks are already addressable from here.
    PeerCommunicator(int rank, int world, size_t slotBytes, std::shared_ptr<PeerMemory> own,
                     std::vector<std::shared_ptr<PeerMemory>> keep, const std::vector<void*>& slots,
                     const std::vector<uint32_t*>& flags, std::vector<void*> ipcMapped, bool waitInFront /* SOME ranks share a GPU */,
                     bool oneDevice /* ALL ranks sit on one GPU */)
        : rank_(rank), world_(world), own_(std::move(own)), keep_(std::move(keep)), ipcMapped_(std::move(ipcMapped)) {
        // The exchange's signal + wait is a one-wave launch in front of its consumer (measured cheaper than consumers
        // that poll, include/emf_hip.h emf_peer_t::waitInFront); EMF_PEER_WAIT_IN_FRONT=0 lets the consumers poll
        // themselves where every rank has a GPU of its own (A/B on a node; never when ranks share a device).
        g_.waitInFront = 1u;
        // ranks on distinct devices: system-scope release / acquire around every flag (peer_core.hpp "Memory ordering":
        // the fence-free protocol has only been validated with the ranks on one device)
        // -- decided by "every rank on ONE device", not by "some two ranks share one": in a mixed layout (4 ranks over 2
        // GPUs) some pairs do sit on distinct devices
        g_.systemFences = (world > 1 && !oneDevice) ? 1u : 0u;
        if (const char* w = debugEnv("EMF_PEER_WAIT_IN_FRONT")) g_.waitInFront = (w[0] == '0' && !waitInFront) ? 0u : 1u;
        g_.rank = rank;
        g_.world = world;
        g_.slotBytes = slotBytes;
        g_.error = own_->error;
        // EMF_PEER_TIMEOUT_MS: longer bound for rehearsals in which many ranks take turns on one GPU
        if (const char* t = std::getenv("EMF_PEER_TIMEOUT_MS")) timeoutMs_ = static_cast<uint32_t>(std::max(1, std::atoi(t)));
        g_.timeoutMs = timeoutMs_;
        for (int p = 0; p < world; ++p) {
            g_.slots[p] = slots[p];
            g_.flags[p] = flags[p];
        }
    }
    ~PeerCommunicator() override {
        (void)hipDeviceSynchronize();
        for (void* p : ipcMapped_) (void)hipIpcCloseMemHandle(p);
    }
    int rank() const override { return rank_; }
    int size() const override { return world_; }

    // Two launches per exchange (round 3: three): the contribution is scattered into the peers' slots, and the
    // consuming kernel's first workgroup signals while all of its workgroups wait before they reduce / copy.
    // What the sharded frame exchanges goes through peerGroup() / beginPeerExchange() instead, fused into the
    // path's own kernels (core/EMFusion.cpp).
    void allReduceSumF32(float* dev, size_t count, Stream& s) override {
        const uint32_t seq = begin(count * sizeof(float));
        emfCheck(emf_hip_peerScatter(&g_, dev, count * sizeof(float), 0, seq, s.abi()), "peerScatter");
        emfCheck(emf_hip_peerWaitReduceSumF32(&g_, seq, count, dev, s.abi()), "peerWaitReduceSumF32");
    }
    void allReduceMinU64(uint64_t* dev, size_t count, Stream& s) override {
        const uint32_t seq = begin(count * sizeof(uint64_t));
        emfCheck(emf_hip_peerScatter(&g_, dev, count * sizeof(uint64_t), 0, seq, s.abi()), "peerScatter");
        emfCheck(emf_hip_peerWaitReduceMinU64(&g_, seq, count, dev, s.abi()), "peerWaitReduceMinU64");
    }
    void broadcast(void* dev, size_t bytes, int root, Stream& s) override {
        if (root < 0 || root >= world_) throw HipError("peer broadcast: root " + std::to_string(root), EMF_E_ARG);
        const uint32_t seq = begin(bytes);
        if (rank_ == root) emfCheck(emf_hip_peerScatter(&g_, dev, bytes, 0, seq, s.abi()), "peerScatter");
        const int32_t sender = root;
        const size_t off = 0;
        void* dst = dev;
        emfCheck(emf_hip_peerWaitCopyFromSlots(&g_, seq, rank_ == root ? 0 : 1, &sender, &off, &dst, &bytes, s.abi()),
                 "peerWaitCopyFromSlots");
    }
    void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows, Stream& s) override {
        if (bandRows < 1 || totalRows < 0 || (bytesPerRow * static_cast<size_t>(bandRows)) % 16)
            throw HipError("peer gatherRowBands: bands of " + std::to_string(bandRows) + " rows x " +
                           std::to_string(bytesPerRow) + " bytes are not whole 16-byte units", EMF_E_ARG);
        auto band = [&](int r, size_t& off, size_t& bytes) {
            const int r0 = r * bandRows, n = std::min(bandRows, totalRows - r0);
            off = static_cast<size_t>(std::max(r0, 0)) * bytesPerRow;
            bytes = n > 0 ? static_cast<size_t>(n) * bytesPerRow : 0;
        };
        const uint32_t seq = begin(bytesPerRow * static_cast<size_t>(totalRows));
        size_t off, bytes;
        band(rank_, off, bytes);
        if (bytes)
            emfCheck(emf_hip_peerScatter(&g_, static_cast<char*>(dev) + off, bytes, off, seq, s.abi()), "peerScatter");
        std::vector<int32_t> senders;
        std::vector<size_t> offs, sizes;
        std::vector<void*> dsts;
        for (int r = 0; r < world_; ++r) {
            band(r, off, bytes);
            if (r == rank_ || !bytes) continue;
            senders.push_back(r);
            offs.push_back(off);
            sizes.push_back(bytes);
            dsts.push_back(static_cast<char*>(dev) + off);
        }
        emfCheck(emf_hip_peerWaitCopyFromSlots(&g_, seq, static_cast<int>(senders.size()), senders.data(), offs.data(),
                                               dsts.data(), sizes.data(), s.abi()),
                 "peerWaitCopyFromSlots");
    }
    uint64_t exchangesIssued() const override { return seq_; }
    std::string describe() const override {
        return "{\"transport\": \"peer-write (hipIpc-mapped receive buffers)\", \"ranks\": " + std::to_string(world_) +
               ", \"rank\": " + std::to_string(rank_) + ", " + describeCurrentDevice() + ", \"version\": null, \"system_fences\": " +
               std::to_string(g_.systemFences) + ", \"wait_in_front\": " + std::to_string(g_.waitInFront) + "}";
    }
    const emf_peer_t* peerGroup() const override { return &g_; }
    uint32_t beginPeerExchange(Stream&) override { return begin(0); }
    void check() override {
        if (*own_->error)
            throw HipError("peer exchange " + std::to_string(*own_->error) + ": a peer's flag did not arrive within " +
                           std::to_string(*own_->error == 1u ? std::max(timeoutMs_, 60000u) : timeoutMs_) + " ms (ranks disagree about the sequence of exchanges?); the "
                           "exchange's consumer left its outputs untouched", EMF_E_PEER_TIMEOUT);
    }

private:
    uint32_t timeoutMs_ = 5000;
    // every argument is validated BEFORE the sequence number moves: a rank that throws here has not taken part in
    // the exchange and its peers' counters stay in step with its own
    uint32_t begin(size_t bytes) {
        check();
        if (bytes > g_.slotBytes)
            throw HipError("peer exchange: message of " + std::to_string(bytes) + " bytes exceeds the slot size " +
                           std::to_string(g_.slotBytes), EMF_E_LIMIT);
        if (bytes % 16)
            throw HipError("peer exchange: message of " + std::to_string(bytes) + " bytes is not a multiple of 16", EMF_E_ARG);
        // the FIRST exchange also absorbs the ranks' start-up skew (a rank whose first launches load code objects, or
        // whose process came up later): a minute for it, the configured bound from then on
        g_.timeoutMs = seq_ == 0 ? std::max(timeoutMs_, 60000u) : timeoutMs_;
        return ++seq_;
    }
    int rank_, world_;
    std::shared_ptr<PeerMemory> own_;
    std::vector<std::shared_ptr<PeerMemory>> keep_;
    std::vector<void*> ipcMapped_;
    emf_peer_t g_{};
    uint32_t seq_ = 0;
};

// ---- latency model (see Communicator.hpp) ------------------------------------------------------------
class DelayedCommunicator final : public Communicator {
public:
    DelayedCommunicator(std::shared_ptr<Communicator> inner, int us) : inner_(std::move(inner)), us_(us) {}
    int rank() const override { return inner_->rank(); }
    int size() const override { return inner_->size(); }
    void allReduceSumF32(float* dev, size_t count, Stream& s) override {
        delay(s);
        inner_->allReduceSumF32(dev, count, s);
    }
    void allReduceMinU64(uint64_t* dev, size_t count, Stream& s) override {
        delay(s);
        inner_->allReduceMinU64(dev, count, s);
    }
    void broadcast(void* dev, size_t bytes, int root, Stream& s) override {
        delay(s);
        inner_->broadcast(dev, bytes, root, s);
    }
    void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows, Stream& s) override {
        delay(s);
        inner_->gatherRowBands(dev, bytesPerRow, bandRows, totalRows, s);
    }
    void groupStart() override {
        inGroup_ = true;
        groupDelayed_ = false;
        inner_->groupStart();
    }
    void groupEnd() override {
        inner_->groupEnd();
        inGroup_ = false;
    }
    uint64_t exchangesIssued() const override { return exchanges_; }
    std::string describe() const override { return inner_->describe(); }
    const emf_peer_t* peerGroup() const override { return inner_->peerGroup(); }
    uint32_t beginPeerExchange(Stream& s) override {
        delay(s);
        return inner_->beginPeerExchange(s);
    }
    void check() override { inner_->check(); }

private:
    void delay(Stream& s) {
        if (inGroup_ && groupDelayed_) return;  // one latency per group
        groupDelayed_ = true;
        ++exchanges_;
        if (us_ > 0) emfCheck(emf_hip_spinDelay(static_cast<uint32_t>(us_), s.abi()), "spinDelay");
    }
    std::shared_ptr<Communicator> inner_;
    int us_;
    bool inGroup_ = false, groupDelayed_ = false;
    uint64_t exchanges_ = 0;
};

// ---- in-process rehearsal group (see Communicator.hpp) -----------------------------------------------

struct LocalGroup {
    explicit LocalGroup(int n) : world(n), slots(n) {}
    const int world;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;
    std::vector<std::vector<unsigned char>> slots;  // what each rank published for the running collective

    // all ranks arrive, then all leave; a rank that waits too long means the ranks disagree about
    // the sequence of collectives
    void barrier() {
        std::unique_lock<std::mutex> lock(m);
        const unsigned long gen = generation;
        if (++waiting == world) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        if (!cv.wait_for(lock, std::chrono::seconds(30), [&] { return generation != gen; }))
            throw HipError("local communicator: a rank did not arrive at the collective within 30 s "
                           "(ranks disagree about the sequence of collectives?)", EMF_E_ARG);
    }
};

class LocalCommunicator final : public Communicator {
public:
    LocalCommunicator(std::shared_ptr<LocalGroup> g, int rank) : g_(std::move(g)), rank_(rank) {}
    int rank() const override { return rank_; }
    int size() const override { return g_->world; }

    void allReduceSumF32(float* dev, size_t count, Stream& s) override {
        publish(dev, count * sizeof(float), s);
        std::vector<float> acc(count, 0.f);
        for (int r = 0; r < g_->world; ++r) {  // rank order: the same sum on every rank
            const float* v = reinterpret_cast<const float*>(g_->slots[r].data());
            if (r == 0) std::copy(v, v + count, acc.begin());
            else for (size_t i = 0; i < count; ++i) acc[i] += v[i];
        }
        finish(dev, acc.data(), count * sizeof(float), s);
    }
    void allReduceMinU64(uint64_t* dev, size_t count, Stream& s) override {
        publish(dev, count * sizeof(uint64_t), s);
        std::vector<uint64_t> acc(count);
        for (int r = 0; r < g_->world; ++r) {
            const uint64_t* v = reinterpret_cast<const uint64_t*>(g_->slots[r].data());
            for (size_t i = 0; i < count; ++i) acc[i] = r == 0 ? v[i] : std::min(acc[i], v[i]);
        }
        finish(dev, acc.data(), count * sizeof(uint64_t), s);
    }
    void broadcast(void* dev, size_t bytes, int root, Stream& s) override {
        publish(dev, bytes, s);
        std::vector<unsigned char> v(g_->slots[root]);
        finish(dev, v.data(), bytes, s);
    }
    void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows, Stream& s) override {
        const size_t bytes = bytesPerRow * static_cast<size_t>(totalRows);
        publish(dev, bytes, s);
        std::vector<unsigned char> v(bytes);
        for (int r = 0; r < g_->world; ++r) {
            const int r0 = r * bandRows, n = std::min(bandRows, totalRows - r0);
            if (n <= 0) break;
            const size_t off = static_cast<size_t>(r0) * bytesPerRow;
            std::memcpy(v.data() + off, g_->slots[r].data() + off, static_cast<size_t>(n) * bytesPerRow);
        }
        // rows beyond the last band (none: the bands cover the image) would keep this rank's values
        finish(dev, v.data(), bytes, s);
    }

private:
    // device -> this rank's slot, then wait until every rank has published
    void publish(const void* dev, size_t bytes, Stream& s) {
        auto& slot = g_->slots[rank_];
        slot.resize(bytes);
        hipCheck(hipMemcpyAsync(slot.data(), dev, bytes, hipMemcpyDeviceToHost, s.get()), "local comm D2H");
        s.waitForCompletion();
        g_->barrier();
    }
    // result -> device, then wait until every rank has consumed the slots (they are reused)
    void finish(void* dev, const void* host, size_t bytes, Stream& s) {
        hipCheck(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s.get()), "local comm H2D");
        s.waitForCompletion();
        g_->barrier();
    }
    std::shared_ptr<LocalGroup> g_;
    int rank_;
};

class HostStagedCommunicator final : public Communicator {
public:
    explicit HostStagedCommunicator(const HostStagedCallbacks& cb) : cb_(cb) {
        if (!cb.allReduceSumF32 || !cb.allReduceMinU64 || !cb.broadcast || cb.world < 1 || cb.rank < 0 ||
            cb.rank >= cb.world)
            throw HipError("makeHostStagedCommunicator: incomplete callbacks", EMF_E_ARG);
    }
    int rank() const override { return cb_.rank; }
    int size() const override { return cb_.world; }
    void allReduceSumF32(float* dev, size_t count, Stream& s) override {
        down(dev, count * sizeof(float), s);
        check(cb_.allReduceSumF32(cb_.user, reinterpret_cast<float*>(host_.data()), count), "all-reduce(sum)");
        up(dev, count * sizeof(float), s);
    }
    void allReduceMinU64(uint64_t* dev, size_t count, Stream& s) override {
        down(dev, count * sizeof(uint64_t), s);
        check(cb_.allReduceMinU64(cb_.user, reinterpret_cast<uint64_t*>(host_.data()), count), "all-reduce(min)");
        up(dev, count * sizeof(uint64_t), s);
    }
    void broadcast(void* dev, size_t bytes, int root, Stream& s) override {
        down(dev, bytes, s);
        check(cb_.broadcast(cb_.user, host_.data(), bytes, root), "broadcast");
        up(dev, bytes, s);
    }
    void gatherRowBands(void* dev, size_t bytesPerRow, int bandRows, int totalRows, Stream& s) override {
        const size_t bytes = bytesPerRow * static_cast<size_t>(totalRows);
        down(dev, bytes, s);
        for (int r = 0; r < cb_.world; ++r) {
            const int r0 = r * bandRows, n = std::min(bandRows, totalRows - r0);
            if (n <= 0) break;
            check(cb_.broadcast(cb_.user, host_.data() + static_cast<size_t>(r0) * bytesPerRow,
                                static_cast<size_t>(n) * bytesPerRow, r), "broadcast(band)");
        }
        up(dev, bytes, s);
    }

private:
    static void check(int rc, const char* what) {
        if (rc != 0) throw HipError(std::string("host-staged communicator: ") + what + " failed", rc);
    }
    void down(const void* dev, size_t bytes, Stream& s) {
        host_.resize(bytes);
        hipCheck(hipMemcpyAsync(host_.data(), dev, bytes, hipMemcpyDeviceToHost, s.get()), "staged D2H");
        s.waitForCompletion();
    }
    void up(void* dev, size_t bytes, Stream& s) {
        hipCheck(hipMemcpyAsync(dev, host_.data(), bytes, hipMemcpyHostToDevice, s.get()), "staged H2D");
        s.waitForCompletion();
    }
    HostStagedCallbacks cb_;
    std::vector<unsigned char> host_;
};

}  // namespace

std::vector<std::shared_ptr<Communicator>> makePeerCommunicatorsLocal(int worldSize, size_t slotBytes) {
    if (worldSize < 1 || worldSize > EMF_MAX_PEERS) throw HipError("makePeerCommunicatorsLocal: world size", EMF_E_LIMIT);
    slotBytes = (slotBytes + 15) / 16 * 16;
    std::vector<std::shared_ptr<PeerMemory>> mem;
    std::vector<void*> slots;
    std::vector<uint32_t*> flags;
    for (int r = 0; r < worldSize; ++r) {
        mem.push_back(std::make_shared<PeerMemory>(worldSize, slotBytes));
        slots.push_back(mem.back()->rx);
        flags.push_back(mem.back()->flags);
    }
    std::vector<std::shared_ptr<Communicator>> out;
    for (int r = 0; r < worldSize; ++r)
        out.push_back(std::make_shared<PeerCommunicator>(r, worldSize, slotBytes, mem[r], mem, slots, flags,
                                                         std::vector<void*>(), worldSize > 1, true));
    return out;
}

std::shared_ptr<Communicator> makePeerCommunicator(const PeerBootstrap& boot, size_t slotBytes) {
    if (boot.world < 1 || boot.world > EMF_MAX_PEERS || boot.rank < 0 || boot.rank >= boot.world || !boot.allGather)
        throw HipError("makePeerCommunicator: bad bootstrap", EMF_E_ARG);
    slotBytes = (slotBytes + 15) / 16 * 16;
    auto own = std::make_shared<PeerMemory>(boot.world, slotBytes);
    struct Handles {
        hipIpcMemHandle_t rx, flags;
        char bus[32];  // PCI bus id of the rank's device: ranks that share a GPU (rehearsals) are told apart by it
    };
    static_assert(sizeof(Handles) == 160, "two 64-byte hipIpcMemHandle_t + the bus id");
    Handles mine;
    std::memset(&mine, 0, sizeof(mine));
    {
        int dev = 0;
        hipCheck(hipGetDevice(&dev), "hipGetDevice");
        hipCheck(hipDeviceGetPCIBusId(mine.bus, static_cast<int>(sizeof(mine.bus)) - 1, dev), "hipDeviceGetPCIBusId");
    }
    hipCheck(hipIpcGetMemHandle(&mine.rx, own->rx), "hipIpcGetMemHandle(receive buffer)");
    hipCheck(hipIpcGetMemHandle(&mine.flags, own->flags), "hipIpcGetMemHandle(flags)");
    std::vector<Handles> all(boot.world);
    if (boot.allGather(boot.user, &mine, sizeof(mine), all.data()) != 0)
        throw HipError("makePeerCommunicator: the bootstrap all-gather failed", EMF_E_ARG);
    std::vector<void*> slots(boot.world), mapped;
    std::vector<uint32_t*> flags(boot.world);
    bool shared = false, oneDevice = true;  // some pair of ranks on one GPU / all ranks on one GPU
    for (int p = 0; p < boot.world; ++p)
        for (int q = p + 1; q < boot.world; ++q) {
            const bool same = std::strncmp(all[p].bus, all[q].bus, sizeof(mine.bus)) == 0;
            shared = shared || same;
            oneDevice = oneDevice && same;
        }
    for (int p = 0; p < boot.world; ++p) {
        if (p == boot.rank) {
            slots[p] = own->rx;
            flags[p] = own->flags;
            continue;
        }
        void *a = nullptr, *b = nullptr;
        hipCheck(hipIpcOpenMemHandle(&a, all[p].rx, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(receive buffer)");
        hipCheck(hipIpcOpenMemHandle(&b, all[p].flags, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(flags)");
        slots[p] = a;
        flags[p] = static_cast<uint32_t*>(b);
        mapped.push_back(a);
        mapped.push_back(b);
    }
    return std::make_shared<PeerCommunicator>(boot.rank, boot.world, slotBytes, own,
                                              std::vector<std::shared_ptr<PeerMemory>>(), slots, flags, mapped, shared, oneDevice);
}

std::shared_ptr<Communicator> makeDelayedCommunicator(std::shared_ptr<Communicator> inner, int microseconds) {
    if (!inner) throw HipError("makeDelayedCommunicator: no inner communicator", EMF_E_NULL);
    return std::make_shared<DelayedCommunicator>(std::move(inner), microseconds);
}

std::shared_ptr<Communicator> makeHostStagedCommunicator(const HostStagedCallbacks& cb) {
    return std::make_shared<HostStagedCommunicator>(cb);
}

std::vector<std::shared_ptr<Communicator>> makeLocalCommunicators(int worldSize) {
    if (worldSize < 1) throw HipError("makeLocalCommunicators: world size < 1", EMF_E_ARG);
    auto g = std::make_shared<LocalGroup>(worldSize);
    std::vector<std::shared_ptr<Communicator>> out;
    for (int r = 0; r < worldSize; ++r) out.push_back(std::make_shared<LocalCommunicator>(g, r));
    return out;
}

void rcclGetUniqueId(void* out) {
    ncclUniqueId id;
    ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, sizeof(id));
}

std::shared_ptr<Communicator> makeRcclCommunicator(const void* uniqueId, int rank, int worldSize) {
    return std::make_shared<RcclCommunicator>(uniqueId, rank, worldSize);
}

}  // namespace emf
