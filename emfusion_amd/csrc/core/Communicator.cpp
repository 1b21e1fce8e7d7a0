// Communicator.cpp -- RCCL implementation of the cross-GPU exchanges.
#include "Communicator.hpp"

#include <algorithm>

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace emf {
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

private:
    ncclComm_t comm_ = nullptr;
    int rank_, world_;
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
