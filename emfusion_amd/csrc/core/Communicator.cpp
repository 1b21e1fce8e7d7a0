// Communicator.cpp -- RCCL implementation of the cross-GPU exchanges.
#include "Communicator.hpp"

#include <algorithm>

#include <rccl/rccl.h>

#include <cstring>
#include <string>

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

private:
    ncclComm_t comm_ = nullptr;
    int rank_, world_;
};

}  // namespace

void rcclGetUniqueId(void* out) {
    ncclUniqueId id;
    ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, sizeof(id));
}

std::shared_ptr<Communicator> makeRcclCommunicator(const void* uniqueId, int rank, int worldSize) {
    return std::make_shared<RcclCommunicator>(uniqueId, rank, worldSize);
}

}  // namespace emf
