#include "KernelTimers.hpp"

namespace emf {

KernelTimers::~KernelTimers() {
    for (auto& p : pairs) {
        if (p.start) (void)hipEventDestroy(p.start);
        if (p.stop) (void)hipEventDestroy(p.stop);
    }
}

void KernelTimers::enable(size_t maxLaunches) {
    for (auto& p : pairs) {
        if (p.start) (void)hipEventDestroy(p.start);
        if (p.stop) (void)hipEventDestroy(p.stop);
    }
    pairs.assign(maxLaunches, Pair{});
    for (auto& p : pairs) {
        hipCheck(hipEventCreate(&p.start), "hipEventCreate");
        hipCheck(hipEventCreate(&p.stop), "hipEventCreate");
    }
    used = dropped = 0;
    seen.fill(0u);
}

KernelTimers::Scope::Scope(KernelTimers* t, Kind k, double units, hipStream_t s)
    : timers(t), stream(s), slot(-1) {
    if (!t || t->pairs.empty() || !((t->kindMask >> k) & 1u)) return;
    // every stride-th launch, starting in the middle of the first stride: the sample then sits centred in a run
    // whose launches drift (volumes filling up over the first frames), not at its slow end
    if (t->seen[k]++ % t->stride != t->stride / 2) return;
    if (t->used >= t->pairs.size()) {
        ++t->dropped;
        return;
    }
    slot = static_cast<long>(t->used++);
    Pair& p = t->pairs[slot];
    p.kind = k;
    p.units = units;
    hipCheck(hipEventRecord(p.start, stream), "hipEventRecord");
}

KernelTimers::Scope::~Scope() {
    if (slot >= 0) (void)hipEventRecord(timers->pairs[slot].stop, stream);
}

std::array<KernelTimers::Summary, KernelTimers::kNumKinds> KernelTimers::collect() const {
    std::array<Summary, kNumKinds> out{};
    for (size_t i = 0; i < used; ++i) {
        const Pair& p = pairs[i];
        float ms = 0.f;
        hipCheck(hipEventElapsedTime(&ms, p.start, p.stop), "hipEventElapsedTime");
        Summary& s = out[p.kind];
        ++s.launches;
        s.total_ms += ms;
        s.units += p.units;
    }
    return out;
}

}  // namespace emf
