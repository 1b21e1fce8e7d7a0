// KernelTimers.hpp -- per-launch HIP-event timing of the path's kernels.
//
// Every launch site in emf::EMFusion can be bracketed by a start/stop event pair recorded on the
// stream the kernel is launched on (so the pair measures that kernel, not the host).  Events come
// from a pre-allocated pool; reading them happens after the timed region (collect()), so enabling
// the timers adds two event records per launch and no synchronisation.  Each launch also reports
// its work units (voxels or pixels) so the harness can turn durations into bytes/s.
#pragma once

#include <array>
#include <vector>

#include "types.hpp"

namespace emf {

class KernelTimers {
public:
    enum Kind {
        Points = 0,   // computePoints
        Assoc,        // computeAssociation (per model)
        Normalize,    // normalizeAssociation / sumAssociation
        Raycast,      // raycastTSDF (per model)
        Composite,    // compositeRaycast (+ visibility)
        Integrate,    // updateTSDF (per model)
        Grads,        // computeTSDFGrads (materialised mode only)
        FgBg,         // updateFgBgProbs + computeFgProbs
        Track,        // one tracking stage (prepare + maxTrackingIter LM iterations)
        IntegrateBg,  // the background's out-of-place integration, concurrent with the raycast
        kNumKinds
    };
    struct Summary {
        uint64_t launches = 0;
        double total_ms = 0.0;
        double units = 0.0;  // voxels (volume sweeps) or pixels (image kernels), summed
    };

    ~KernelTimers();
    /** Allocate `maxLaunches` event pairs and start recording; 0 disables. */
    void enable(size_t maxLaunches);
    bool enabled() const { return !pairs.empty(); }
    /**
     * Which kinds get an event pair (bit k = Kind k; default: all).  Two event records around a
     * 5 us kernel cost more than the kernel: bench.py brackets only the large kernels by default.
     */
    void select(uint32_t mask) { kindMask = mask; }
    /**
     * Bracket only every n-th launch of a kind (n >= 1; default 1 = all).  Two timed event records cost the
     * stream they sit on a few microseconds each: around the two long kernels of a 0.58 ms frame that is 2.6 %
     * of the frame rate.  A sample of the launches gives the same average for a quarter of the disturbance.
     */
    void setStride(unsigned n) { stride = n ? n : 1; }
    /** Drop recorded launches, keep the pool. */
    void clear() { used = 0; dropped = 0; seen.fill(0u); }
    /** After the device is idle: per-kind launch count, summed duration and work units. */
    std::array<Summary, kNumKinds> collect() const;
    size_t droppedLaunches() const { return dropped; }

    class Scope {
    public:
        Scope(KernelTimers* t, Kind k, double units, hipStream_t s);
        ~Scope();
        Scope(const Scope&) = delete;
        Scope& operator=(const Scope&) = delete;

    private:
        KernelTimers* timers;
        hipStream_t stream;
        long slot;
    };
    Scope scope(Kind k, double units, const Stream& s) { return Scope(this, k, units, s.get()); }

private:
    struct Pair {
        hipEvent_t start = nullptr, stop = nullptr;
        Kind kind = Points;
        double units = 0.0;
    };
    std::vector<Pair> pairs;
    uint32_t kindMask = 0xffffffffu;
    unsigned stride = 1;
    std::array<unsigned, kNumKinds> seen{};
    size_t used = 0, dropped = 0;
};

}  // namespace emf
