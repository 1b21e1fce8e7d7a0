// EMFusionTracking.cpp -- emf::EMFusion: the tracking driver (reference src/core/EMFusion.cpp:672-724, TSDF.cpp:170-344).
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace emf {

using namespace detail;

// ---- tracking -------------------------------------------------------------------------------------

namespace {
// Q of the QR decomposition of M with a positive diagonal of R -- what TSDF::prepareTracking's
// Householder QR + sign fix computes (TSDF.cpp:176-183) -- by Gram-Schmidt in double.
Matx33f orthonormalised(const Matx33f& M) {
    double c[3][3], q[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[j][i] = M(i, j);  // c[j] = column j
    for (int j = 0; j < 3; ++j) {
        double v[3] = {c[j][0], c[j][1], c[j][2]};
        for (int k = 0; k < j; ++k) {
            const double d = q[k][0] * c[j][0] + q[k][1] * c[j][1] + q[k][2] * c[j][2];
            for (int i = 0; i < 3; ++i) v[i] -= d * q[k][i];
        }
        const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) q[j][i] = v[i] / n;
    }
    Matx33f Q;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q(i, j) = static_cast<float>(q[j][i]);
    return Q;
}
}  // namespace

namespace {
// the pinned block the step kernel reports to: progress and done words, then (from this byte on) the states of models
// that are done
constexpr size_t kTrackFinalOffset = 256;
static_assert(kTrackFinalOffset >= sizeof(uint32_t) * (1 + EMF_MAX_BATCH), "room for the words");
}  // namespace

void EMFusion::trackModels(int first, int count) {
    if (count <= 0) return;
    if (!batched)
        throw HipError("EMFusion: tracking needs the batched path (on-the-fly gradients, EMF_PER_VOLUME unset)",
                       EMF_E_LIMIT);
    if (count > EMF_MAX_BATCH || first / EMF_MAX_BATCH != (first + count - 1) / EMF_MAX_BATCH) {
        // more models than one launch takes: the stage runs chunk by chunk of the table (the models of a stage are
        // independent of one another: reference EMFusion.cpp:689-723 tracks object after object)
        forChunks(first, first + count, [&](int from, int cnt) { trackModels(from, cnt); });
        return;
    }
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t per = emf_hip_trackScratchBytes(w, h);
    const size_t slots = modelsHost.size();
    if (trackScratch.bytes() < per * slots) {  // one block per table slot (storeTrackWeights reads them after the stage)
        main.waitForCompletion();
        trackScratch = DeviceBuffer(per * std::max(slots, std::min<size_t>(2 * slots, EMF_MAX_MODELS)));
    }
    if (trackStates.empty()) {
        trackStates = DeviceBuffer(sizeof(emf_track_state_t) * EMF_MAX_MODELS);
        hipCheck(hipHostMalloc(reinterpret_cast<void**>(&trackStatesHost),
                               sizeof(emf_track_state_t) * EMF_MAX_MODELS, hipHostMallocDefault),
                 "hipHostMalloc");
        // progress words the step kernel writes while the stream runs (emf_hip_trackStep)
        if (trackWindow > 0 &&
            (hipHostMalloc(reinterpret_cast<void**>(&trackWatch), kTrackFinalOffset + sizeof(emf_track_state_t) * EMF_MAX_BATCH,
                           hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
             hipHostGetDevicePointer(reinterpret_cast<void**>(&trackWatchDev), trackWatch, 0) != hipSuccess)) {
            (void)hipGetLastError();  // no device-visible host memory here: poll in chunks instead
            if (trackWatch) (void)hipHostFree(trackWatch);
            trackWatch = trackWatchDev = nullptr;
            trackWindow = 0;
        }
    }
    std::vector<emf_pose_t> co;
    posesCO(co);
    for (int m = first; m < first + count; ++m) {  // prepareTracking: re-orthonormalised rel_pose_CO
        const Matx33f Q = orthonormalised(Matx33f(co[m].R));
        for (int k = 0; k < 9; ++k) co[m].R[k] = Q.val[k];
    }
    emf_track_params_t tp;
    tp.huberThresh = params.tsdfParams.huberThresh;
    tp.maxWeight = params.tsdfParams.maxTSDFWeight;
    tp.tau = params.tsdfParams.tau;
    tp.eps1 = params.tsdfParams.eps1;
    tp.eps2 = params.tsdfParams.eps2;
    tp.nuInit = params.tsdfParams.nu_init;
    emf_track_state_t* states = trackStates.as<emf_track_state_t>() + first;
    const emf_image_t pv = points.view();
    {
        auto kt = ktimers.scope(KernelTimers::Track,
                                pixels() * count * params.maxTrackingIter, main);
        emfCheck(emf_hip_trackPrepare(states, co.data() + first, count, tp.nuInit, main.abi()),
                 "trackPrepare");
        char* const scratch = static_cast<char*>(trackScratch.data()) + per * first;
        if (trackWindow > 0) {
            // The loop needs the host only to stop enqueuing: one launch per LM iteration, kept
            // `trackWindow` launches ahead of the device, which reports -- into host memory, while the
            // stream runs -- how far it is and which models are done (LM converges in 20-60 of the
            // 100 iterations, differently in every frame).  The launches already enqueued when the
            // last model finishes return at once (~2 us each); the states are read back once.
            volatile uint32_t* watch = trackWatch;
            for (int i = 0; i <= count; ++i) watch[i] = 0u;
            const emf_track_state_t* const finalHost =
                reinterpret_cast<const emf_track_state_t*>(reinterpret_cast<const char*>(trackWatch) + kTrackFinalOffset);
            emf_track_state_t* const finalDev =
                reinterpret_cast<emf_track_state_t*>(reinterpret_cast<char*>(trackWatchDev) + kTrackFinalOffset);
            const int maxLaunches = 2 * params.maxTrackingIter + 4;  // (every step a speculation miss)
            const auto t0 = std::chrono::steady_clock::now();
            int launch = 0;
            // the stage's tag in the upper half of every sequence number and done word: the previous stage's last launches
            // may still be queued (nobody waits for them) and write their words after the reset above
            trackStageTag = (trackStageTag + 1u) & 0xffffu;
            if (trackStageTag == 0u) trackStageTag = 1u;
            const uint32_t tag = trackStageTag << 16;
            const auto progress = [&]() { const uint32_t w = watch[0]; return (w & 0xffff0000u) == tag ? static_cast<int>(w & 0xffffu) : 0; };
            const auto done = [&](int m) { const uint32_t w = watch[1 + m]; return (w & 0xffff0000u) == tag && (w & 3u) != 0u; };
            for (; launch < maxLaunches; ++launch) {
                for (unsigned spins = 0; launch - progress() >= trackWindow; ++spins)
                    if ((spins & 0xffffu) == 0xffffu &&
                        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
                        throw HipError("EMFusion: the tracking launches make no progress", EMF_E_ARG);
                bool all = launch > 0;
                for (int m = 0; m < count && all; ++m) all = done(m);
                if (all) break;
                emfCheck(emf_hip_trackStep(currentTable() + first, states, count, &pv, &tp, scratch, per, launch,
                                           params.maxTrackingIter, trackWatchDev, tag | static_cast<uint32_t>(launch + 1),
                                           finalDev, main.abi()),
                         "trackStep");
            }
            if (launch & 1)  // an even number of launches leaves the state in `states`
                emfCheck(emf_hip_trackStep(currentTable() + first, states, count, &pv, &tp, scratch, per, launch,
                                           params.maxTrackingIter, nullptr, 0u, nullptr, main.abi()),
                         "trackStep");
            bool all = true;
            for (int m = 0; m < count && all; ++m) all = done(m);
            if (all) {
                // every model's state arrived in front of its word: no copy command, no wait for the stream (the launches
                // still queued pass the states on and return)
                std::atomic_thread_fence(std::memory_order_acquire);
                std::memcpy(trackStatesHost + first, finalHost, sizeof(emf_track_state_t) * count);
            } else {  // (the launch budget ran out first)
                hipCheck(hipMemcpyAsync(trackStatesHost + first, states, sizeof(emf_track_state_t) * count,
                                        hipMemcpyDeviceToHost, main.get()),
                         "hipMemcpyAsync");
                main.waitForCompletion();
            }
            if (debugEnv("EMF_TRACK_LOG")) {  // diagnosis: launches against judged steps
                int it = 0, acc = 0;
                for (int m = first; m < first + count; ++m) {
                    it = std::max(it, trackStatesHost[m].iterations);
                    acc = std::max(acc, trackStatesHost[m].accepted);
                }
                std::fprintf(stderr, "track stage first %d count %d: launches %d, most steps %d, most accepted %d\n", first, count,
                             launch, it, acc);
            }
            return;
        }
        // Without the progress words (EMF_TRACK_WINDOW=0): iterations are enqueued in chunks and the
        // device-side states are polled once per chunk, only to stop enqueuing launches that would
        // return at once.  A chunk normally advances every model by its n iterations; after more
        // than one speculation miss (see emf_hip_trackIterate) by fewer -- the iteration counts come
        // back with the poll.
        // The first chunk is as long as the stage was in the last frame (+8): an idle launch costs
        // ~2 us, a poll ~50.
        const int chunk = trackChunk > 0 ? trackChunk : params.maxTrackingIter;
        int& predicted = trackPredicted[first == 0 ? 0 : 1];
        int taken = 0;
        // diagnosis (scripts/track_verdict_sequences.py): one line per poll and model, every chunk as long as asked
        static const bool logVerdicts = debugEnv("EMF_TRACK_LOG") != nullptr;
        for (int done = 0; done < params.maxTrackingIter;) {
            const int want = done == 0 && predicted > 0 && trackChunk > 0 && !logVerdicts ? std::max(chunk, predicted + 8) : chunk;
            const int n = std::min(want, params.maxTrackingIter - done);
            emfCheck(emf_hip_trackIterate(currentTable() + first, states, count, &pv, &tp, scratch, per, n,
                                          main.abi()),
                     "trackIterate");
            hipCheck(hipMemcpyAsync(trackStatesHost + first, states,
                                    sizeof(emf_track_state_t) * count, hipMemcpyDeviceToHost,
                                    main.get()),
                     "hipMemcpyAsync");
            main.waitForCompletion();
            bool all = true;
            done = params.maxTrackingIter;
            for (int m = first; m < first + count; ++m) {
                const emf_track_state_t& st = trackStatesHost[m];
                if (logVerdicts)
                    std::fprintf(stderr, "track model %d: iterations %d accepted %d rho %g mu %g nu %g converged %d\n", m,
                                 st.iterations, st.accepted, st.rho, st.mu, st.nu, st.converged);
                taken = std::max(taken, st.iterations);
                if (st.converged) continue;
                all = false;
                done = std::min(done, st.iterations);  // (judged steps; a pending trial is not counted yet)
            }
            if (all) break;
        }
        predicted = taken;
    }
}

void EMFusion::trackCamera() {
    trackModels(0, 1);
    const emf_track_state_t& st = trackStatesHost[0];
    const Affine3f rel(Matx33f(st.R), Vec3f(st.t[0], st.t[1], st.t[2]));
    pose = background.getPose() * rel;  // TSDF::syncTrack (TSDF.cpp:339-345)
    TrackResult r;
    r.iterations = st.iterations;
    r.accepted = st.accepted;
    r.converged = st.converged != 0;
    r.error = st.err;
    trackResults[0] = r;
}

void EMFusion::trackObjects() {
    const int n = static_cast<int>(objects.size());
    trackModels(1, n);
    int m = 1;
    for (auto& obj : objects) {
        const emf_track_state_t& st = trackStatesHost[m++];
        const Affine3f rel(Matx33f(st.R), Vec3f(st.t[0], st.t[1], st.t[2]));
        obj.setPose(pose * rel.inv());  // ObjTSDF::syncTrack (ObjTSDF.cpp:228-235)
        TrackResult r;
        r.iterations = st.iterations;
        r.accepted = st.accepted;
        r.converged = st.converged != 0;
        r.error = st.err;
        trackResults[obj.getID()] = r;
    }
}

const TrackResult* EMFusion::getTrackResult(int id) const {
    auto it = trackResults.find(id);
    return it == trackResults.end() ? nullptr : &it->second;
}

}  // namespace emf
