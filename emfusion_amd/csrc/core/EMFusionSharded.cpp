// EMFusionSharded.cpp -- emf::EMFusion: the cross-rank exchanges of the sharded path (no reference counterpart).
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"

#include <algorithm>
#include <exception>

namespace emf {

using namespace detail;

// The E-step of the sharded path: every rank computes the likelihood maps of ITS models, the per-pixel sum of the
// object maps is exchanged (the ONE all-reduce SURVEY 8e / north_star name: reference EMFusion.cpp:653-665 sums all
// maps), then every rank normalises its own maps with the joint sum.
void EMFusion::estepSharded(const std::vector<emf_pose_t>& co, bool fromDepth) {
    const int n = static_cast<int>(co.size());
    const emf_image_t pv = points.view(), nv = associationNorm.view(), sv = objPartialSum.view();
    std::vector<emf_image_t> maps;
    maps.push_back(bg_associationWeights.view());
    for (auto& obj : objects) maps.push_back(objImages.at(obj.getID()).associationWeights.view());
    if (peerFused && maps.size() <= 16) {
        // direct peer writes: the E-step's kernel stores its partial sum straight into the peers' slots, and ONE
        // more launch waits for the peers, sums the slots in rank order and normalises -- two launches per E-step
        // where the unsharded frame has one (round 3: five)
        const uint32_t seq = comm->beginPeerExchange(main);
        {
            auto kt = ktimers.scope(KernelTimers::Assoc, pixels() * n, main);
            emfCheck(emf_hip_estepBatchedPeer(currentTable(), co.data(), n, fromDepth ? &depth : nullptr, params.intr.val, &pv,
                                              comm->peerGroup(), seq, main.abi()),
                     "estepBatchedPeer");
        }
        auto kt = ktimers.scope(KernelTimers::Normalize, pixels() * maps.size(), main);
        emfCheck(emf_hip_peerNormalizeAssociation(comm->peerGroup(), seq, maps.data(), static_cast<int>(maps.size()), &sv,
                                                  &nv, main.abi()),
                 "peerNormalizeAssociation");
        return;
    }
    // sharded objects: likelihoods + local object partial in one launch, ONE all-reduce over
    // xGMI, then every rank normalises its own maps
    if (n <= EMF_MAX_BATCH) {
        launchEstep(co, 0, n, fromDepth, 0, nullptr, &sv);
    } else {  // chunks of the table, then the local partial over all object maps in the same (table) order
        forChunks(0, n, [&](int first, int count) { launchEstep(co, first, count, fromDepth && first == 0, 0, nullptr, nullptr); });
        emfCheck(emf_hip_sumAssociation(maps.data() + 1, static_cast<int>(maps.size()) - 1, &sv, main.abi()), "sumAssociation");
    }
    // (Measured and dropped, round 3: the frame's LAST all-reduce + normalisation on a stream of their own beside
    // the raycast -- they feed the integrations only.  With a 30 us latency model the frame got no shorter: the
    // background's sweep needs the normalised weights and is as long as the raycast it runs beside.)
    comm->allReduceSumF32(objPartialSum.ptr(), params.frameSize.area(), main);
    {
        auto kt = ktimers.scope(KernelTimers::Normalize, pixels() * maps.size(), main);
        emfCheck(emf_hip_normalizeAssociation(maps.data(), static_cast<int>(maps.size()), 1, &sv, &nv, main.abi()),
                 "normalizeAssociation");
    }
}

// Object volumes are sharded over ranks: merge the nearest hit of ALL objects with one
// all-reduce(min) of packed (raylength, list position) keys, then finish the composite locally.
// Every rank ends up with the same segmentation and the visibility counts of all objects.
void EMFusion::compositeAcrossRanks(bool deviceGate) {
    const int w = params.frameSize.width, h = params.frameSize.height;
    std::vector<int32_t> listPos;
    std::vector<emf_image_t> oray, overt, onorm, oseg;
    for (auto& obj : objects) {
        ObjImages& im = objImages.at(obj.getID());
        const auto it = std::find(allIds.begin(), allIds.end(), obj.getID());
        listPos.push_back(static_cast<int32_t>(it - allIds.begin()));
        oray.push_back(im.raylengths.view());
        overt.push_back(im.vertices.view());
        onorm.push_back(im.normals.view());
        oseg.push_back(im.modelSegmentation.view());
    }
    const int nlocal = static_cast<int>(listPos.size());
    const int nall = static_cast<int>(allIds.size());
    const emf_image_t v_bgRay = bg_raylengths.view(), v_bgVert = bg_vertices.view(),
                      v_bgNorm = bg_normals.view(), v_bgMask = bg_mask.view(),
                      v_ray = raylengths.view(), v_vert = vertices.view(),
                      v_norm = normals.view(), v_seg = modelSegmentation.view(),
                      v_diff = diffRaylengths.view(), v_noObj = noObjMask.view();
    std::vector<int32_t> countIndex(1, 0);
    countIndex.insert(countIndex.end(), listPos.begin(), listPos.end());
    if (peerFused) {
        // direct peer writes: k_pack_keys_peer stores the keys and this rank's band of the background raycast
        // straight into the peers' slots; ONE more launch waits, takes the minimum key, fetches the foreign bands,
        // composites and counts visibility; a one-workgroup launch turns the counts into the gate (as unsharded)
        auto kt = ktimers.scope(KernelTimers::Composite, pixels() * (1.0 + nlocal), main);
        const uint32_t seq = comm->beginPeerExchange(main);
        const int band = bandRowsPending;
        const int row0 = std::min(rank * band, h);
        emfCheck(emf_hip_packHitKeysPeer(nlocal, listPos.data(), oray.data(), oseg.data(), &v_bgRay, &v_bgMask, row0,
                                         band ? std::min(band, h - row0) : 0, comm->peerGroup(), seq, main.abi()),
                 "packHitKeysPeer");
        if (!visCountsClear) visCounts.setZero(main);
        emfCheck(emf_hip_compositeFromKeysPeer(comm->peerGroup(), seq, band, nall, allIds.data(), nlocal, listPos.data(),
                                               oray.data(), overt.data(), onorm.data(), &v_bgRay, &v_bgVert, &v_bgNorm,
                                               &v_bgMask, &v_ray, &v_vert, &v_norm, &v_seg, &v_diff, &v_noObj,
                                               params.boundary, visCounts.as<int32_t>(), main.abi()),
                 "compositeFromKeysPeer");
        bandRowsPending = 0;
        emfCheck(emf_hip_visibilityFlagsMirror(visCounts.as<int32_t>(), nall, nlocal + 1, countIndex.data(),
                                               params.visibilityThresh, visibleDev.as<int32_t>(),
                                               deviceGate ? visibleHost : visCountsHost, main.abi()),
                 "visibilityFlagsMirror");
        visCountsClear = true;
    } else {
        auto kt = ktimers.scope(KernelTimers::Composite, pixels() * (1.0 + nlocal), main);
        emfCheck(emf_hip_packHitKeys(nlocal, listPos.data(), oray.data(), oseg.data(),
                                     hitKeys.as<uint64_t>(), w, h, main.abi()),
                 "packHitKeys");
        // ONE exchange per raycast: nearest-hit keys of the objects + the ranks' bands of the background's
        // raylengths and hit mask (ncclGroup: a single launch on the transport)
        struct Group {  // closes the group also when a collective inside throws
            Communicator& c;
            const int unwinding = std::uncaught_exceptions();
            explicit Group(Communicator& comm_) : c(comm_) { c.groupStart(); }
            ~Group() noexcept(false) {
                if (std::uncaught_exceptions() == unwinding) {
                    c.groupEnd();
                } else {
                    try { c.groupEnd(); } catch (...) {}
                }
            }
        };
        {
            Group group(*comm);
            comm->allReduceMinU64(hitKeys.as<uint64_t>(), params.frameSize.area(), main);
            if (bandRowsPending) {
                comm->gatherRowBands(bg_raylengths.ptr(), static_cast<size_t>(w) * sizeof(float), bandRowsPending, h, main);
                comm->gatherRowBands(bg_mask.ptr(), static_cast<size_t>(w), bandRowsPending, h, main);
            }
        }
        bandRowsPending = 0;
        visCountsClear = false;
        emfCheck(emf_hip_compositeFromKeys(hitKeys.as<uint64_t>(), nall, allIds.data(), nlocal,
                                           listPos.data(), oray.data(), overt.data(),
                                           onorm.data(), &v_bgRay, &v_bgVert, &v_bgNorm,
                                           &v_bgMask, &v_ray, &v_vert, &v_norm, &v_seg, &v_diff,
                                           &v_noObj, params.boundary, visCounts.as<int32_t>(),
                                           main.abi()),
                 "compositeFromKeys");
        if (deviceGate) {
            emfCheck(emf_hip_visibilityFlagsIndexed(visCounts.as<int32_t>(), nlocal + 1,
                                                    countIndex.data(), params.visibilityThresh,
                                                    visibleDev.as<int32_t>(), main.abi()),
                     "visibilityFlagsIndexed");
        }
    }
    stamp(kComposite);
    vis_objs.clear();
    visPending = false;
    if (nall == 0) return;
    if (!peerFused) {  // (the fused path's flag kernel has mirrored the counts already)
        int32_t* dst = deviceGate ? visibleHost : visCountsHost;
        hipCheck(hipMemcpyAsync(dst, visCounts.data(), sizeof(int32_t) * nall, hipMemcpyDeviceToHost,
                                main.get()),
                 "visCounts D2H");
    }
    if (deviceGate) {
        visIds = allIds;
        visPending = true;
        return;
    }
    main.waitForCompletion();
    for (int k = 0; k < nall; ++k)
        if (visCountsHost[k] > params.visibilityThresh) vis_objs.insert(allIds[k]);
}

}  // namespace emf
