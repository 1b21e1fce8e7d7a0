// EMFusionPerVolume.cpp -- emf::EMFusion: the per-volume fallback path -- one HIP stream per volume, host-side visibility gate (reference EMFusion.h:471, EMFusion.cpp:635-670, 726-795, 865-889).
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"

namespace emf {

using namespace detail;

void EMFusion::forkVolumeStreams() {
    // one stream per volume, as in the reference (EMFusion.h:471) -- created when the per-volume path first runs,
    // not with the volume: an instance on the batched path owns three streams, and every further stream of a
    // process makes it likelier that two of them share a hardware queue
    streamOf(0);
    for (auto& obj : objects) streamOf(obj.getID());
    main.record();
    for (auto& kv : streams) kv.second.waitOn(main);
}

void EMFusion::joinVolumeStreams() {
    for (auto& kv : streams) main.waitFor(kv.second);
}

// ---- per-volume path -------------------------------------------------------------------------------

void EMFusion::estepPerVolume() {
    const emf_image_t pv = points.view();
    forkVolumeStreams();
    {
        auto kt = ktimers.scope(KernelTimers::Assoc, pixels(), streamOf(0));
        background.computeAssociation(pv, pose, bg_associationWeights.view(), streamOf(0));
    }
    for (auto& obj : objects) {
        auto kt = ktimers.scope(KernelTimers::Assoc, pixels(), streamOf(obj.getID()));
        obj.computeAssociation(pv, pose, objImages.at(obj.getID()).associationWeights.view(),
                               streamOf(obj.getID()));
    }
    joinVolumeStreams();

    // normalisation: background first, then objects in ascending id (std::map) order
    std::vector<emf_image_t> maps;
    maps.push_back(bg_associationWeights.view());
    for (auto& kv : objImages) maps.push_back(kv.second.associationWeights.view());
    const emf_image_t nv = associationNorm.view();
    auto kt = ktimers.scope(KernelTimers::Normalize, pixels() * static_cast<double>(maps.size()),
                            main);
    if (!sharded) {
        emfCheck(emf_hip_normalizeAssociation(maps.data(), static_cast<int>(maps.size()),
                                              static_cast<int>(maps.size()), nullptr, &nv,
                                              main.abi()),
                 "normalizeAssociation");
        return;
    }
    const emf_image_t sv = objPartialSum.view();
    if (maps.size() > 1) {
        emfCheck(emf_hip_sumAssociation(maps.data() + 1, static_cast<int>(maps.size()) - 1, &sv,
                                        main.abi()),
                 "sumAssociation");
    } else {
        objPartialSum.setZero(main);
    }
    comm->allReduceSumF32(objPartialSum.ptr(), params.frameSize.area(), main);
    emfCheck(emf_hip_normalizeAssociation(maps.data(), static_cast<int>(maps.size()), 1, &sv, &nv,
                                          main.abi()),
             "normalizeAssociation");
}

void EMFusion::raycastPerVolume() {
    uint64_t* stats = statsOn ? raycastStatsDev.as<uint64_t>() : nullptr;
    forkVolumeStreams();
    {
        Stream& s = streamOf(0);
        bg_raylengths.setZero(s);
        bg_vertices.setZero(s);
        bg_normals.setZero(s);
        bg_mask.setZero(s);
        auto kt = ktimers.scope(KernelTimers::Raycast, pixels(), s);
        background.raycast(pose, params.intr, bg_raylengths.view(), bg_vertices.view(),
                           bg_normals.view(), bg_mask.view(), s, stats);
    }
    for (auto& obj : objects) {
        Stream& s = streamOf(obj.getID());
        ObjImages& im = objImages.at(obj.getID());
        im.raylengths.setZero(s);
        im.vertices.setZero(s);
        im.normals.setZero(s);
        im.modelSegmentation.setZero(s);
        auto kt = ktimers.scope(KernelTimers::Raycast, pixels(), s);
        obj.raycast(pose, params.intr, im.raylengths.view(), im.vertices.view(),
                    im.normals.view(), im.modelSegmentation.view(), s, stats);
    }
    joinVolumeStreams();
    stamp(kRaycast);
    compositeAndVisibility(false);
}

void EMFusion::integratePerVolume() {
    if (background.doubleBuffered()) bgBackStale = true;  // integrated in place below
    refreshVisibleFromDevice();
    forkVolumeStreams();
    const bool grads = gradMode == TSDF::Gradients::Materialized;
    const emf_image_t il = invLambda.view();
    const emf_image_t* ilp = useLambdaTable ? &il : nullptr;
    {
        auto kt = ktimers.scope(KernelTimers::Integrate,
                                static_cast<double>(background.voxels()), streamOf(0));
        background.integrate(depth, bg_associationWeights.view(), pose, params.intr, streamOf(0),
                             ilp);
    }
    if (grads) {
        auto kt = ktimers.scope(KernelTimers::Grads, static_cast<double>(background.voxels()),
                                streamOf(0));
        background.updateGradients(streamOf(0));
    }
    for (auto& obj : objects) {
        if (!vis_objs.count(obj.getID())) continue;
        Stream& s = streamOf(obj.getID());
        {
            auto kt = ktimers.scope(KernelTimers::Integrate, static_cast<double>(obj.voxels()), s);
            obj.integrate(depth, objImages.at(obj.getID()).associationWeights.view(), pose,
                          params.intr, s, ilp);
        }
        if (grads) {
            auto kt = ktimers.scope(KernelTimers::Grads, static_cast<double>(obj.voxels()), s);
            obj.updateGradients(s);
        }
    }
    joinVolumeStreams();
}

}  // namespace emf
