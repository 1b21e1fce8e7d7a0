// EMFusionLifecycle.cpp -- emf::EMFusion: objects from masks (reference src/core/EMFusion.cpp:329-557, 797-863, 891-989).
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"
#include "Readers.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>

namespace emf {

using namespace detail;

// runMaskRCNN with a mask path (reference EMFusion.cpp:383-389) + the label image getLastMasks hands out
void EMFusion::loadPreprocMasks(FrameInputs& in) {
    char name[32];
    std::snprintf(name, sizeof(name), "Mask%04d.plk", frameCount);
    PreprocMasks pm;
    int n = 0;
    {
        std::ifstream probe(maskPath + "/" + name, std::ios::binary);
        if (probe.good()) n = loadPreprocessedMasks(maskPath + "/" + name, pm);
    }
    const int w = params.frameSize.width, h = params.frameSize.height;
    if (n > 0 && (pm.width != w || pm.height != h))
        throw HipError(std::string("EMFusion::usePreprocMasks: ") + name + " holds masks of another size than the frames",
                       EMF_E_SHAPE);
    main.waitForCompletion();  // the previous mask frame's device copies are being replaced
    preprocMaskDev.clear();
    in.instanceMasks.clear();
    in.instanceScores.clear();
    // the reference's instance colours (MaskRCNN.cpp:290-301), index 0 = no instance
    static const unsigned char colors[31][3] = {
        {0, 0, 0},       {0, 0, 255},     {255, 0, 0},    {0, 255, 0},     {255, 26, 184},  {255, 211, 0},   {0, 131, 246},
        {0, 140, 70},    {167, 96, 61},   {79, 0, 105},   {0, 255, 246},   {61, 123, 140},  {237, 167, 255}, {211, 255, 149},
        {184, 79, 255},  {228, 26, 87},   {131, 131, 0},  {0, 255, 149},   {96, 0, 43},     {246, 131, 17},  {202, 255, 0},
        {43, 61, 0},     {0, 52, 193},    {255, 202, 131}, {0, 43, 96},    {158, 114, 140}, {79, 184, 17},   {158, 193, 255},
        {149, 158, 123}, {255, 123, 175}, {158, 8, 0}};
    lastMaskVis.assign(static_cast<size_t>(w) * h * 3, 0);
    lastMaskInstances = n;
    for (int k = 0; k < n; ++k) {
        preprocMaskDev.emplace_back(params.frameSize);
        preprocMaskDev.back().upload(pm.masks[k].data(), main);
        const unsigned char* c = colors[1 + k % 30];
        for (size_t i = 0; i < pm.masks[k].size(); ++i)
            if (pm.masks[k][i]) {
                lastMaskVis[3 * i] = c[0];
                lastMaskVis[3 * i + 1] = c[1];
                lastMaskVis[3 * i + 2] = c[2];
            }
    }
    main.waitForCompletion();  // (pm's host buffers go out of scope)
    for (auto& m : preprocMaskDev) in.instanceMasks.push_back(m.view());
    in.instanceScores = pm.scores;
}

// ---- object creation / matching from masks ---------------------------------------------------------

void EMFusion::ensureLifecycleBuffers() {
    if (!statsScratch.empty()) return;
    statsScratch = DeviceBuffer(emf_hip_pointStatsScratchBytes());
    statsDev = DeviceBuffer(sizeof(emf_point_stats_t));
    overlapDev = DeviceBuffer(513 * sizeof(uint32_t));
    massDev = DeviceBuffer(8 * emf_hip_maskAssociationMassBytes());
    static_assert(EMF_MAX_MODELS * sizeof(emf_mask_mass_t) >= 513 * sizeof(uint32_t), "the larger of the two uses");
    hipCheck(hipHostMalloc(&lifecycleHost, EMF_MAX_MODELS * sizeof(emf_mask_mass_t), hipHostMallocDefault),
             "hipHostMalloc");
}

emf_point_stats_t EMFusion::maskedStats(const emf_image_t& mask, const Affine3f& frame) {
    ensureLifecycleBuffers();
    const emf_image_t pv = points.view();
    emfCheck(emf_hip_maskedPointStats(&pv, &mask, frame.rotation().val, frame.translation().val,
                                      statsScratch.data(), statsDev.as<emf_point_stats_t>(),
                                      main.abi()),
             "maskedPointStats");
    hipCheck(hipMemcpyAsync(lifecycleHost, statsDev.data(), sizeof(emf_point_stats_t),
                            hipMemcpyDeviceToHost, main.get()),
             "hipMemcpyAsync");
    main.waitForCompletion();
    return *static_cast<emf_point_stats_t*>(lifecycleHost);
}

float EMFusion::volumeIOU(const ObjTSDF& obj, const Vec3f& p10, const Vec3f& p90) const {
    const Vec3f center = (p10 + p90) / 2.f;
    const Vec3f dims = p90 - p10;
    const float volSize = params.volPad * std::max(dims[0], std::max(dims[1], dims[2]));
    const Vec3f hv(volSize / 2, volSize / 2, volSize / 2);
    const Vec3f low_new = center - hv, high_new = center + hv;
    Vec3f low, high;
    obj.getCorners(low, high);
    const Vec3f prev = obj.getVolumeSize();
    const float vol = 1.f * prev[0] * prev[1] * prev[2];
    // pow(float, int) of the reference promotes to double (C++11 [c.math]); the float keeps its rounding
    const float vol_new = static_cast<float>(std::pow(static_cast<double>(volSize), 3));
    float vol_int = 1.f;
    for (int k = 0; k < 3; ++k) {
        const float d = std::min(high[k], high_new[k]) - std::max(low[k], low_new[k]);
        if (d < 0) return 0.f;  // no overlap
        vol_int = vol_int * d;
    }
    return vol_int / (vol_new + vol - vol_int);
}

int EMFusion::initNewObjVolume(const emf_image_t& mask) {
    if (sharded)
        throw HipError("EMFusion::initNewObjVolume: not available on the sharded path (an overlap "
                       "test needs every object's geometry on every rank)", EMF_E_ARG);
    // world frame first: the count decides whether anything else is needed (EMFusion.cpp:501-503)
    const emf_point_stats_t world_stats = maskedStats(mask, pose);
    if (static_cast<int>(world_stats.count) < params.visibilityThresh) return -1;
    for (const auto& obj : objects) {  // EMFusion.cpp:508-524
        const emf_point_stats_t s = maskedStats(mask, obj.getPose().inv() * pose);
        const float iou = volumeIOU(obj, Vec3f(s.p10[0], s.p10[1], s.p10[2]),
                                    Vec3f(s.p90[0], s.p90[1], s.p90[2]));
        if (iou > params.volIOUThresh) return -1;
    }
    const Vec3f p10(world_stats.p10[0], world_stats.p10[1], world_stats.p10[2]);
    const Vec3f p90(world_stats.p90[0], world_stats.p90[1], world_stats.p90[2]);
    const Vec3f center = (p10 + p90) / 2.f;
    const Vec3f off = center - pose.translation();
    // cv::norm accumulates the squares in double (EMFusion.cpp:531-533)
    const double o0 = off[0], o1 = off[1], o2 = off[2];
    if (std::sqrt(o0 * o0 + o1 * o1 + o2 * o2) > static_cast<double>(params.distanceThresh)) return -1;
    const Vec3f dims = p90 - p10;
    const float volSize = params.volPad * std::max(dims[0], std::max(dims[1], dims[2]));
    if (static_cast<int>(allIds.size()) >= EMF_MAX_MODELS - 1) {
        // every slot of the model table is live: this mask gets no volume (the frame loop goes on,
        // as the reference's would); addObject() itself keeps rejecting the explicit call
        std::fprintf(stderr, "EMFusion::initNewObjVolume: %d live objects, no new volume for this mask\n",
                     static_cast<int>(allIds.size()));
        return -1;
    }
    return addObject(center, volSize);
}

int EMFusion::matchSegmentation(const emf_image_t& mask, float& match_iou) {
    refreshVisibleFromDevice();
    ensureLifecycleBuffers();
    const emf_image_t seg = modelSegmentation.view();
    emfCheck(emf_hip_maskOverlap(&mask, &seg, overlapDev.as<uint32_t>(), main.abi()), "maskOverlap");
    hipCheck(hipMemcpyAsync(lifecycleHost, overlapDev.data(), 513 * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, main.get()),
             "hipMemcpyAsync");
    main.waitForCompletion();
    const uint32_t* c = static_cast<const uint32_t*>(lifecycleHost);
    int match_id = -1;
    for (const auto& obj : objects) {
        const int id = obj.getID();
        if (!vis_objs.count(id) || id > 255) continue;
        const float inter = static_cast<float>(c[1 + id]);
        const float uni = static_cast<float>(c[0] + c[257 + id] - c[1 + id]);
        const float iou = inter / uni;  // 0 / 0 = NaN never exceeds match_iou, as in the reference
        if (iou > match_iou) {
            match_iou = iou;
            match_id = id;
        }
    }
    return match_iou > params.matchIOUThresh ? match_id : -1;
}

std::map<int, emf_image_t> EMFusion::initOrMatchObjs(std::vector<emf_image_t>& segs,
                                                     std::vector<int>& assigned,
                                                     const std::vector<std::vector<double>>& scores) {
    if (sharded) throw HipError("EMFusion::initOrMatchObjs: not available on the sharded path", EMF_E_ARG);
    ensureLifecycleBuffers();
    std::map<int, emf_image_t> matches;
    std::vector<int> unmatched;
    assigned.assign(segs.size(), -1);
    const emf_image_t modelSeg = modelSegmentation.view();
    auto overlapCounts = [&](const emf_image_t& seg) -> const uint32_t* {
        emfCheck(emf_hip_maskOverlap(&seg, &modelSeg, overlapDev.as<uint32_t>(), main.abi()), "maskOverlap");
        hipCheck(hipMemcpyAsync(lifecycleHost, overlapDev.data(), 513 * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, main.get()),
                 "hipMemcpyAsync");
        main.waitForCompletion();
        return static_cast<const uint32_t*>(lifecycleHost);
    };
    // ---- matchSegmentation over all masks (EMFusion.cpp:417-444) ----
    for (size_t i = 0; i < segs.size(); ++i) {
        int matched = -1;
        if (frameCount > 0) {
            float new_iou = 0.f;
            matched = matchSegmentation(segs[i], new_iou);
            if (matched >= 0 && matches.count(matched)) {
                // a second mask for the same model: the better one becomes the match; THIS mask goes
                // on as unmatched either way (EMFusion.cpp:424-437).  Quirk Q20: when it replaced the
                // earlier match it is carved below against the match of that model -- itself, the
                // reference's matches[id] being a shallow GpuMat copy of seg_gpus[i] -- so the model
                // ends up matched to an all-zero mask.  Reproduced: matches[] holds views of the same
                // device buffers.
                const uint32_t* c = overlapCounts(matches[matched]);
                const float prev_iou = static_cast<float>(c[1 + matched]) /
                                       static_cast<float>(c[0] + c[257 + matched] - c[1 + matched]);
                if (new_iou > prev_iou) {
                    for (size_t k = 0; k < i; ++k)
                        if (assigned[k] == matched) assigned[k] = -1;
                    matches[matched] = segs[i];
                    assigned[i] = matched;
                }
                matched = -1;
            }
        }
        if (matched >= 0) {
            matches[matched] = segs[i];
            assigned[i] = matched;
        } else {
            unmatched.push_back(static_cast<int>(i));
        }
    }
    // ---- initObjsFromUnmatched (EMFusion.cpp:446-494) ----
    for (int i : unmatched) {
        for (const auto& obj : objects) {
            const int id = obj.getID();
            if (id > 255) continue;
            auto it = matches.find(id);
            emfCheck(emf_hip_carveMask(&segs[i], &modelSeg, id, it == matches.end() ? nullptr : &it->second,
                                       overlapDev.as<uint32_t>(), main.abi()),
                     "carveMask");
            hipCheck(hipMemcpyAsync(lifecycleHost, overlapDev.data(), 2 * sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, main.get()),
                     "hipMemcpyAsync");
            main.waitForCompletion();
            const uint32_t* c = static_cast<const uint32_t*>(lifecycleHost);
            // more than half of the mask belonged to an existing object: no new volume from it
            if (static_cast<float>(c[1]) / static_cast<float>(c[0]) < .5f)
                hipCheck(hipMemset2DAsync(segs[i].data, segs[i].pitch, 0, static_cast<size_t>(segs[i].width),
                                          static_cast<size_t>(segs[i].height), main.get()),
                         "hipMemset2DAsync");
        }
        const int id = initNewObjVolume(segs[i]);
        lastCreated.push_back(id);
        matches.insert(std::make_pair(id, segs[i]));  // even id == -1 (EMFusion.cpp:491); callers drop that key
        if (assigned[i] < 0) assigned[i] = id;        // a replacing mask keeps scoring its model (score_matches)
    }
    bool resized = false;
    for (auto& obj : objects) {  // EMFusion.cpp:358-369
        auto it = matches.find(obj.getID());
        if (it != matches.end()) {
            // score_matches (EMFusion.cpp:442, 492): the scores of the mask that ended up with this object
            for (size_t i = 0; i < assigned.size() && i < scores.size(); ++i)
                if (assigned[i] == obj.getID()) obj.updateClassProbs(scores[i]);
            const Vec3i before = obj.getVolumeRes();
            const Vec3f offset = updateObj(obj, it->second);
            if (poseLog) obj_pose_offsets[obj.getID()][frameCount] = offset;
            resized |= offset[0] != 0.f || offset[1] != 0.f || offset[2] != 0.f ||
                       before[0] != obj.getVolumeRes()[0];
        }
        obj.updateExProb(it != matches.end());
    }
    if (resized) rebuildModelTable();  // new buffers, new resolution, new pose
    return matches;
}

// Reference EMFusion::updateObj (EMFusion.cpp:827-863) without the class scores: percentiles of the
// object's surface (the vertex cloud of its mesh) united with the newly matched points, in the
// object's frame, decide whether the volume has to grow or move (ObjTSDF::resize).  No mesh is
// built: emf_hip_objectExtentStats streams the marching-cubes vertices into the selection.
Vec3f EMFusion::updateObj(ObjTSDF& obj, const emf_image_t& mask) {
    ensureLifecycleBuffers();
    if (maskedStats(mask, pose).count == 0) return Vec3f::all(0.f);  // no valid point under the mask
    const Affine3f frame = obj.getPose().inv() * pose;
    const emf_image_t pv = points.view();
    const Vec3i res = obj.getVolumeRes();
    emfCheck(emf_hip_objectExtentStats(&pv, &mask, frame.rotation().val, frame.translation().val,
                                       obj.tsdfPtr(), obj.weightsPtr(), obj.fgVolMaskPtr(), res.val,
                                       obj.getVoxelSize(), statsScratch.data(),
                                       statsDev.as<emf_point_stats_t>(), main.abi()),
             "objectExtentStats");
    hipCheck(hipMemcpyAsync(lifecycleHost, statsDev.data(), sizeof(emf_point_stats_t),
                            hipMemcpyDeviceToHost, main.get()),
             "hipMemcpyAsync");
    main.waitForCompletion();
    const emf_point_stats_t s = *static_cast<emf_point_stats_t*>(lifecycleHost);
    const Vec3f offset = obj.resize(Vec3f(s.p10[0], s.p10[1], s.p10[2]),
                                    Vec3f(s.p90[0], s.p90[1], s.p90[2]), params.volPad, main);
    // the pose may have moved with the volume centre (EMFusion.cpp:858-860)
    if (poseLog) obj_poses[obj.getID()][frameCount] = obj.getPose();
    return offset;
}

Vec3f EMFusion::updateObject(int id, const emf_image_t& mask) {
    if (sharded) throw HipError("EMFusion::updateObject: not available on the sharded path", EMF_E_ARG);
    for (auto& obj : objects)
        if (obj.getID() == id) {
            quiesce();
            refreshVisibleFromDevice();  // rebuildModelTable below uploads the gate from the host set
            const Vec3f offset = updateObj(obj, mask);
            if (poseLog) {  // several calls between two frames add up
                Vec3f& logged = obj_pose_offsets[id][frameCount];
                logged = logged + offset;
            }
            rebuildModelTable();
            return offset;
        }
    throw HipError("EMFusion::updateObject: no object " + std::to_string(id), EMF_E_ARG);
}

void EMFusion::deleteObj(int id) {  // reference EMFusion.cpp:982-989
    // the slot of a deleted object is free again: EMF_MAX_MODELS bounds the LIVE models, not the
    // number ever created (a long run spawns and cleans up spurious objects all the time)
    allIds.erase(std::remove(allIds.begin(), allIds.end(), id), allIds.end());
    streams.erase(id);
    objImages.erase(id);
    vis_objs.erase(id);
    trackResults.erase(id);
}

std::vector<int> EMFusion::cleanUpObjs(bool maskFrame, const std::map<int, emf_image_t>& matches) {
    if (sharded) throw HipError("EMFusion::cleanUpObjs: not available on the sharded path", EMF_E_ARG);
    std::set<int> spurious;
    if (maskFrame)
        for (const auto& obj : objects)
            if (obj.getExProb() < params.existenceThresh) spurious.insert(obj.getID());
    ensureLifecycleBuffers();
    // The association mass of EVERY object of this rank is enqueued before the host waits for anything: ONE
    // synchronisation per frame then yields both the visible set (it decides whose mass counts, EMFusion.cpp:936) and
    // the masses (rounds 3-5: one wait for the visible set, then a launch, a copy and a wait per visible object).
    const size_t nobj = objects.size();
    const size_t stride = emf_hip_maskAssociationMassBytes();
    if (massDev.bytes() < nobj * stride) massDev = DeviceBuffer(std::max(nobj, size_t(8)) * stride);
    emf_mask_mass_t* const massHost = static_cast<emf_mask_mass_t*>(lifecycleHost);  // pinned, EMF_MAX_MODELS entries
    size_t k = 0;
    for (const auto& obj : objects) {
        const ObjImages& im = objImages.at(obj.getID());
        const emf_image_t seg = im.modelSegmentation.view(), assoc = im.associationWeights.view();
        auto it = matches.find(obj.getID());
        emfCheck(emf_hip_maskAssociationMass(&seg, it == matches.end() ? nullptr : &it->second, &assoc,
                                             reinterpret_cast<emf_mask_mass_t*>(static_cast<char*>(massDev.data()) + k * stride),
                                             main.abi()),
                 "maskAssociationMass");
        ++k;
    }
    if (nobj)  // the answers (first entry of every object's block) in one strided copy
        hipCheck(hipMemcpy2DAsync(massHost, sizeof(emf_mask_mass_t), massDev.data(), stride, sizeof(emf_mask_mass_t), nobj,
                                  hipMemcpyDeviceToHost, main.get()),
                 "hipMemcpy2DAsync(mask masses)");
    main.waitForCompletion();
    refreshVisibleFromDevice();  // the host copy of vis_objs decides (the stream is idle: no further wait)
    k = 0;
    for (const auto& obj : objects) {
        const emf_mask_mass_t mm = massHost[k++];
        if (vis_objs.count(obj.getID()) && params.assocThresh * static_cast<float>(mm.count) > mm.sum) spurious.insert(obj.getID());
    }
    std::vector<int> deleted;
    for (auto it = objects.begin(); it != objects.end();) {
        const int id = it->getID();
        if (spurious.count(id) || !vis_objs.count(id)) {
            deleted.push_back(id);
            quiesce();  // nothing in flight may still use the volume
            deleteObj(id);
            if (poseLog && !(ignorePerson && isPerson(*it))) {
                meshes[id] = it->getMesh();  // saveOutput: EMFusion.cpp:962-966
                if (expVols) savedVolumes[id] = saveVolumes(*it);  // EMFusion.cpp:967-973
            }
            it = objects.erase(it);
        } else {
            ++it;
        }
    }
    if (!deleted.empty()) rebuildModelTable();
    return deleted;
}

void EMFusion::integrateMasks(const std::map<int, emf_image_t>& matches) {
    const emf_image_t segv = modelSegmentation.view();
    const emf_image_t occv = occludedMask.view();
    for (auto& obj : objects) {
        auto it = matches.find(obj.getID());
        if (it == matches.end()) continue;
        // pixels where this object's own raycast hit but another model is in front are not
        // used for the foreground statistics (reference EMFusion.cpp:897-900)
        const emf_image_t objSeg = objImages.at(obj.getID()).modelSegmentation.view();
        emfCheck(emf_hip_occludedMask(&objSeg, &segv, obj.getID(), &occv, main.abi()),
                 "occludedMask");
        auto kt = ktimers.scope(KernelTimers::FgBg, static_cast<double>(obj.voxels()), main);
        obj.integrateMask(it->second, occv, pose, params.intr, main);
    }
}

}  // namespace emf
