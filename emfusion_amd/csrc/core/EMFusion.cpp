// EMFusion.cpp -- emf::EMFusion: construction, model table and the per-frame schedule of the batched path
// (see EMFusion.hpp).  Reference: src/core/EMFusion.cpp:28-129, 635-670, 726-795, 865-889.
//
// The class is spread over six translation units:
//   EMFusion.cpp           construction, model table, processFrame / runSchedule, the batched path's stages
//   EMFusionLifecycle.cpp  object creation / matching / resizing / clean-up from masks, mask integration (SURVEY f-3)
//   EMFusionTracking.cpp   device-resident LM-ICP driver (SURVEY f-1)
//   EMFusionSharded.cpp    cross-rank exchanges of the sharded path
//   EMFusionCapture.cpp    poses / meshes / volumes / debug images / rendering (SURVEY f-4)
//   EMFusionPerVolume.cpp  the reference-structured fallback: one stream per volume, host visibility gate
//
// Two execution paths produce the same results:
//   batched  (default)  one launch per stage for all models of this rank, driven by a
//            device-resident model table; the visibility gate of integrateDepth is evaluated on
//            the device, so a frame contains no host synchronisation at all
//   per-volume (fallback: volumes whose Nx is not a multiple of
//            4, materialised gradient volumes, or EMF_PER_VOLUME=1)  the reference's structure:
//            one HIP stream per volume joined by events, host-side visibility gate
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"
#include "Readers.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace emf {

using namespace detail;

EMFusion::EMFusion(const Params& _params, TSDF::Gradients gradients,
                   std::shared_ptr<Communicator> _comm)
    : params(_params),
      gradMode(gradients),
      comm(std::move(_comm)),
      background(_params.globalVolumeDims, _params.globalVoxelSize,
                 _params.globalRelTruncDist * _params.globalVoxelSize, _params.volumePose,
                 _params.tsdfParams, _params.frameSize, gradients),
      depthFiltered(_params.frameSize),
      invLambda(_params.frameSize),
      points(_params.frameSize),
      raylengths(_params.frameSize),
      bg_raylengths(_params.frameSize),
      associationNorm(_params.frameSize),
      bg_associationWeights(_params.frameSize),
      diffRaylengths(_params.frameSize),
      objPartialSum(_params.frameSize),
      vertices(_params.frameSize),
      normals(_params.frameSize),
      bg_vertices(_params.frameSize),
      bg_normals(_params.frameSize),
      modelSegmentation(_params.frameSize),
      bg_mask(_params.frameSize),
      noObjMask(_params.frameSize),
      occludedMask(_params.frameSize),
      visCounts(sizeof(int32_t) * EMF_MAX_MODELS),
      raycastStatsDev(4 * sizeof(uint64_t)),
      modelTable(2 * sizeof(emf_model_t) * EMF_MAX_MODELS),
      visibleDev(sizeof(int32_t) * EMF_MAX_MODELS),
      integrateStatsDev(sizeof(uint64_t)) {
    if (comm) {
        rank = comm->rank();
        world = comm->size();
        hitKeys = DeviceBuffer(params.frameSize.area() * sizeof(uint64_t));
    }
    ignorePerson = params.ignore_person;  // data.h:198; config/tum.cfg sets it
    const char* env = std::getenv("EMF_PER_VOLUME");
    forceLegacy = env && env[0] == '1';
    // EMF_LAMBDA_TABLE=0: integrate with the inline 1 / lambda (A/B measurements; same results)
    // EMF_BG_BANDS=0: every rank raycasts the whole (replicated) background itself
    const char* bb = std::getenv("EMF_BG_BANDS");
    bgBands = !(bb && bb[0] == '0');
    // EMF_INT_CULL=0: one-level integration launch (every tile gets a workgroup and culls itself)
    const char* ic = std::getenv("EMF_INT_CULL");
    cullBoxes = !(ic && ic[0] == '0');
    // Switches read through debugEnv() exist in builds with -DEMF_DEBUG_SWITCHES only: their A/B is on record as lost.
    // EMF_OBJ_CULL=1: the two-level launch also for the objects alone (A/B: measured slower for 4 and for 8 volumes of 128^3)
    const char* oc = debugEnv("EMF_OBJ_CULL");
    objCull = oc && oc[0] == '1';
    // EMF_TRACK_CHUNK: LM iterations enqueued between two polls of the convergence flags
    if (const char* tc = debugEnv("EMF_TRACK_CHUNK")) trackChunk = std::atoi(tc);
    if (const char* tw = debugEnv("EMF_TRACK_WINDOW")) trackWindow = std::atoi(tw);
    if (const char* fp = debugEnv("EMF_FUSE_POINTS")) fusePoints = fp[0] != '0';
    if (const char* fv = debugEnv("EMF_FUSE_VISIBILITY")) fuseVisibility = fv[0] != '0';
    if (const char* ef = debugEnv("EMF_EARLY_FAR_BOUNDS")) earlyFarBounds = ef[0] != '0';
    // EMF_BG_OVERLAP=0: integrate the background in place after the raycast, as the reference does
    const char* bo = std::getenv("EMF_BG_OVERLAP");
    bgOverlap = !(bo && bo[0] == '0');
    // EMF_FAR_BOUNDS=0: no far bounds for the raycast (A/B measurements; same results)
    const char* fb = std::getenv("EMF_FAR_BOUNDS");
    useFarBounds = !(fb && fb[0] == '0');
    // EMF_RAY_FOOTPRINTS=0: every object gets a marching workgroup for every tile of the image
    const char* rf = debugEnv("EMF_RAY_FOOTPRINTS");
    useFootprints = !(rf && rf[0] == '0');
    // EMF_MARCH_ROWS = 1 (default) / 2 / 4 lanes per BACKGROUND ray (march_lane / march_quad<2> / march_quad<4>): same
    // images (A/B measurements); read here, once -- anything else is refused, not silently marched with one lane
    if (const char* mr = std::getenv("EMF_MARCH_ROWS")) {
        marchLanes = std::atoi(mr);
        if (marchLanes != 1 && marchLanes != 2 && marchLanes != 4)
            throw HipError(std::string("EMFusion: EMF_MARCH_ROWS=") + mr + " (1, 2 or 4 lanes per background ray)", EMF_E_ARG);
    }
    const char* au = std::getenv("EMF_ASYNC_UPLOAD");
    asyncUpload = !(au && au[0] == '0');
    const char* lt = std::getenv("EMF_LAMBDA_TABLE");
    useLambdaTable = !(lt && lt[0] == '0');
    // sharded mode: the two cross-rank exchanges are used.  EMF_FORCE_SHARDED=1 turns it on for a
    // 1-rank communicator too, so the whole exchange path can be exercised on a single GPU.
    const char* fs = std::getenv("EMF_FORCE_SHARDED");
    sharded = comm && (world > 1 || (fs && fs[0] == '1'));
    // Over a direct peer-write transport the sharded path's exchanges are fused into the kernels around them
    // (Communicator::peerGroup); EMF_PEER_FUSED=0 keeps the transport's own two-launch collectives (A/B; same bits)
    if (sharded) {
        const char* pf = debugEnv("EMF_PEER_FUSED");
        const emf_peer_t* pg = comm->peerGroup();
        peerFused = pg && !(pf && pf[0] == '0') &&
                    pg->slotBytes >= emf_hip_peerRaycastSlotBytes(params.frameSize.width, params.frameSize.height);
    }
    hipCheck(hipHostMalloc(reinterpret_cast<void**>(&visCountsHost),
                           sizeof(int32_t) * EMF_MAX_MODELS, hipHostMallocDefault),
             "hipHostMalloc");
    hipCheck(hipHostMalloc(reinterpret_cast<void**>(&visibleHost),
                           sizeof(int32_t) * EMF_MAX_MODELS, hipHostMallocDefault),
             "hipHostMalloc");
    stamps.resize(kNumStamps);
    for (auto& e : stamps) hipCheck(hipEventCreate(&e), "hipEventCreate");
    Stream& s = Stream::Null();
    // reference EMFusion.cpp:44-55: bg_mask = 0, noObjMask = 1, bg_associationWeights = 1;
    // diffRaylengths is uninitialised memory in the reference, defined as 0 here (Q12)
    bg_mask.setZero(s);
    noObjMask.setTo(1, s);
    bg_associationWeights.setTo(1.f, s);
    diffRaylengths.setZero(s);
    modelSegmentation.setZero(s);
    raylengths.setZero(s);
    vertices.setZero(s);
    normals.setZero(s);
    bg_raylengths.setZero(s);
    associationNorm.setZero(s);  // read by getters before the first E-step (frame 0 has none)
    objPartialSum.setZero(s);
    raycastStatsDev.setZero(s);
    integrateStatsDev.setZero(s);
    visCounts.setZero(s);
    visibleDev.fill32(1u, s);  // background and freshly created objects integrate (Q18)
    {
        emf_image_t il = invLambda.view();
        emfCheck(emf_hip_computeInvLambda(params.intr.val, &il, s.abi()), "computeInvLambda");
    }
    s.waitForCompletion();
    rebuildModelTable();
    // volumes created from here on (objects, inside frames) do not wait for a reciprocal check
    TSDF::deferReciprocalChecks(true);
}

EMFusion::~EMFusion() {
    (void)hipDeviceSynchronize();
    for (auto& e : stamps) (void)hipEventDestroy(e);
    for (auto& u : uploadSlots) {
        if (u.pinned) (void)hipHostFree(u.pinned);
        if (u.copied) (void)hipEventDestroy(u.copied);
        if (u.frameDone) (void)hipEventDestroy(u.frameDone);
    }
    if (visCountsHost) (void)hipHostFree(visCountsHost);
    if (visibleHost) (void)hipHostFree(visibleHost);
    if (trackStatesHost) (void)hipHostFree(trackStatesHost);
    if (trackWatch) (void)hipHostFree(trackWatch);
    if (lifecycleHost) (void)hipHostFree(lifecycleHost);
}

void EMFusion::reset() {
    synchronize();
    pose = Affine3f::Identity();
    background.reset(params.volumePose);
    objects.clear();
    objImages.clear();
    allIds.clear();
    vis_objs.clear();
    visPending = false;
    for (auto it = streams.begin(); it != streams.end();)
        it = it->first == 0 ? std::next(it) : streams.erase(it);
    frameCount = 0;
    nextId = 1;
    bgInFlight = false;
    bgBackStale = false;
    bgPrepared = false;
    bgListPending = false;
    trackPredicted[0] = trackPredicted[1] = 0;
    Stream& s = Stream::Null();
    bg_associationWeights.setTo(1.f, s);
    diffRaylengths.setZero(s);
    modelSegmentation.setZero(s);
    visibleDev.fill32(1u, s);
    s.waitForCompletion();
    rebuildModelTable();
}

Stream& EMFusion::streamOf(int key) {
    auto it = streams.find(key);
    if (it == streams.end()) it = streams.emplace(key, Stream()).first;
    return it->second;
}

bool EMFusion::ownsObject(int id) const { return ownerOf(id, world) == rank; }

ObjTSDF* EMFusion::findObject(int id) {
    for (auto& o : objects)
        if (o.getID() == id) return &o;
    return nullptr;
}

const DeviceImage<float>* EMFusion::getObjAssociation(int id) const {
    auto it = objImages.find(id);
    return it == objImages.end() ? nullptr : &it->second.associationWeights;
}

const DeviceImage<float>* EMFusion::getObjRaylengths(int id) const {
    auto it = objImages.find(id);
    return it == objImages.end() ? nullptr : &it->second.raylengths;
}

int EMFusion::addObject(const Vec3f& center, float volSize) {
    return addObject(center, volSize, params.objVolumeDims);
}

int EMFusion::addObject(const Vec3f& center, float volSize, const Vec3i& res) {
    if (static_cast<int>(allIds.size()) >= EMF_MAX_MODELS - 1)
        throw HipError("EMFusion::addObject: too many live objects", EMF_E_LIMIT);
    quiesce();  // object creation changes the model table: nothing of this instance may be in flight
    refreshVisibleFromDevice();
    const int id = nextId++;
    allIds.push_back(id);
    // new objects are aligned with the world frame; only the centre matters for the pose
    // (reference EMFusion.cpp:541-547)
    if (ownsObject(id)) {
        const Affine3f obj_pose(Matx33f::eye(), center);
        const float vox = volSize / static_cast<float>(res[0]);
        // truncation distance as the reference forms it: (objRelTruncDist * volSize) / res (EMFusion.cpp:545-547)
        objects.emplace_back(id, res, vox, params.objRelTruncDist * volSize / static_cast<float>(res[0]), obj_pose,
                             params.tsdfParams, params.frameSize, gradMode);
        createObj(id);
    }
    vis_objs.insert(id);  // a new object integrates its first frame (Q18)
    rebuildModelTable();
    return id;
}

void EMFusion::createObj(int id) {
    // reference EMFusion.cpp:908-920
    ObjImages im;
    im.raylengths = DeviceImage<float>(params.frameSize);
    im.vertices = DeviceImage<float, 3>(params.frameSize);
    im.normals = DeviceImage<float, 3>(params.frameSize);
    im.modelSegmentation = DeviceImage<uint8_t>(params.frameSize);
    im.associationWeights = DeviceImage<float>(params.frameSize);
    Stream& s = Stream::Null();
    im.modelSegmentation.setZero(s);
    im.associationWeights.setTo(1.f, s);
    im.raylengths.setZero(s);
    im.vertices.setZero(s);
    im.normals.setZero(s);
    s.waitForCompletion();
    objImages.emplace(id, std::move(im));
}

// Rebuild and upload the device model table: slot 0 = background, then this rank's objects in
// creation order.  Called with the device idle (construction, addObject, reset).
void EMFusion::rebuildModelTable() {
    modelsHost.clear();
    resHost.clear();
    aux.waitForCompletion();  // the background's integration / the list rebuilds may still be running
    lists.waitForCompletion();
    if (useFarBounds) {  // sign maps that something other than the tile integration made stale
        background.refreshSignMaps();
        for (auto& obj : objects) obj.refreshSignMaps();
    }
    emf_model_t m{};
    background.describe(m);
    m.assoc = bg_associationWeights.ptr();
    m.raylengths = bg_raylengths.ptr();
    m.vertices = bg_vertices.ptr();
    m.normals = bg_normals.ptr();
    m.hitMask = bg_mask.ptr();
    modelsHost.push_back(m);
    for (auto& obj : objects) {
        emf_model_t o{};
        obj.describe(o);
        ObjImages& im = objImages.at(obj.getID());
        o.assoc = im.associationWeights.ptr();
        o.raylengths = im.raylengths.ptr();
        o.vertices = im.vertices.ptr();
        o.normals = im.normals.ptr();
        o.hitMask = im.modelSegmentation.ptr();
        modelsHost.push_back(o);
    }
    bool tiled = true;
    voxelHost.clear();
    scanSlot.clear();
    listSlot.clear();
    anyScan = false;
    for (const auto& md : modelsHost) {
        // A volume too small for a relevant-tile list (an object: < 8192 tiles) could have its sign maps scanned
        // for far bounds; its rays are short anyway and the scan (23 us beside the E-steps, which it slows from
        // 12 to 37 us) costs the frame more than the cut saves the raycast: 0.6095 vs 0.5956 ms.  EMF_FAR_SCAN=1
        // scans them.
        const char* fsc = debugEnv("EMF_FAR_SCAN");
        const bool scanSmall = fsc && fsc[0] == '1';
        scanSlot.push_back(md.signMaps && !md.relevantTiles && scanSmall);
        listSlot.push_back(md.signMaps && md.relevantTiles);
        anyScan = anyScan || scanSlot.back();
        voxelHost.push_back(md.voxelSize);
        resHost.insert(resHost.end(), md.res, md.res + 3);
        tiled &= md.res[0] % 4 == 0;
    }
    // scratch of the two-level integration launch (every model on float4 tiles, no brick flags to keep): for the
    // table chunk that holds the background -- the launch exists for the background's sake (integrateBatched)
    integrateCullScratch = DeviceBuffer();
    if (cullBoxes && tiled && TSDF::brickFlagMode() == 0)
        integrateCullScratch = DeviceBuffer(emf_hip_integrateCullScratchBytes(
            resHost.data(), std::min(static_cast<int>(modelsHost.size()), EMF_MAX_BATCH)));
    // any number of models: stages are launched per chunk of EMF_MAX_BATCH table slots (forChunks)
    batched = !forceLegacy && gradMode == TSDF::Gradients::OnTheFly;
    farBounds = DeviceBuffer();
    if (batched && useFarBounds)
        farBounds = DeviceBuffer(2 * emf_hip_raycastFarBoundBytes(static_cast<int>(modelsHost.size()),  // two halves, see computeFarBounds
                                                                  params.frameSize.width, params.frameSize.height));
    if (batched && !integrateCullScratch.empty() && bgOverlap && bgCullScratch.empty()) {
        // the background gets its second copy the first time the two-level launch is usable
        background.enableDoubleBuffer();
        bgCullScratch = DeviceBuffer(emf_hip_integrateCullScratchBytes(resHost.data(), 1));
    }
    tableSel = 0;
    if (batched) {
        hipCheck(hipMemcpy(modelTable.data(), modelsHost.data(),
                           modelsHost.size() * sizeof(emf_model_t), hipMemcpyHostToDevice),
                 "model table upload");
        if (background.doubleBuffered()) {
            std::vector<emf_model_t> alt = modelsHost;
            const emf_volume_out_t back = background.backBuffers();
            alt[0].tsdf = back.tsdf;
            alt[0].weights = back.weights;
            hipCheck(hipMemcpy(modelTable.as<emf_model_t>() + EMF_MAX_MODELS, alt.data(),
                               alt.size() * sizeof(emf_model_t), hipMemcpyHostToDevice),
                     "model table upload");
        }
        // device gate: keep what the last raycast decided, new slots start visible
        std::vector<int32_t> vis(modelsHost.size(), 0);
        vis[0] = 1;
        size_t slot = 1;
        for (auto& obj : objects) vis[slot++] = vis_objs.count(obj.getID()) ? 1 : 0;
        hipCheck(hipMemcpy(visibleDev.data(), vis.data(), vis.size() * sizeof(int32_t),
                           hipMemcpyHostToDevice),
                 "visibility upload");
        if (useFarBounds) {  // lists of new / rebuilt sign maps
            forChunks(0, static_cast<int>(modelsHost.size()), [&](int first, int count) {
                emfCheck(emf_hip_updateRelevantTiles(modelTable.as<emf_model_t>() + first, resHost.data() + 3 * first,
                                                     count, Stream::Null().abi()),
                         "updateRelevantTiles");
            });
            Stream::Null().waitForCompletion();
        }
    }
}

// Verdicts of deferred reciprocal checks (TSDF::pollReciprocal) that have come in: from this frame on those
// volumes are marched with the reciprocal instead of the division -- same results, so a launch that reads the
// table while it is patched sees either form.  One small copy per adopted volume; no wait.
void EMFusion::adoptReciprocals() {
    auto adopt = [&](TSDF& vol, size_t slot) {
        if (!vol.pollReciprocal() || !batched || slot >= modelsHost.size()) return;
        modelsHost[slot].rcpVoxel = vol.reciprocal();
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !background.doubleBuffered()) break;
            emf_model_t* row = modelTable.as<emf_model_t>() + t * EMF_MAX_MODELS + slot;
            hipCheck(hipMemcpyAsync(&row->rcpVoxel, &modelsHost[slot].rcpVoxel, sizeof(float),
                                    hipMemcpyHostToDevice, main.get()),
                     "reciprocal patch");
        }
    };
    adopt(background, 0);  // (a second instance in the process: its background was created with deferral on)
    size_t slot = 1;
    for (auto& obj : objects) adopt(obj, slot++);
}

void EMFusion::settleReciprocals() {
    background.settleReciprocal();
    for (auto& obj : objects) obj.settleReciprocal();
    adoptReciprocals();
}

void EMFusion::posesCO(std::vector<emf_pose_t>& out) const {
    out.clear();
    out.push_back(toPose(background.getPose().inv() * pose));  // reference TSDF.cpp:141,162
    for (const auto& obj : objects) out.push_back(toPose(obj.getPose().inv() * pose));
}

void EMFusion::posesOC(std::vector<emf_pose_t>& out) const {
    out.clear();
    out.push_back(toPose(pose.inv() * background.getPose()));  // reference TSDF.cpp:112
    for (const auto& obj : objects) out.push_back(toPose(pose.inv() * obj.getPose()));
}

float EMFusion::stamp(int slot) {
    if (timingsOn) hipCheck(hipEventRecord(stamps[slot], main.get()), "hipEventRecord");
    return 0.f;
}

double EMFusion::pixels() const { return static_cast<double>(params.frameSize.area()); }

void EMFusion::synchronize() {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (comm) comm->check();  // an exchange that timed out on the device surfaces here (EMF_E_PEER_TIMEOUT)
}

// Wait for everything THIS instance has enqueued -- its three frame streams, the per-volume streams and the
// null stream its constructors clear on -- and for nothing else: unlike hipDeviceSynchronize() this does not
// stall on (or get stalled by) other work on the device.  What the frame itself uses where the object set
// changes (reference EMFusion.cpp:495-560, 827-863, 922-980 run inside processFrame).
void EMFusion::quiesce() {
    main.waitForCompletion();
    aux.waitForCompletion();
    lists.waitForCompletion();
    for (auto& kv : streams) kv.second.waitForCompletion();
    Stream::Null().waitForCompletion();
}

void EMFusion::enableRaycastStats(bool on) {
    statsOn = on;
    raycastStatsDev.setZero(main);
}

std::array<uint64_t, 4> EMFusion::raycastStats() {
    synchronize();
    std::array<uint64_t, 4> h{};
    raycastStatsDev.download(h.data(), main);
    return h;
}

uint64_t EMFusion::takeIntegratedVoxels() {
    synchronize();
    uint64_t v = 0;
    integrateStatsDev.download(&v, main);
    integrateStatsDev.setZero(main);
    main.waitForCompletion();
    return v;
}

// The batched path leaves the visibility counts in pinned memory (stored by the kernel itself) instead of
// stalling the frame; turn them into the host-side set when somebody asks.
void EMFusion::refreshVisibleFromDevice() {
    if (!visPending) return;
    // (round 4: no event behind the counts any more -- a record costs the critical stream ~8 us per frame whether
    // or not anybody asks; whoever asks waits for the stream instead)
    main.waitForCompletion();
    vis_objs.clear();
    for (size_t k = 0; k < visIds.size(); ++k)
        if (visibleHost[k] > params.visibilityThresh) vis_objs.insert(visIds[k]);
    visPending = false;
}

const std::set<int>& EMFusion::visibleObjects() {
    refreshVisibleFromDevice();
    return vis_objs;
}

void EMFusion::processFrame(const RGBD& frame) {
    if (frame.size.width != params.frameSize.width || frame.size.height != params.frameSize.height)
        throw HipError("EMFusion::processFrame: frame size differs from Params::frameSize",
                       EMF_E_SHAPE);
    const auto t0 = std::chrono::steady_clock::now();
    int slot = 0;
    const emf_image_t depthDev = stageDepth(frame.depth, slot);  // reference EMFusion.cpp:72
    uploadHostSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++uploads;
    FrameInputs in = pending;
    in.preprocessDepth = true;              // reference EMFusion.cpp:74
    if (!maskPath.empty() && frameCount % params.maskRCNNFrames == 0) loadPreprocMasks(in);  // EMFusion.cpp:99-101, 375-395
    // the slot's device image may be overwritten once this frame's kernels are through -- also those a frame that
    // throws half-way has already enqueued
    auto markDone = [&]() {
        if (!asyncUpload) return;
        UploadSlot& u = uploadSlots[slot];
        if (hipEventRecord(u.frameDone, main.get()) == hipSuccess) u.frameDoneValid = true;
        else (void)hipGetLastError();
    };
    try {
        runSchedule(depthDev, in);
    } catch (...) {
        markDone();
        throw;
    }
    markDone();
}

// Hand the host depth map to the device; returns the view the frame's kernels read.
emf_image_t EMFusion::stageDepth(const float* host, int& slot) {
    slot = static_cast<int>(uploads & 1u);
    UploadSlot& u = uploadSlots[slot];
    const size_t bytes = static_cast<size_t>(params.frameSize.area()) * sizeof(float);
    if (u.dev.empty()) {
        u.dev = DeviceImage<float>(params.frameSize);
        hipCheck(hipHostMalloc(reinterpret_cast<void**>(&u.pinned), bytes, hipHostMallocDefault), "hipHostMalloc(depth staging)");
        hipCheck(hipEventCreateWithFlags(&u.copied, hipEventDisableTiming), "hipEventCreate");
        hipCheck(hipEventCreateWithFlags(&u.frameDone, hipEventDisableTiming), "hipEventCreate");
    }
    const emf_image_t view = u.dev.view();
    if (!asyncUpload) {  // pageable source on the frame's stream: the runtime stages it, this call blocks
        hipCheck(hipMemcpyAsync(u.dev.ptr(), host, bytes, hipMemcpyHostToDevice, main.get()), "hipMemcpyAsync H2D");
        return view;
    }
    // the staging buffer is free once the copy of two frames ago has left it (long ago: no wait in practice)
    if (u.copiedValid) hipCheck(hipEventSynchronize(u.copied), "hipEventSynchronize(depth staging)");
    std::memcpy(u.pinned, host, bytes);
    // ... and the device image once the frame of two frames ago is through with it
    if (u.frameDoneValid) hipCheck(hipStreamWaitEvent(copyStream.get(), u.frameDone, 0), "hipStreamWaitEvent(frame done)");
    hipCheck(hipMemcpyAsync(u.dev.ptr(), u.pinned, bytes, hipMemcpyHostToDevice, copyStream.get()), "hipMemcpyAsync H2D (pinned)");
    hipCheck(hipEventRecord(u.copied, copyStream.get()), "hipEventRecord(depth copied)");
    u.copiedValid = true;
    hipCheck(hipStreamWaitEvent(main.get(), u.copied, 0), "hipStreamWaitEvent(depth copied)");
    return view;
}

void EMFusion::processFrame(const emf_image_t& depthDev, const FrameInputs& in) {
    runSchedule(depthDev, in);
}

void EMFusion::preprocessDepth(const emf_image_t& depthRaw, const emf_image_t& depthOut) {
    emfCheck(emf_hip_preprocessDepth(&depthRaw, &depthOut, params.bilateral_kernel_size,
                                     params.bilateral_sigma_depth, params.bilateral_sigma_spatial,
                                     main.abi()),
             "preprocessDepth");
}

void EMFusion::runSchedule(const emf_image_t& depthDev, const FrameInputs& in) {
    // A frame that threw between the fork of the background's integration and its join (object creation,
    // mask integration ... run in between) leaves the fork open: that integration WAS the frame's own, so
    // it is joined here -- copies flipped once, now -- instead of being mistaken for this frame's launch.
    if (bgInFlight) joinBackground();
    adoptReciprocals();
    depth = depthDev;
    stamp(kStart);
    if (sharded && depthRoot >= 0)  // 1.2 MB at VGA, once per frame
        comm->broadcast(depthDev.data, depthDev.pitch * static_cast<size_t>(depthDev.height), depthRoot, main);
    if (in.preprocessDepth) {
        preprocessDepth(depthDev, depthFiltered.view());
        depth = depthFiltered.view();
    }
    // computePoints (EMFusion.cpp:73): on the batched path the first E-step of the frame forms the
    // points from the depth on its way and stores them (one launch less); frame 0 has no E-step
    pointsPending = batched && frameCount > 0 && fusePoints;
    if (!pointsPending) {
        const emf_image_t pv = points.view();
        auto kt = ktimers.scope(KernelTimers::Points, pixels(), main);
        emfCheck(emf_hip_computePoints(&depth, &pv, params.intr.val, main.abi()),
                 "computePoints");
    }
    stamp(kPoints);

    auto applyObjectPoses = [&]() {
        for (auto& obj : objects) {
            auto it = in.obj_poses.find(obj.getID());
            if (it != in.obj_poses.end()) obj.setPose(it->second);
        }
    };

    if (frameCount > 0) {
        // Q17: the E-step runs three times per frame around the two tracking stages
        // (reference EMFusion.cpp:79, 687, 87).  The stages either run here (trackCamera /
        // trackObjects) or their results arrive as in.cam_pose / in.obj_poses.
        // The far bounds of this frame's raycast need the frame's FINAL poses and nothing else of it:
        // with both poses supplied they are known now and the bounds are computed beside the E-steps,
        // otherwise as soon as the object tracking is through (beside the last E-step).
        const bool posesKnown = !in.trackCamera && !in.trackObjects;
        if (posesKnown) {
            std::vector<emf_pose_t> co;
            co.push_back(toPose(background.getPose().inv() * in.cam_pose));
            for (const auto& obj : objects) {
                const auto it = in.obj_poses.find(obj.getID());
                co.push_back(toPose((it != in.obj_poses.end() ? it->second : obj.getPose()).inv() * in.cam_pose));
            }
            computeFarBounds(co);
        }
        computeAssociationWeights();
        if (saveOutput) storeAssocs(bg_assocWeight_preTrack, obj_assocWeights_preTrack);  // EMFusion.cpp:80-83
        if (in.trackCamera) trackCamera();  // EMFusion.cpp:673-685
        else pose = in.cam_pose;            // ... or its result, supplied
        if (saveOutput && in.trackCamera) storeTrackWeights(0, 1);
        computeAssociationWeights();
        if (in.trackObjects) trackObjects();  // EMFusion.cpp:689-723
        else applyObjectPoses();
        if (saveOutput && in.trackObjects) storeTrackWeights(1, static_cast<int>(objects.size()));
        if (!posesKnown) {
            std::vector<emf_pose_t> co;
            posesCO(co);
            computeFarBounds(co);
        }
        computeAssociationWeights();
        if (saveOutput) {
            storeAssocs(bg_assocWeight_postTrack, obj_assocWeights_postTrack);  // EMFusion.cpp:88-91
            storeFgProbs();  // what obj.getFgProbVals returns at the frame's end (EMFusion.cpp:120): this E-step's look-ups
        }
        stamp(kEstep);
        integrateBackgroundAsync();  // runs beside the raycast (see there)
        joinFarBounds();
        raycast();
    } else {
        pose = in.cam_pose;
        applyObjectPoses();
        stamp(kEstep);
        stamp(kRaycast);
        stamp(kComposite);
    }

    // object creation from unmatched masks happens between raycast and integration, so that a new
    // (empty, hence invisible to the raycast) volume still integrates its first frame (Q18)
    lastCreated.clear();
    std::map<int, emf_image_t> masks = in.masks;
    for (const emf_image_t& m : in.newObjectMasks) {
        const int id = initNewObjVolume(m);
        lastCreated.push_back(id);
        if (id >= 0) masks[id] = m;
    }
    lastAssigned.clear();
    const bool instances = !in.instanceMasks.empty();
    if (instances) {  // reference EMFusion.cpp:100-101
        std::vector<emf_image_t> segs = in.instanceMasks;
        masks = initOrMatchObjs(segs, lastAssigned, in.instanceScores);
        masks.erase(-1);
    }

    if (poseLog) {  // storePoses (EMFusion.cpp:322-327)
        poses[frameCount] = pose;
        for (const auto& obj : objects) obj_poses[obj.getID()][frameCount] = obj.getPose();
    }

    integrateBackgroundAsync();  // frame 0 (no raycast): same path, nothing to overlap with
    integrateDepth();
    stamp(kIntegrate);

    if ((in.runMasks || instances || !in.newObjectMasks.empty()) && !masks.empty())
        integrateMasks(masks);
    lastDeleted.clear();
    if (in.cleanUp) {
        if (in.runMasks && !instances)  // initOrMatchObjs' bookkeeping (EMFusion.cpp:358-369)
            for (auto& obj : objects) obj.updateExProb(masks.count(obj.getID()) != 0);
        lastDeleted = cleanUpObjs(in.runMasks || instances, masks);
    }
    stamp(kMasks);

    if (timingsOn) {
        hipCheck(hipEventSynchronize(stamps[kMasks]), "hipEventSynchronize");
        auto ms = [&](int a, int b) {
            float t = 0.f;
            hipCheck(hipEventElapsedTime(&t, stamps[a], stamps[b]), "hipEventElapsedTime");
            return t;
        };
        timings.points = ms(kStart, kPoints);
        timings.estep = ms(kPoints, kEstep);
        timings.raycast = ms(kEstep, kRaycast);
        timings.composite = ms(kRaycast, kComposite);
        timings.integrate = ms(kComposite, kIntegrate);
        timings.masks = ms(kIntegrate, kMasks);
        timings.total = ms(kStart, kMasks);
    }
    ++frameCount;
}

const ObjTSDF* EMFusion::getObject(int id) const {
    for (const auto& o : objects)
        if (o.getID() == id) return &o;
    return nullptr;
}

void EMFusion::computeAssociationWeights() {
    if (batched)
        estepBatched();
    else
        estepPerVolume();
}

void EMFusion::raycast() {
    if (batched)
        raycastBatched();
    else
        raycastPerVolume();
}

void EMFusion::integrateDepth() {
    if (batched)
        integrateBatched();
    else
        integratePerVolume();
}

// ---- batched path ----------------------------------------------------------------------------------

// One E-step launch over the table slots [first, first + count) (<= EMF_MAX_BATCH of them).
void EMFusion::launchEstep(const std::vector<emf_pose_t>& co, int first, int count, bool fromDepth, int normalize,
                           const emf_image_t* norm, const emf_image_t* objSum) {
    const emf_image_t pv = points.view();
    const emf_model_t* table = currentTable() + first;
    auto kt = ktimers.scope(KernelTimers::Assoc, pixels() * count, main);
    if (fromDepth)
        emfCheck(emf_hip_estepBatchedFromDepth(table, co.data() + first, count, &depth, params.intr.val, &pv, normalize,
                                               norm, objSum, main.abi()),
                 "estepBatchedFromDepth");
    else
        emfCheck(emf_hip_estepBatched(table, co.data() + first, count, &pv, normalize, norm, objSum, main.abi()),
                 "estepBatched");
}

void EMFusion::estepBatched() {
    std::vector<emf_pose_t> co;
    posesCO(co);
    const int n = static_cast<int>(co.size());
    const bool fromDepth = pointsPending;  // the frame's first E-step also makes the points
    pointsPending = false;
    if (sharded) {
        estepSharded(co, fromDepth);
        return;
    }
    const emf_image_t nv = associationNorm.view();
    if (n <= EMF_MAX_BATCH) {  // likelihoods, their sum and the normalised maps in ONE launch
        launchEstep(co, 0, n, fromDepth, 1, &nv, nullptr);
        return;
    }
    // More models than one launch takes: un-normalised likelihoods chunk by chunk (the first one forms the points on
    // its way), then ONE normalisation over all maps -- the same add chain, background first, objects in ascending id
    // (= table) order (EMFusion.cpp:654-657): the same bits as the fused launch and as the per-volume path.
    forChunks(0, n, [&](int first, int count) { launchEstep(co, first, count, fromDepth && first == 0, 0, nullptr, nullptr); });
    // (one launch over the table's maps: emf_hip_normalizeAssociation would take ceil(n / 16) launches to sum and as many to divide)
    auto kt = ktimers.scope(KernelTimers::Normalize, pixels() * n, main);
    emfCheck(emf_hip_normalizeAssociationTable(currentTable(), n, params.frameSize.width, params.frameSize.height,
                                               associationNorm.ptr(), main.abi()),
             "normalizeAssociationTable");
}

void EMFusion::raycastBatched() {
    std::vector<emf_pose_t> co;
    posesCO(co);
    const int n = static_cast<int>(co.size());
    uint64_t* stats = statsOn ? raycastStatsDev.as<uint64_t>() : nullptr;
    {
        auto kt = ktimers.scope(KernelTimers::Raycast, pixels() * n, main);
        const emf_model_t* table = currentTable();
        const int w = params.frameSize.width, h = params.frameSize.height;
        const int flags = TSDF::brickFlagMode() != 0;
        // One grid for all models.  (Measured alternative: the objects' grid on a second stream so
        // that their waves need not queue behind the resident background -- 3 % slower, the two
        // queues did not interleave usefully; scripts/raycast_timeline.py shows the queueing.)
        // Sharded: the background is replicated, so its raycast -- the largest kernel of the frame
        // -- is split into row bands, one per rank (Plan A); raylengths and hit mask of
        // the bands are then gathered (1.5 MB at VGA).  Background vertices / normals stay
        // band-local: like the remote objects' they only feed rendering.
        const int band = sharded && bgBands ? bgBandRows(h, world) : 0;
        const float* far = farBoundsReady && !flags ? farBoundsHalf() : nullptr;
        farBoundsReady = false;
        const size_t cells = emf_hip_raycastFarBoundBytes(1, w, h) / sizeof(float);  // bounds per model
        forChunks(0, n, [&](int first, int count) {
            const float* farChunk = far ? far + cells * first : nullptr;
            const float* vox = useFootprints ? voxelHost.data() + first : nullptr;
            if (first == 0)
                emfCheck(emf_hip_raycastBatchedLanes(table, co.data(), resHost.data(), count, w, h, params.intr.val,
                                                     flags, band ? std::min(rank * band, ((h + 15) / 16) * 16) : 0,
                                                     band, farChunk, vox, marchLanes, stats, main.abi()),
                         "raycastBatched");
            else  // a chunk of objects only (more than EMF_MAX_BATCH models)
                emfCheck(emf_hip_raycastBatchedObjects(table + first, co.data() + first, resHost.data() + 3 * first, count,
                                                       w, h, params.intr.val, flags, farChunk, vox, stats, main.abi()),
                         "raycastBatchedObjects");
        });
        bandRowsPending = band;  // gathered together with the nearest-hit keys: one exchange (compositeAcrossRanks)
    }
    stamp(kRaycast);
    compositeAndVisibility(true);
}

// Far bounds of this frame's raycast for its final camera -> volume poses `co`.  They read the relevant-tile
// lists, which are rebuilt on the `lists` stream behind the integrations -- so that is where the bounds are
// computed too, in order behind last frame's rebuilds and beside whatever `main` is doing (E-steps);
// joinFarBounds() makes `main` wait for them in front of the raycast.
void EMFusion::computeFarBounds(const std::vector<emf_pose_t>& co) {
    farBoundsReady = false;
    if (!batched || farBounds.empty() || TSDF::brickFlagMode() != 0) return;
    // Listed models only (no sign-map scan: nothing of this frame's object integration is read), the
    // background's list rebuilt on `lists` itself: the bounds need nothing of `main` but the previous raycast
    // to be through with the buffer -- they run beside the composite and the objects' integration instead of
    // beside the E-steps.
    // Round 4: the bounds alternate between two halves of the buffer, so the raycast of the PREVIOUS frame may
    // still be reading its half while these are written; the last reader of this half is the raycast of two frames
    // ago, and `main`'s event was re-recorded behind that one when the previous frame forked the background's
    // integration (integrateBackgroundAsync: behind its last E-step).  No event of its own behind every raycast
    // any more (a record costs the critical stream ~8 us per frame).
    farSel ^= 1;
    if (earlyFarBounds && !anyScan && forkFrame == frameCount - 1 && overlapUsable() && !bgBackStale)
        lists.waitOn(main);
    else
        lists.waitFor(main);  // the previous raycast has read the bounds (and in-place paths rebuilt lists on main)
    const size_t cells = emf_hip_raycastFarBoundBytes(1, params.frameSize.width, params.frameSize.height) / sizeof(float);
    forChunks(0, static_cast<int>(co.size()), [&](int first, int count) {
        emfCheck(emf_hip_raycastFarBounds(currentTable() + first, co.data() + first, resHost.data() + 3 * first, count,
                                          params.frameSize.width, params.frameSize.height, params.intr.val,
                                          chunkMask(scanSlot, first, count), farBoundsHalf() + cells * first, lists.abi()),
                 "raycastFarBounds");
    });
    farBoundsReady = true;
}

void EMFusion::joinFarBounds() {
    if (farBoundsReady) main.waitFor(lists);  // before this frame's list rebuild is enqueued there
    rebuildBackgroundList();
}

// The background's sign maps may have grown in the integration just forked: rebuild the list the NEXT
// frame's far bounds read -- behind that integration and behind whatever of this frame still reads the
// list, on a stream nobody waits for this frame.
void EMFusion::rebuildBackgroundList() {
    if (!bgListPending) return;
    bgListPending = false;
    if (!useFarBounds || farBounds.empty()) return;
    lists.waitOn(aux);  // the record() behind the integration kernels (this frame's far bounds ran on `lists`)
    emfCheck(emf_hip_updateRelevantTiles(currentTable(), resHost.data(), 1, lists.abi()), "updateRelevantTiles");
}

bool EMFusion::overlapUsable() const {
    return batched && bgOverlap && background.doubleBuffered() && !bgCullScratch.empty();
}

// Fork: the background's integration of this frame needs the pose, the depth map and the background
// association weights of the last E-step -- all known BEFORE the raycast -- and nothing the raycast
// produces (only object volumes are gated by its visibility counts, EMFusion.cpp:869-872).  With the
// background kept twice it runs out of place on `aux` while `main` ray-marches the front copy: the
// raycast is a latency chain of its longest rays that leaves most of the chip idle, the integration
// is a streaming sweep that fills it.  Same values as the reference's raycast -> integrate sequence.
// It is forked as soon as the last E-step is enqueued, before the far bounds: its box cull then runs
// beside them instead of fighting the raycast's workgroup dispatch (8 us instead of 40).
void EMFusion::integrateBackgroundAsync() {
    if (!overlapUsable() || bgInFlight) return;
    if (bgBackStale) {  // an in-place integration (other path) in between: re-equalise the copies
        quiesce();
        background.resyncBack();
        bgBackStale = false;
        bgPrepared = false;
    }
    aux.waitFor(main);
    const emf_pose_t oc = toPose(pose.inv() * background.getPose());  // reference TSDF.cpp:112
    const double vox = static_cast<double>(resHost[0]) * resHost[1] * resHost[2];
    const emf_image_t il = invLambda.view();
    const emf_volume_out_t out = background.backBuffers();
    {
        auto kt = ktimers.scope(KernelTimers::IntegrateBg, vox, aux);
        emfCheck(emf_hip_integrateBatchedCulledOut(currentTable(), &oc, resHost.data(), 1, nullptr, &depth,
                                                   useLambdaTable ? &il : nullptr, params.intr.val, &out,
                                                   bgPrepared ? 1 : 0, bgCullScratch.data(), 0, nullptr,
                                                   integrateStatsDev.as<uint64_t>(), aux.abi()),
                 "integrateBatchedCulledOut");
    }
    aux.record();  // what joinBackground() and the list rebuild wait for
    // clear, behind this call and off everybody's path, what the NEXT call wants clean: the box counter
    // and the map that will be its dirtyNext (this call's dirtyPrev: the copies swap roles)
    emf_volume_out_t next = out;
    next.dirtyNext = const_cast<uint8_t*>(out.dirtyPrev);
    emfCheck(emf_hip_integratePrepareOut(&next, resHost.data(), 1, bgCullScratch.data(), aux.abi()),
             "integratePrepareOut");
    bgPrepared = true;
    bgInFlight = true;
    bgListPending = true;
    forkFrame = frameCount;
}

// Join: the frame's later stages (and the next frame) see the integrated background.
void EMFusion::joinBackground() {
    if (!bgInFlight) return;
    rebuildBackgroundList();  // (a frame without far bounds: frame 0)
    main.waitOn(aux);  // the record() behind the integration kernels
    background.flip();
    tableSel ^= 1;
    bgInFlight = false;
}

void EMFusion::integrateBatched() {
    std::vector<emf_pose_t> oc;
    posesOC(oc);
    const int n = static_cast<int>(oc.size());
    const int first = bgInFlight ? 1 : 0;  // the background is already on its way
    if (!bgInFlight && background.doubleBuffered()) bgBackStale = true;  // in place below
    if (n > first) {
        double vox = 0;
        for (int m = first; m < n; ++m)
            vox += static_cast<double>(resHost[3 * m]) * resHost[3 * m + 1] * resHost[3 * m + 2];
        auto kt = ktimers.scope(KernelTimers::Integrate, vox, main);
        const emf_image_t il = invLambda.view();
        const emf_image_t* ilp = useLambdaTable ? &il : nullptr;
        forChunks(first, n, [&](int from, int count) {
            const emf_model_t* table = currentTable() + from;
            const int32_t* vis = visibleDev.as<int32_t>() + from;
            // two-level launch: the boxes of tiles outside the view cone never get a workgroup -- what the
            // background needs; object volumes alone are small and mostly in view, and the list's counter
            // reset + cull kernel cost them more (24 us of the frame) than the culled tiles would
            if ((from == 0 || (objCull && from < EMF_MAX_BATCH)) && cullBoxes && !integrateCullScratch.empty()) {
                emfCheck(emf_hip_integrateBatchedCulled(table, oc.data() + from, resHost.data() + 3 * from, count,
                                                        vis, &depth, ilp, params.intr.val,
                                                        integrateCullScratch.data(), 0, nullptr,
                                                        integrateStatsDev.as<uint64_t>(), main.abi()),
                         "integrateBatchedCulled");
            } else {
                emfCheck(emf_hip_integrateBatched(table, oc.data() + from, resHost.data() + 3 * from, count, vis,
                                                  &depth, ilp, params.intr.val, TSDF::brickFlagMode() != 0,
                                                  integrateStatsDev.as<uint64_t>(), main.abi()),
                         "integrateBatched");
            }
        });
    }
    const bool overlapped = bgInFlight;
    joinBackground();
    if (useFarBounds && !farBounds.empty()) {
        // The sign maps may have grown: rebuild the relevant-tile lists the NEXT frame's far bounds read.
        // Nothing of this frame needs them: with the streams in use they go to `lists`, behind the
        // integration above (the background's own list went there behind its integration already).
        bool waited = false;
        forChunks(overlapped ? 1 : 0, n, [&](int from, int count) {
            if (chunkMask(listSlot, from, count) == 0) return;  // no model of this chunk keeps a list
            if (!waited) lists.waitFor(main);
            waited = true;
            emfCheck(emf_hip_updateRelevantTiles(currentTable() + from, resHost.data() + 3 * from, count, lists.abi()),
                     "updateRelevantTiles");
        });
    }
}

// Compositing in list (creation) order + visibility counts (reference EMFusion.cpp:760-794).
// deviceGate: turn the counts into the integrate gate on the device and mirror them to pinned
// memory; otherwise wait for them here (the reference's behaviour).
void EMFusion::compositeAndVisibility(bool deviceGate) {
    if (sharded) {
        compositeAcrossRanks(deviceGate);
        return;
    }
    std::vector<int32_t> ids;
    std::vector<emf_image_t> oray, overt, onorm, oseg;
    for (auto& obj : objects) {
        ObjImages& im = objImages.at(obj.getID());
        ids.push_back(obj.getID());
        oray.push_back(im.raylengths.view());
        overt.push_back(im.vertices.view());
        onorm.push_back(im.normals.view());
        oseg.push_back(im.modelSegmentation.view());
    }
    const emf_image_t v_bgRay = bg_raylengths.view(), v_bgVert = bg_vertices.view(),
                      v_bgNorm = bg_normals.view(), v_bgMask = bg_mask.view(),
                      v_ray = raylengths.view(), v_vert = vertices.view(),
                      v_norm = normals.view(), v_seg = modelSegmentation.view(),
                      v_diff = diffRaylengths.view(), v_noObj = noObjMask.view();
    const int nobj = static_cast<int>(ids.size());
    {
        auto kt = ktimers.scope(KernelTimers::Composite, pixels() * (1.0 + nobj), main);
        if (deviceGate && fuseVisibility) {
            // the composite's own launch counts; the counts also go to pinned host memory straight from
            // the flag kernel, which leaves visCounts cleared for the next frame
            if (!visCountsClear) visCounts.setZero(main);  // (another path left its numbers there)
            visCountsClear = true;
            emfCheck(emf_hip_compositeVisibility(nobj, ids.data(), oray.data(), overt.data(), onorm.data(),
                                                 oseg.data(), &v_bgRay, &v_bgVert, &v_bgNorm, &v_bgMask, &v_ray,
                                                 &v_vert, &v_norm, &v_seg, &v_diff, &v_noObj, params.boundary,
                                                 visCounts.as<int32_t>(), params.visibilityThresh,
                                                 visibleDev.as<int32_t>(), visibleHost, main.abi()),
                     "compositeVisibility");
        } else {
            visCountsClear = false;
            emfCheck(emf_hip_compositeRaycast(nobj, ids.data(), oray.data(), overt.data(),
                                              onorm.data(), oseg.data(), &v_bgRay, &v_bgVert,
                                              &v_bgNorm, &v_bgMask, &v_ray, &v_vert, &v_norm, &v_seg,
                                              &v_diff, &v_noObj, params.boundary,
                                              visCounts.as<int32_t>(), main.abi()),
                     "compositeRaycast");
            if (deviceGate)  // the counts also go to pinned host memory straight from the kernel
                emfCheck(emf_hip_visibilityFlags(visCounts.as<int32_t>(), nobj + 1,
                                                 params.visibilityThresh, visibleDev.as<int32_t>(),
                                                 visibleHost, main.abi()),
                         "visibilityFlags");
        }
    }
    stamp(kComposite);
    vis_objs.clear();
    visPending = false;
    if (nobj == 0) return;
    if (deviceGate) {
        visIds = ids;
        visPending = true;
        return;
    }
    hipCheck(hipMemcpyAsync(visCountsHost, visCounts.data(), sizeof(int32_t) * nobj,
                            hipMemcpyDeviceToHost, main.get()),
             "visCounts D2H");
    main.waitForCompletion();  // the visible set gates integrateDepth (EMFusion.cpp:869-872)
    for (int k = 0; k < nobj; ++k)
        if (visCountsHost[k] > params.visibilityThresh) vis_objs.insert(ids[k]);
}

}  // namespace emf
