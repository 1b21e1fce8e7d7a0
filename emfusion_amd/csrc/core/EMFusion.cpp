// EMFusion.cpp -- per-frame schedule (see EMFusion.hpp).  Reference: src/core/EMFusion.cpp.
//
// Two execution paths produce the same results:
//   batched  (default)  one launch per stage for all models of this rank, driven by a
//            device-resident model table; the visibility gate of integrateDepth is evaluated on
//            the device, so a frame contains no host synchronisation at all
//   per-volume (fallback: more than EMF_MAX_BATCH models, volumes whose Nx is not a multiple of
//            4, materialised gradient volumes, or EMF_PER_VOLUME=1)  the reference's structure:
//            one HIP stream per volume joined by events, host-side visibility gate
#include "EMFusion.hpp"
#include "Readers.hpp"

#include <sys/stat.h>

#include <cerrno>

#include "Output.hpp"

#include <algorithm>
#include <atomic>
#include <exception>
#include <fstream>
#include <chrono>
#include <cmath>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace emf {

namespace {
enum Stamp { kStart = 0, kPoints, kEstep, kRaycast, kComposite, kIntegrate, kMasks, kNumStamps };

emf_pose_t toPose(const Affine3f& a) {
    emf_pose_t p;
    for (int i = 0; i < 9; ++i) p.R[i] = a.rotation().val[i];
    for (int i = 0; i < 3; ++i) p.t[i] = a.translation().val[i];
    return p;
}
}  // namespace

EMFusion::EMFusion(const Params& _params, TSDF::Gradients gradients,
                   std::shared_ptr<Communicator> _comm)
    : params(_params),
      gradMode(gradients),
      comm(std::move(_comm)),
      background(_params.globalVolumeDims, _params.globalVoxelSize,
                 _params.globalRelTruncDist * _params.globalVoxelSize, _params.volumePose,
                 _params.tsdfParams, _params.frameSize, gradients),
      depthUpload(_params.frameSize),
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
      modelTable(2 * sizeof(emf_model_t) * EMF_MAX_BATCH),
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
    scanMask = 0;
    listMask = 0;
    for (const auto& md : modelsHost) {
        const unsigned bit = 1u << (voxelHost.size() & 31);
        // A volume too small for a relevant-tile list (an object: < 8192 tiles) could have its sign maps scanned
        // for far bounds; its rays are short anyway and the scan (23 us beside the E-steps, which it slows from
        // 12 to 37 us) costs the frame more than the cut saves the raycast: 0.6095 vs 0.5956 ms.  EMF_FAR_SCAN=1
        // scans them.
        const char* fsc = debugEnv("EMF_FAR_SCAN");
        const bool scanSmall = fsc && fsc[0] == '1';
        if (md.signMaps && !md.relevantTiles && scanSmall) scanMask |= bit;
        if (md.signMaps && md.relevantTiles) listMask |= bit;
        voxelHost.push_back(md.voxelSize);
        resHost.insert(resHost.end(), md.res, md.res + 3);
        tiled &= md.res[0] % 4 == 0;
    }
    // scratch of the two-level integration launch (every model on float4 tiles, no brick flags to keep)
    integrateCullScratch = DeviceBuffer();
    if (cullBoxes && tiled && TSDF::brickFlagMode() == 0 &&
        static_cast<int>(modelsHost.size()) <= EMF_MAX_BATCH)
        integrateCullScratch = DeviceBuffer(emf_hip_integrateCullScratchBytes(
            resHost.data(), static_cast<int>(modelsHost.size())));
    batched = !forceLegacy && gradMode == TSDF::Gradients::OnTheFly &&
              static_cast<int>(modelsHost.size()) <= EMF_MAX_BATCH;
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
            hipCheck(hipMemcpy(modelTable.as<emf_model_t>() + EMF_MAX_BATCH, alt.data(),
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
            emfCheck(emf_hip_updateRelevantTiles(modelTable.as<emf_model_t>(), resHost.data(),
                                                 static_cast<int>(modelsHost.size()), Stream::Null().abi()),
                     "updateRelevantTiles");
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
            emf_model_t* row = modelTable.as<emf_model_t>() + t * EMF_MAX_BATCH + slot;
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
    depthUpload.upload(frame.depth, main);  // reference EMFusion.cpp:72
    FrameInputs in = pending;
    in.preprocessDepth = true;              // reference EMFusion.cpp:74
    if (!maskPath.empty() && frameCount % params.maskRCNNFrames == 0) loadPreprocMasks(in);  // EMFusion.cpp:99-101, 375-395
    runSchedule(depthUpload.view(), in);
}

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

// ---- object creation / matching from masks ---------------------------------------------------------

void EMFusion::ensureLifecycleBuffers() {
    if (!statsScratch.empty()) return;
    statsScratch = DeviceBuffer(emf_hip_pointStatsScratchBytes());
    statsDev = DeviceBuffer(sizeof(emf_point_stats_t));
    overlapDev = DeviceBuffer(513 * sizeof(uint32_t));
    massDev = DeviceBuffer(sizeof(emf_mask_mass_t));
    hipCheck(hipHostMalloc(&lifecycleHost, 513 * sizeof(uint32_t), hipHostMallocDefault),
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

void EMFusion::writeResults(const std::string& dir, bool volumes) {
    synchronize();
    // boost::filesystem::create_directories(p) (EMFusion.cpp:254-255): every missing component of the path
    for (size_t k = 1; k <= dir.size(); ++k)
        if (k == dir.size() || dir[k] == '/') {
            const std::string part = dir.substr(0, k);
            if (!part.empty() && mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
                throw std::runtime_error("EMFusion::writeResults: cannot create " + part);
        }
    io::writePoseFile(dir + "/poses-cam.txt", poses);
    for (const auto& op : obj_poses)
        io::writePoseFile(dir + "/poses-" + std::to_string(op.first) + ".txt", op.second);
    for (const auto& op : addPoseOffsets(obj_poses, obj_pose_offsets))  // EMFusion.cpp:1000-1006
        io::writePoseFile(dir + "/poses-" + std::to_string(op.first) + "-corrected.txt", op.second);
    // writeMeshes (EMFusion.cpp:1147-1156) runs whether or not volumes are exported: the background,
    // the live objects, and the objects that were deleted while the log was on (their last mesh,
    // EMFusion.cpp:966)
    io::writeMesh(dir + "/mesh_bg.ply", background.getMesh());
    for (auto& obj : objects)
        if (!(ignorePerson && isPerson(obj))) meshes[obj.getID()] = obj.getMesh();
    for (const auto& m : meshes) io::writeMesh(dir + "/mesh_" + std::to_string(m.first) + ".ply", m.second);
    // writeRenderings / writeAssocs / writeHuberWeights / writeTrackWeights / writeFgProbs (EMFusion.cpp:1009-1145):
    // directories are created whether or not the log holds anything, like the reference's
    io::writeImageLog(dir + "/output", renderings);
    io::writeImageLog(dir + "/assoc_weights/bg/preTrack", bg_assocWeight_preTrack);
    io::writeImageLog(dir + "/assoc_weights/bg/postTrack", bg_assocWeight_postTrack);
    for (const auto& o : obj_assocWeights_preTrack)
        io::writeImageLog(dir + "/assoc_weights/" + std::to_string(o.first) + "/preTrack", o.second);
    for (const auto& o : obj_assocWeights_postTrack)
        io::writeImageLog(dir + "/assoc_weights/" + std::to_string(o.first) + "/postTrack", o.second);
    io::writeImageLog(dir + "/huber_weights/bg", bg_huberWeights);
    for (const auto& o : obj_huberWeights) io::writeImageLog(dir + "/huber_weights/" + std::to_string(o.first), o.second);
    io::writeImageLog(dir + "/track_weights/bg", bg_trackWeights);
    for (const auto& o : obj_trackWeights) io::writeImageLog(dir + "/track_weights/" + std::to_string(o.first), o.second);
    io::createDirectories(dir + "/fg_probs");
    for (const auto& o : obj_fgProbs) io::writeImageLog(dir + "/fg_probs/" + std::to_string(o.first), o.second);
    if (!(volumes || expVols)) return;  // `if ( expVols ) writeTSDFs ( p )` (EMFusion.cpp:290-291)
    const std::string t = dir + "/tsdfs";
    if (mkdir(t.c_str(), 0777) != 0 && errno != EEXIST)
        throw std::runtime_error("EMFusion::writeResults: cannot create " + t);
    auto dump = [&](const std::string& name, const std::vector<float>& v, const Vec3i& res, float vox) {
        io::writeVolume(t + "/" + name + ".bin", v.data(), sizeof(float), res, vox);
    };
    dump("bg_tsdf", background.getTSDF(), background.getVolumeRes(), background.getVoxelSize());
    for (auto& obj : objects) {
        if (ignorePerson && isPerson(obj)) continue;  // the same `continue` skips them (EMFusion.cpp:274-277)
        savedVolumes[obj.getID()] = saveVolumes(obj);
    }
    for (const auto& sv : savedVolumes) {  // writeTSDFs (EMFusion.cpp:1195-1216): live and deleted objects
        const std::string id = std::to_string(sv.first);
        dump("tsdf_" + id, sv.second.tsdf, sv.second.res, sv.second.voxelSize);
        dump("weights_" + id, sv.second.weights, sv.second.res, sv.second.voxelSize);
        dump("fgProbs_" + id, sv.second.fgProbs, sv.second.res, sv.second.voxelSize);
    }
}

EMFusion::SavedVolumes EMFusion::saveVolumes(ObjTSDF& obj) {  // EMFusion.cpp:279-285, 967-973
    SavedVolumes sv;
    sv.tsdf = obj.getTSDF();
    sv.weights = obj.getWeightsVol();
    sv.fgProbs = obj.getFgProbVol();
    sv.res = obj.getVolumeRes();
    sv.voxelSize = obj.getVoxelSize();
    return sv;
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

// Reference EMFusion::addPoseOffsets (EMFusion.cpp:1220-1236): undo the accumulated centre shifts so
// that the trajectory refers to the volume centre the object was created with.
std::map<int, std::map<int, Affine3f>> EMFusion::addPoseOffsets(
    const std::map<int, std::map<int, Affine3f>>& all,
    const std::map<int, std::map<int, Vec3f>>& offsets) {
    std::map<int, std::map<int, Affine3f>> cleaned;
    for (const auto& op : all) {
        Vec3f cum = Vec3f::all(0.f);
        const auto off = offsets.find(op.first);
        for (const auto& fp : op.second) {
            if (off != offsets.end()) {
                const auto o = off->second.find(fp.first);
                if (o != off->second.end()) cum = cum - o->second;
            }
            cleaned[op.first][fp.first] = fp.second.translate(fp.second.rotation() * cum);
        }
    }
    return cleaned;
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
    refreshVisibleFromDevice();  // the host copy of vis_objs decides (one synchronisation)
    std::set<int> spurious;
    if (maskFrame)
        for (const auto& obj : objects)
            if (obj.getExProb() < params.existenceThresh) spurious.insert(obj.getID());
    ensureLifecycleBuffers();
    for (const auto& obj : objects) {
        const int id = obj.getID();
        if (!vis_objs.count(id)) continue;
        const ObjImages& im = objImages.at(id);
        const emf_image_t seg = im.modelSegmentation.view(), assoc = im.associationWeights.view();
        auto it = matches.find(id);
        emfCheck(emf_hip_maskAssociationMass(&seg, it == matches.end() ? nullptr : &it->second, &assoc,
                                             massDev.as<emf_mask_mass_t>(), main.abi()),
                 "maskAssociationMass");
        hipCheck(hipMemcpyAsync(lifecycleHost, massDev.data(), sizeof(emf_mask_mass_t),
                                hipMemcpyDeviceToHost, main.get()),
                 "hipMemcpyAsync");
        main.waitForCompletion();
        const emf_mask_mass_t mm = *static_cast<emf_mask_mass_t*>(lifecycleHost);
        if (params.assocThresh * static_cast<float>(mm.count) > mm.sum) spurious.insert(id);
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
        throw HipError("EMFusion: tracking needs the batched path (<= 32 models, on-the-fly gradients)",
                       EMF_E_LIMIT);
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t per = emf_hip_trackScratchBytes(w, h);
    if (trackStates.empty()) {
        trackStates = DeviceBuffer(sizeof(emf_track_state_t) * EMF_MAX_BATCH);
        trackScratch = DeviceBuffer(per * EMF_MAX_BATCH);
        hipCheck(hipHostMalloc(reinterpret_cast<void**>(&trackStatesHost),
                               sizeof(emf_track_state_t) * EMF_MAX_BATCH, hipHostMallocDefault),
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

void EMFusion::render(uint8_t* rgb) {
    if (sharded)
        throw HipError("EMFusion::render: not available on the sharded path (vertices / normals of "
                       "remote objects and background bands stay on their ranks)", EMF_E_ARG);
    const size_t bytes = static_cast<size_t>(params.frameSize.area()) * 3;
    if (frameCount < 1) {
        std::fill(rgb, rgb + bytes, uint8_t{0});
        return;
    }
    if (frameCount == 1) raycast();  // frame 0 ran without one (EMFusion.cpp:135-137)
    if (image.empty()) image = DeviceImage<uint8_t, 3>(params.frameSize);
    const emf_image_t vv = vertices.view(), nv = normals.view(), sv = modelSegmentation.view(),
                      iv = image.view();
    if (ignorePerson) {  // EMFusion.cpp:139-150: in place, like the reference
        const emf_image_t bv = bg_vertices.view(), bn = bg_normals.view();
        for (const auto& obj : objects)
            if (isPerson(obj))
                emfCheck(emf_hip_hideLabel(&sv, obj.getID(), &vv, &nv, &bv, &bn, main.abi()), "hideLabel");
    }
    const float light[3] = {0.f, 0.f, 0.f};  // cv::Affine3f::Identity()
    emfCheck(emf_hip_renderPhong(&vv, &nv, &sv, colorMap.data(), light, &iv, main.abi()), "renderPhong");
    hipCheck(hipMemcpyAsync(rgb, image.ptr(), bytes, hipMemcpyDeviceToHost, main.get()), "render D2H");
    main.waitForCompletion();
    if (saveOutput)  // `rendered.copyTo ( renderings[frameCount-1] )`, EMFusion.cpp:158-160
        renderings[frameCount - 1] = io::encodePng(rgb, params.frameSize.width, params.frameSize.height, 3);
}

// ---- per-frame debug images (reference saveOutput mode) ---------------------------------------------------

std::vector<uint8_t> EMFusion::pngOf(const float* dev, size_t pitchBytes) {
    const int w = params.frameSize.width, h = params.frameSize.height;
    std::vector<float> host(static_cast<size_t>(w) * h);
    hipCheck(hipMemcpy2DAsync(host.data(), static_cast<size_t>(w) * sizeof(float), dev, pitchBytes,
                              static_cast<size_t>(w) * sizeof(float), static_cast<size_t>(h), hipMemcpyDeviceToHost,
                              main.get()),
             "hipMemcpy2DAsync(debug image)");
    main.waitForCompletion();
    const std::vector<uint8_t> u8 = io::toU8Times255(host.data(), w, h, static_cast<size_t>(w));
    return io::encodePng(u8.data(), w, h, 1);
}

void EMFusion::storeAssocs(ImageLog& bg, std::map<int, ImageLog>& objs) {
    if (sharded) return;  // (remote objects' maps are not on this rank; the reference is single-GPU)
    const emf_image_t b = bg_associationWeights.view();
    bg[frameCount] = pngOf(static_cast<const float*>(b.data), b.pitch);
    for (const auto& obj : objects) {
        const emf_image_t a = objImages.at(obj.getID()).associationWeights.view();
        objs[obj.getID()][frameCount] = pngOf(static_cast<const float*>(a.data), a.pitch);
    }
}

void EMFusion::storeTrackWeights(int first, int count) {
    if (count <= 0 || trackStates.empty()) return;
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t px = static_cast<size_t>(w) * h, per = emf_hip_trackScratchBytes(w, h);
    if (logScratch.bytes() < 2 * px * sizeof(float) * count) logScratch = DeviceBuffer(2 * px * sizeof(float) * count);
    emf_track_params_t tp;
    tp.huberThresh = params.tsdfParams.huberThresh;
    tp.maxWeight = params.tsdfParams.maxTSDFWeight;
    tp.tau = params.tsdfParams.tau;
    tp.eps1 = params.tsdfParams.eps1;
    tp.eps2 = params.tsdfParams.eps2;
    tp.nuInit = params.tsdfParams.nu_init;
    const emf_image_t pv = points.view();
    float* huber = logScratch.as<float>();
    float* track = huber + px * count;
    // the stage's states are final and the models' association maps are still the ones it tracked with
    emfCheck(emf_hip_trackWeightImages(currentTable() + first, trackStates.as<emf_track_state_t>() + first, count, &pv, &tp,
                                       static_cast<const char*>(trackScratch.data()) + per * first, per, huber, track,
                                       main.abi()),
             "trackWeightImages");
    auto it = objects.begin();
    for (int m = 0; m < count; ++m) {
        const std::vector<uint8_t> hp = pngOf(huber + px * m, static_cast<size_t>(w) * sizeof(float));
        const std::vector<uint8_t> tpng = pngOf(track + px * m, static_cast<size_t>(w) * sizeof(float));
        if (first + m == 0) {
            bg_huberWeights[frameCount] = hp;
            bg_trackWeights[frameCount] = tpng;
        } else {
            const int id = (it++)->getID();
            obj_huberWeights[id][frameCount] = hp;
            obj_trackWeights[id][frameCount] = tpng;
        }
    }
}

void EMFusion::storeFgProbs() {
    if (sharded || objects.empty()) return;
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t px = static_cast<size_t>(w) * h;
    if (logScratch.bytes() < px * sizeof(float)) logScratch = DeviceBuffer(px * sizeof(float));
    const emf_image_t pv = points.view();
    const emf_image_t out{logScratch.data(), static_cast<size_t>(w) * sizeof(float), w, h};
    for (auto& obj : objects) {
        // cuda::TSDF::getVolumeVals ( fgProbs, points, rel_pose_CO ... fgProbVals ), ObjTSDF.cpp:189-191
        const Affine3f co = obj.getPose().inv() * pose;
        const Vec3i res = obj.getVolumeRes();
        const int32_t r[3] = {res[0], res[1], res[2]};
        emfCheck(emf_hip_getVolumeVals(obj.fgProbsPtr(), 1, &pv, co.rotation().val, co.translation().val, r,
                                       obj.getVoxelSize(), &out, main.abi()),
                 "getVolumeVals(fgProbs)");
        obj_fgProbs[obj.getID()][frameCount] = pngOf(logScratch.as<float>(), out.pitch);
    }
}

Mesh EMFusion::getMesh(int id) {
    synchronize();
    if (id == 0) return background.getMesh();
    for (auto& o : objects)
        if (o.getID() == id) return o.getMesh();
    throw HipError("EMFusion::getMesh: no object " + std::to_string(id) + " on this rank", EMF_E_ARG);
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

void EMFusion::estepBatched() {
    std::vector<emf_pose_t> co;
    posesCO(co);
    const int n = static_cast<int>(co.size());
    const emf_image_t pv = points.view(), nv = associationNorm.view(), sv = objPartialSum.view();
    const emf_model_t* table = currentTable();
    const bool fromDepth = pointsPending;  // the frame's first E-step also makes the points
    pointsPending = false;
    auto launch = [&](int normalize, const emf_image_t* norm, const emf_image_t* objSum) {
        auto kt = ktimers.scope(KernelTimers::Assoc, pixels() * n, main);
        if (fromDepth)
            emfCheck(emf_hip_estepBatchedFromDepth(table, co.data(), n, &depth, params.intr.val, &pv, normalize,
                                                   norm, objSum, main.abi()),
                     "estepBatchedFromDepth");
        else
            emfCheck(emf_hip_estepBatched(table, co.data(), n, &pv, normalize, norm, objSum, main.abi()),
                     "estepBatched");
    };
    if (!sharded) {
        launch(1, &nv, nullptr);
        return;
    }
    std::vector<emf_image_t> maps;
    maps.push_back(bg_associationWeights.view());
    for (auto& kv : objImages) maps.push_back(kv.second.associationWeights.view());
    if (peerFused && maps.size() <= 16) {
        // direct peer writes: the E-step's kernel stores its partial sum straight into the peers' slots, and ONE
        // more launch waits for the peers, sums the slots in rank order and normalises -- two launches per E-step
        // where the unsharded frame has one (round 3: five)
        const uint32_t seq = comm->beginPeerExchange(main);
        {
            auto kt = ktimers.scope(KernelTimers::Assoc, pixels() * n, main);
            emfCheck(emf_hip_estepBatchedPeer(table, co.data(), n, fromDepth ? &depth : nullptr, params.intr.val, &pv,
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
    launch(0, nullptr, &sv);
    // (Measured and dropped, round 3: the frame's LAST all-reduce + normalisation on a stream of their own beside
    // the raycast -- they feed the integrations only.  With a 30 us latency model the frame got no shorter: the
    // background's sweep needs the normalised weights and is as long as the raycast it runs beside.)
    Stream& st = main;
    comm->allReduceSumF32(objPartialSum.ptr(), params.frameSize.area(), st);
    {
        auto kt = ktimers.scope(KernelTimers::Normalize, pixels() * maps.size(), st);
        emfCheck(emf_hip_normalizeAssociation(maps.data(), static_cast<int>(maps.size()), 1, &sv, &nv,
                                              st.abi()),
                 "normalizeAssociation");
    }
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
        emfCheck(emf_hip_raycastBatched(table, co.data(), resHost.data(), n, w, h, params.intr.val,
                                        flags, band ? std::min(rank * band, ((h + 15) / 16) * 16) : 0,
                                        band, far, useFootprints ? voxelHost.data() : nullptr, stats, main.abi()),
                 "raycastBatched");
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
    if (earlyFarBounds && scanMask == 0 && forkFrame == frameCount - 1 && overlapUsable() && !bgBackStale)
        lists.waitOn(main);
    else
        lists.waitFor(main);  // the previous raycast has read the bounds (and in-place paths rebuilt lists on main)
    emfCheck(emf_hip_raycastFarBounds(currentTable(), co.data(), resHost.data(), static_cast<int>(co.size()),
                                      params.frameSize.width, params.frameSize.height, params.intr.val, scanMask,
                                      farBoundsHalf(), lists.abi()),
             "raycastFarBounds");
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
        const emf_model_t* table = currentTable() + first;
        const int32_t* vis = visibleDev.as<int32_t>() + first;
        // two-level launch: the boxes of tiles outside the view cone never get a workgroup -- what the
        // background needs; object volumes alone are small and mostly in view, and the list's counter
        // reset + cull kernel cost them more (24 us of the frame) than the culled tiles would
        if ((first == 0 || objCull) && cullBoxes && !integrateCullScratch.empty()) {
            emfCheck(emf_hip_integrateBatchedCulled(table, oc.data() + first, resHost.data() + 3 * first, n - first,
                                                    vis, &depth, ilp, params.intr.val,
                                                    integrateCullScratch.data(), 0, nullptr,
                                                    integrateStatsDev.as<uint64_t>(), main.abi()),
                     "integrateBatchedCulled");
        } else {
            emfCheck(emf_hip_integrateBatched(table, oc.data() + first, resHost.data() + 3 * first, n - first, vis,
                                              &depth, ilp, params.intr.val, TSDF::brickFlagMode() != 0,
                                              integrateStatsDev.as<uint64_t>(), main.abi()),
                     "integrateBatched");
        }
    }
    const bool overlapped = bgInFlight;
    joinBackground();
    if (useFarBounds && !farBounds.empty()) {
        // The sign maps may have grown: rebuild the relevant-tile lists the NEXT frame's far bounds read.
        // Nothing of this frame needs them: with the streams in use they go to `lists`, behind the
        // integration above (the background's own list went there behind its integration already).
        const int from = overlapped ? 1 : 0;
        if (n > from && (listMask >> from) != 0) {
            lists.waitFor(main);
            emfCheck(emf_hip_updateRelevantTiles(currentTable() + from, resHost.data() + 3 * from, n - from, lists.abi()),
                     "updateRelevantTiles");
        }
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
