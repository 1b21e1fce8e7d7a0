// EMFusion.hpp -- emf::EMFusion: per-frame schedule over one background volume + N object volumes.
//
// Keeps the orchestration surface of the reference's emf::EMFusion for the volumetric hot path
// (reference include/EMFusion/core/EMFusion.h:47-510, src/core/EMFusion.cpp:28-129, 635-670,
// 726-795, 865-906): processFrame() runs computePoints -> E-step -> [pose update] -> E-step ->
// raycast -> integrateDepth -> integrateMasks in the reference's order, with one HIP stream per
// volume (reference EMFusion.h:471) joined by events instead of host synchronisation.
//
// Outside this build's scope, and therefore supplied by the caller: camera / object poses
// (tracking), object creation and masks (Mask R-CNN + lifecycle, f-3), depth
// pre-filtering (f-2).  processFrame(RGBD) is kept for the reference call site
// (apps/EM-Fusion.cpp:152); it consumes poses / masks registered with setFrameInputs().
#pragma once

#include <algorithm>
#include <list>
#include <map>
#include <string>
#include <memory>
#include <set>
#include <vector>

#include "Communicator.hpp"
#include "KernelTimers.hpp"
#include "ObjTSDF.hpp"
#include "Output.hpp"
#include "TSDF.hpp"

namespace emf {

/** Host-side RGB-D frame (stand-in for emf::RGBD, reference include/EMFusion/utils/data.h). */
struct RGBD {
    Size size;
    const float* depth = nullptr;   // metres, W x H, row-major, host memory
    const uint8_t* rgb = nullptr;   // optional, unused by the volumetric path
};

/** What tracking and Mask R-CNN would have produced for one frame. */
struct FrameInputs {
    Affine3f cam_pose;                       // camera -> world for this frame
    std::map<int, Affine3f> obj_poses;       // object id -> volume-centre -> world
    std::map<int, emf_image_t> masks;        // object id -> u8 0/1 mask (device), mask frames only
    bool runMasks = false;                   // this frame is a mask frame (frame % maskRCNNFrames)
    // Tracking (reference EMFusion::performTracking, EMFusion.cpp:672-724).  When set, the
    // corresponding supplied pose(s) are ignored: the camera pose is tracked against the
    // background from the previous frame's pose, then every object's pose against its volume.
    bool trackCamera = false;
    bool trackObjects = false;
    // EMFusion::preprocessDepth (EMFusion.cpp:294-305): bilateral filter + NaN / zero patches on
    // the incoming depth.  The reference always runs it; off by default here so that the hot-path
    // frame keeps SURVEY 8(d)'s definition.  processFrame(const RGBD&) always filters.
    bool preprocessDepth = false;
    // Instance masks (device u8 W x H) that matched no existing object: after the raycast and
    // before the integration each one runs through initNewObjVolume (EMFusion.cpp:446-494, 104-109
    // of processFrame); a created object integrates this frame's depth and this mask.
    std::vector<emf_image_t> newObjectMasks;
    // The instance masks of a Mask R-CNN frame (device u8 W x H, MODIFIED in place by the carving
    // step): after the raycast they run through the reference's initOrMatchObjs -- match against
    // the visible models, resolve double matches, carve and spawn the unmatched ones, existence
    // bookkeeping (EMFusion.cpp:329-372, 417-494) -- and the resulting id -> mask map feeds
    // integrateMasks and cleanUpObjs.  Takes the place of `masks` / `newObjectMasks`.
    std::vector<emf_image_t> instanceMasks;
    // The 81 class scores of each instance mask (same order; may be empty): a matched object
    // accumulates them (ObjTSDF::updateClassProbs, EMFusion.cpp:830), which is what ignore_person reads.
    std::vector<std::vector<double>> instanceScores;
    // Run the reference's cleanUpObjs at the end of the frame (EMFusion.cpp:922-980): objects with
    // a low existence probability (mask frames), with too little association mass under their
    // mask, or not visible are deleted.  Needs the visible set on the host (one synchronisation),
    // which is why it is a switch here.
    bool cleanUp = false;
};

/** Outcome of the last tracking run of one model (0 = camera against the background). */
struct TrackResult {
    int iterations = 0;  // LM trial steps evaluated
    int accepted = 0;    // ... of which accepted
    bool converged = false;
    float error = 0.f;   // weighted squared residual sum at the final pose
};

/** Per-stage GPU time of the last processed frame (milliseconds, from HIP events). */
struct FrameTimings {
    float points = 0, estep = 0, raycast = 0, composite = 0, integrate = 0, masks = 0, total = 0;
};

class EMFusion {
public:
    explicit EMFusion(const Params& params, TSDF::Gradients gradients = TSDF::Gradients::OnTheFly,
                      std::shared_ptr<Communicator> comm = nullptr);
    ~EMFusion();

    /** Drop all objects and clear the background (reference EMFusion.cpp:58-68). */
    void reset();

    /**
     * Track the camera against the background volume (reference EMFusion.cpp:673-685) or all
     * objects against the camera (EMFusion.cpp:689-723) with the current association weights.
     * Device-resident Levenberg-Marquardt (emf_hip_trackStep, one launch per iteration, up to
     * params.maxTrackingIter): the host only stops enqueuing when the device reports every model
     * done; the poses are read back once per stage.
     */
    void trackCamera();
    void trackObjects();
    /**
     * Reference EMFusion::initNewObjVolume (EMFusion.cpp:498-557): spawn an object volume from an
     * instance mask (device u8 W x H, non-zero = inside) of the CURRENT frame's points.  Returns
     * the new object id, or -1 if the mask has too few valid points (visibilityThresh), overlaps an
     * existing volume too much (volIOUThresh) or lies too far away (distanceThresh).
     */
    int initNewObjVolume(const emf_image_t& mask);
    /** Reference EMFusion::volumeIOU (EMFusion.cpp:559-611). */
    float volumeIOU(const ObjTSDF& obj, const Vec3f& p10, const Vec3f& p90) const;
    /**
     * Reference EMFusion::matchSegmentation (EMFusion.cpp:797-825): the visible object whose
     * raycast segmentation overlaps `mask` best; its id if the IoU exceeds matchIOUThresh, else
     * -1.  match_iou is updated as in the reference (in/out).
     */
    int matchSegmentation(const emf_image_t& mask, float& match_iou);
    /**
     * Reference EMFusion::initOrMatchObjs (EMFusion.cpp:329-372) without the class scores:
     * returns object id -> mask; `assigned[i]` = the id mask i ended up with (-1: none).
     */
    std::map<int, emf_image_t> initOrMatchObjs(std::vector<emf_image_t>& segs,
                                               std::vector<int>& assigned,
                                               const std::vector<std::vector<double>>& scores = {});
    /**
     * Params.ignore_person of the reference (data.h:198, config/tum.cfg): objects whose most likely
     * class is "person" (COCO index 1) are tracked and fused like all others but left out of the
     * rendering (their pixels show the background, EMFusion.cpp:139-150) and of the mesh files
     * (EMFusion.cpp:274-278, 962-966).
     */
    void setIgnorePerson(bool on) { ignorePerson = on; }
    static bool isPerson(const ObjTSDF& obj) { return obj.getClassID() == 1; }  // class_names[1]
    const std::vector<int>& lastMaskAssignment() const { return lastAssigned; }
    /**
     * Reference EMFusion::updateObj (EMFusion.cpp:827-863): grow / recentre a matched object's
     * volume around its surface and the newly matched points; returns the centre shift (0: none).
     * updateObject(id, mask) is the stand-alone form (looks the object up, refreshes the model table).
     */
    Vec3f updateObj(ObjTSDF& obj, const emf_image_t& mask);
    Vec3f updateObject(int id, const emf_image_t& mask);
    static std::map<int, std::map<int, Affine3f>> addPoseOffsets(
        const std::map<int, std::map<int, Affine3f>>& poses,
        const std::map<int, std::map<int, Vec3f>>& offsets);
    /** Reference EMFusion::cleanUpObjs (EMFusion.cpp:922-980); returns the deleted ids. */
    std::vector<int> cleanUpObjs(bool maskFrame, const std::map<int, emf_image_t>& matches);
    const std::vector<int>& lastDeletedObjects() const { return lastDeleted; }
    /** Reference EMFusion::preprocessDepth (EMFusion.cpp:294-305), one launch. */
    void preprocessDepth(const emf_image_t& depthRaw, const emf_image_t& depthOut);
    /**
     * Wait for the reciprocal checks of volumes created so far and adopt their verdicts.  Volumes created
     * inside frames are checked in the background and divide meanwhile; an application that adds its objects up front
     * (emf_fusion_add_object does this) calls it once so that the checks -- some tens of microseconds per distinct voxel
     * size -- do not run beside its first frames.  Same results either way.
     */
    void settleReciprocals();
    /** Result of the last tracking run of model `id` (0 = camera), or nullptr. */
    const TrackResult* getTrackResult(int id) const;
    /**
     * Keep the camera / object poses of every processed frame (reference EMFusion::storePoses,
     * EMFusion.cpp:322-327) and write them and the volumes in the reference's formats:
     * <dir>/poses-cam.txt, poses-<id>.txt, poses-<id>-corrected.txt (writePoses, EMFusion.cpp:991-1007)
     * and, with volumes, <dir>/mesh_bg.ply, mesh_<id>.ply (writeMeshes, EMFusion.cpp:1147-1156) and
     * <dir>/tsdfs/{bg_tsdf,tsdf_<id>,weights_<id>,fgProbs_<id>}.bin (writeTSDFs, EMFusion.cpp:1187-1218).
     */
    void enablePoseLog(bool on) { poseLog = on; }
    /**
     * Reference EMFusion::setupOutput (EMFusion.cpp:243-247): turns the log on (saveOutput) and
     * chooses whether the volumes are exported too.  With exp_vols the volumes of objects deleted
     * during the run are kept on the host like their mesh (EMFusion.cpp:966-973).  The per-frame
     * mesh export of exp_frame_meshes belongs to the viz path and is not kept.
     * From here on every frame also keeps the reference's per-frame debug images (as PNG bytes, not as raw
     * images): association weights before and after tracking (storeAssocs, EMFusion.cpp:79-91, 307-320), Huber
     * and combined tracking weights of the stages that ran (EMFusion.cpp:110-118; TSDF.cpp:346-354), the
     * objects' foreground-probability look-ups (ObjTSDF.cpp:237-240) and what render() produced
     * (EMFusion.cpp:158-160); writeResults() writes them where writeRenderings / writeAssocs / writeHuberWeights
     * / writeTrackWeights / writeFgProbs put them (EMFusion.cpp:1009-1145).  Each costs a device-to-host copy
     * and a synchronisation per image: a debugging mode, as in the reference.  One-rank path only.
     */
    void setupOutput(bool expFrameMeshes, bool exp_vols) {
        (void)expFrameMeshes;
        poseLog = true;
        saveOutput = true;
        expVols = exp_vols;
    }
    /**
     * Reference EMFusion::writeResults (EMFusion.cpp:248-292): pose files and meshes always, the
     * tsdfs/ dumps only with `volumes` (or setupOutput's exp_vols).
     */
    void writeResults(const std::string& dir, bool volumes);
    /**
     * Prepare for using preprocessed masks (reference EMFusion.h:98, EMFusion.cpp:249-251): from now on
     * processFrame(const RGBD&) reads `<path>/Mask%04d.plk` (numbered by the frame count, EMFusion.cpp:383-389) on
     * every maskRCNNFrames-th frame and runs its instances through initOrMatchObjs, as runMaskRCNN does with a
     * mask path set.  A missing file counts as "no instances" (the reference's loadPreprocessed returns -1).
     */
    void usePreprocMasks(const std::string& path) { maskPath = path; }
    /**
     * Get the last Mask R-CNN segmentation image (reference EMFusion.h:83, EMFusion.cpp:237-240: the visualisation
     * of the last mask frame).  Without the colour frame the instances are drawn in the reference's instance colours
     * (MaskRCNN.cpp:290-301) on black: rgb = W x H x 3 bytes; empty before the first mask frame.  Returns the number
     * of instances of that frame.
     */
    int getLastMasks(std::vector<uint8_t>& rgb) const {
        rgb = lastMaskVis;
        return lastMaskInstances;
    }
    /** Ids returned by initNewObjVolume for FrameInputs::newObjectMasks of the last frame (-1: none). */
    const std::vector<int>& lastCreatedObjects() const { return lastCreated; }
    Affine3f getCameraPose() const { return pose; }
    const ObjTSDF* getObject(int id) const;
    /**
     * Phong rendering of the current model view (reference EMFusion::render, EMFusion.cpp:131-160,
     * without the viz window): rgb = W x H x 3 bytes on the host.  Black before the first frame;
     * after the first frame the models are raycast once for it, as in the reference.
     */
    void render(uint8_t* rgb);
    const std::array<uint8_t, 768>& getColorMap() const { return colorMap; }
    /**
     * Multi-GPU: the depth image enters the node on ONE rank.  With a root >= 0 every frame starts
     * with a broadcast of the depth buffer handed to processFrame (source on `root`, destination on
     * the other ranks: same size and pitch everywhere) over the communicator -- the per-frame
     * broadcast SURVEY 8(e) lists next to the two reductions.  -1 (default): every rank already
     * holds the frame.
     */
    void setDepthBroadcastRoot(int root) { depthRoot = root; }
    /** getMesh() of the background (id 0) or of an object held by this rank. */
    Mesh getMesh(int id);

    /**
     * Create an object volume centred at `center` (world) with edge length `volSize` metres --
     * the geometric tail of the reference's initNewObjVolume (EMFusion.cpp:541-557).  With more
     * than one rank every rank must issue the same calls; only the owning rank allocates the
     * volume.  Returns the object id (1-based, creation order = compositing order).
     */
    int addObject(const Vec3f& center, float volSize);
    int addObject(const Vec3f& center, float volSize, const Vec3i& res);

    /** Poses / masks the next processFrame(RGBD) call consumes. */
    void setFrameInputs(const FrameInputs& in) { pending = in; }
    void loadPreprocMasks(FrameInputs& in);

    /** Reference entry point (EMFusion.cpp:70): uploads the depth map, then runs the schedule. */
    void processFrame(const RGBD& frame);
    /** Same schedule on a depth map already resident in device memory (f32 W x H). */
    void processFrame(const emf_image_t& depthDev, const FrameInputs& in);

    // ---- the four path stages; public so they can be driven and timed one by one ----
    /** E-step over all models (reference EMFusion.cpp:635-670). */
    void computeAssociationWeights();
    /** Raycast all models + compositing + visibility (reference EMFusion.cpp:726-795). */
    void raycast();
    /** Fuse depth into the background and every visible object (reference EMFusion.cpp:865-889). */
    void integrateDepth();
    /** fg/bg update of the matched objects (reference EMFusion.cpp:891-906). */
    void integrateMasks(const std::map<int, emf_image_t>& matches);

    /** Block until everything enqueued so far has finished. */
    void synchronize();
    /** Wait for the work of this instance only (its streams), not for the device. */
    void quiesce();

    // ---- state access ----
    int frameIndex() const { return frameCount; }
    const Params& getParams() const { return params; }
    TSDF& getBackground() { return background; }
    std::list<ObjTSDF>& getObjects() { return objects; }
    ObjTSDF* findObject(int id);
    /** Objects classified visible by the last raycast (waits for the device if necessary). */
    const std::set<int>& visibleObjects();
    /** Voxels swept by the batched integration since the counter was last read (and reset). */
    uint64_t takeIntegratedVoxels();
    bool usesBatchedLaunches() const { return batched; }
    /** Launches per stage on the batched path: 1 up to EMF_MAX_BATCH models, then one per chunk of the model table. */
    int batchedChunks() const { return batched ? launchChunks() : 0; }
    /** The background is integrated out of place on a second stream, beside the raycast. */
    bool overlapsBackground() const { return overlapUsable(); }
    std::vector<int> objectIds() const { return allIds; }
    bool ownsObject(int id) const;
    /** Host seconds processFrame(const RGBD&) has spent so far handing the depth maps to the device (staging memcpy +
     *  enqueue with the double-buffered upload; the blocking copy with EMF_ASYNC_UPLOAD=0), and the number of frames. */
    std::pair<double, uint64_t> uploadHostTime() const { return {uploadHostSeconds, uploads}; }
    const FrameTimings& lastTimings() const { return timings; }
    void enableTimings(bool on) { timingsOn = on; }
    /** Per-launch HIP-event timers (see KernelTimers.hpp); maxLaunches = 0 switches them off. */
    KernelTimers& kernelTimers() { return ktimers; }
    /** Device counters [march samples, hits, gathered samples, fast-forwarded samples]
     *  accumulated by raycast() while enabled. */
    void enableRaycastStats(bool on);
    std::array<uint64_t, 4> raycastStats();

    // device images of the last frame (valid until the next call)
    const DeviceImage<float, 3>& getPoints() const { return points; }
    const DeviceImage<float>& getBgAssociation() const { return bg_associationWeights; }
    const DeviceImage<float>* getObjAssociation(int id) const;
    const DeviceImage<float>& getAssociationNorm() const { return associationNorm; }
    const DeviceImage<float>& getRaylengths() const { return raylengths; }
    const DeviceImage<float, 3>& getVertices() const { return vertices; }
    const DeviceImage<float, 3>& getNormals() const { return normals; }
    const DeviceImage<uint8_t>& getModelSegmentation() const { return modelSegmentation; }
    const DeviceImage<float>& getBgRaylengths() const { return bg_raylengths; }
    const DeviceImage<float>* getObjRaylengths(int id) const;
    Stream& mainStream() { return main; }

private:
    struct ObjImages {
        DeviceImage<float> raylengths;
        DeviceImage<float, 3> vertices, normals;
        DeviceImage<uint8_t> modelSegmentation;
        DeviceImage<float> associationWeights;
    };

    void createObj(int id);
    void runSchedule(const emf_image_t& depthDev, const FrameInputs& in);
    // batched (model-table) path: one launch per stage, no host synchronisation inside a frame
    void rebuildModelTable();
    void adoptReciprocals();
    void posesCO(std::vector<emf_pose_t>& out) const;
    void posesOC(std::vector<emf_pose_t>& out) const;
    void estepBatched();
    void estepSharded(const std::vector<emf_pose_t>& co, bool fromDepth);
    void launchEstep(const std::vector<emf_pose_t>& co, int first, int count, bool fromDepth, int normalize,
                     const emf_image_t* norm, const emf_image_t* objSum);
    bool fusePoints = true;       // the frame's first E-step makes the points (EMF_FUSE_POINTS=0: own launch)
    bool pointsPending = false;   // ... and has not run yet
    bool visCountsClear = true;   // visCounts holds zeros (cleared at construction, left so by the fused pair)
    bool fuseVisibility = true;   // the composite's launch takes the visibility counts (EMF_FUSE_VISIBILITY=0: own launch)
    void raycastBatched();
    void integrateBatched();
    void compositeAndVisibility(bool deviceGate);
    void compositeAcrossRanks(bool deviceGate);
    void refreshVisibleFromDevice();
    // legacy path: one stream per volume, host-side visibility gate (reference structure)
    void estepPerVolume();
    void raycastPerVolume();
    void integratePerVolume();
    void forkVolumeStreams();
    void joinVolumeStreams();
    Stream& streamOf(int key);
    float stamp(int slot);
    double pixels() const;

    Params params;
    TSDF::Gradients gradMode;
    std::shared_ptr<Communicator> comm;
    int rank = 0, world = 1;

    TSDF background;
    std::list<ObjTSDF> objects;          // objects owned by this rank, creation order
    std::vector<int> allIds;             // every object id of the job, creation order
    std::map<int, ObjImages> objImages;  // per owned object
    std::map<int, Stream> streams;       // key 0 = background, else object id
    // Queue priority classes of the frame's three streams: NONE of them in the "normal" class.  HIP serves each class
    // from its own pool of at most four hardware queues, and streams of an embedding application (the reference
    // creates one cv::cuda::Stream per object, EMFusion.h:471) are normal-priority ones: with `main` or `lists` in
    // that class the frame took 0.76-0.91 ms instead of 0.57 for 2, 5, 6 or 9 live foreign streams, and was flat
    // over 0 .. 9 of them with main = high, aux = lists = low (scripts/stream_history_probe.py --matrix3,
    // DESIGN.md section 6; tests/test_gpu_stream_history.py)
    Stream main{streamPriority("EMF_PRIO_MAIN", 1)};
    Affine3f pose;                       // current camera pose
    std::set<int> vis_objs;
    int frameCount = 0;
    int nextId = 1;
    FrameInputs pending;

    // frame-sized device images (reference EMFusion.h:447-489)
    emf_image_t depth{};  // view of the current depth map
    // processFrame(const RGBD&): the host depth map goes through one of TWO pinned staging buffers and, on a copy stream
    // of its own, into one of TWO device images, so that the transfer of frame k + 1 runs while frame k's kernels do
    // (what the reference's reader thread + upload amount to, RGBDReader.cpp:72-117, EMFusion.cpp:72); the frame's
    // `main` stream waits for the copy's event.  EMF_ASYNC_UPLOAD=0: hipMemcpyAsync from the caller's pageable memory on
    // `main` (rounds 1-5: the runtime stages it and the host blocks; A/B measurements).
    struct UploadSlot {
        DeviceImage<float> dev;
        float* pinned = nullptr;
        hipEvent_t copied = nullptr;     // the H2D copy out of `pinned` into `dev` is through
        hipEvent_t frameDone = nullptr;  // the frame that read `dev` is through (recorded on `main` at its end)
        bool copiedValid = false, frameDoneValid = false;
    };
    UploadSlot uploadSlots[2];
    Stream copyStream{streamPriority("EMF_PRIO_COPY", 1)};
    uint64_t uploads = 0;
    bool asyncUpload = true;
    double uploadHostSeconds = 0.0;  // host time processFrame(RGBD) spent getting the depth map on its way (sum)
    emf_image_t stageDepth(const float* host, int& slotOut);
    std::string maskPath;                              // usePreprocMasks
    std::vector<DeviceImage<uint8_t>> preprocMaskDev;  // the instances of the last mask frame (device copies)
    std::vector<uint8_t> lastMaskVis;
    int lastMaskInstances = 0;
    DeviceImage<float> depthFiltered;  // output of preprocessDepth
    DeviceImage<float> invLambda;  // per-pixel 1 / lambda of the integration, fixed by the intrinsics
    bool useLambdaTable = true;
    DeviceBuffer integrateCullScratch;  // survivor list of emf_hip_integrateBatchedCulled (empty: plain launch)
    bool cullBoxes = true;               // EMF_INT_CULL=0 keeps the one-level launch (A/B measurements)
    bool objCull = false;                // EMF_OBJ_CULL=1: two-level launch for the objects alone too (A/B)
    bool ignorePerson = false;
    int depthRoot = -1;  // sharded path: rank whose depth image is broadcast each frame (-1: none)
    bool bgBands = true;  // sharded path: split the background raycast into row bands per rank

    // ---- object creation / matching (SURVEY f-3) ----
    emf_point_stats_t maskedStats(const emf_image_t& mask, const Affine3f& frame);  // synchronises
    DeviceBuffer statsScratch, statsDev, overlapDev;
    std::vector<int> lastCreated, lastDeleted, lastAssigned;
    bool poseLog = false;
    std::map<int, Affine3f> poses;                    // frame -> camera pose
    std::map<int, std::map<int, Affine3f>> obj_poses;  // id -> frame -> pose
    std::map<int, std::map<int, Vec3f>> obj_pose_offsets;  // id -> frame -> centre shift of resize()
    std::array<uint8_t, 768> colorMap = io::randomColors();
    DeviceImage<uint8_t, 3> image;  // rendering
    std::map<int, Mesh> meshes;                            // id -> last mesh (deleted objects keep theirs)
    bool expVols = false;                                  // setupOutput: keep / dump volumes too
    // ---- per-frame debug images of the reference's saveOutput mode, kept as encoded PNGs ----
    bool saveOutput = false;
    using ImageLog = std::map<int, std::vector<uint8_t>>;  // frame -> PNG bytes
    ImageLog renderings, bg_assocWeight_preTrack, bg_assocWeight_postTrack, bg_huberWeights, bg_trackWeights;
    std::map<int, ImageLog> obj_assocWeights_preTrack, obj_assocWeights_postTrack, obj_huberWeights, obj_trackWeights,
        obj_fgProbs;                                       // id -> frame -> PNG bytes
    std::vector<uint8_t> pngOf(const float* dev, size_t pitchBytes);  // x 255 -> u8 -> PNG; synchronises `main`
    void storeAssocs(ImageLog& bg, std::map<int, ImageLog>& objs);    // EMFusion.cpp:307-320
    void storeTrackWeights(int first, int count);                     // behind a tracking stage
    void storeFgProbs();                                              // behind the frame's last E-step
    DeviceBuffer logScratch;                                          // 2 x EMF_MAX_BATCH float images
    struct SavedVolumes {                                  // tsdfs / intWeights / fgProbs / meta of the reference
        std::vector<float> tsdf, weights, fgProbs;
        Vec3i res;
        float voxelSize = 0.f;
    };
    static SavedVolumes saveVolumes(ObjTSDF& obj);
    std::map<int, SavedVolumes> savedVolumes;              // id -> volumes of objects deleted while the log was on
    DeviceBuffer massDev;                      // one emf_hip_maskAssociationMassBytes() block per object (cleanUpObjs)
    void deleteObj(int id);
    void ensureLifecycleBuffers();
    void* lifecycleHost = nullptr;  // pinned: emf_point_stats_t / 513 x u32 / EMF_MAX_MODELS x emf_mask_mass_t

    // ---- tracking (SURVEY f-1) ----
    void trackModels(int first, int count);    // LM-ICP of table slots [first, first + count)
    int trackChunk = 8;                        // iterations per convergence poll (0: never poll)
    int trackPredicted[2] = {0, 0};            // iterations the camera / object stage took last frame
    int trackWindow = 4;                       // launches kept ahead of the device's progress report (0: poll in chunks); 2 ... 6 measured: 4 leaves a stage 3 idle launches instead of 5
    uint32_t* trackWatch = nullptr;            // pinned host words the step kernel reports to
    uint32_t* trackWatchDev = nullptr;         // ... as the device addresses them
    uint32_t trackStageTag = 0;                // upper half of the words of the stage in flight (trackModels)
    DeviceBuffer trackStates;                  // emf_track_state_t[EMF_MAX_MODELS]
    DeviceBuffer trackScratch;                 // one emf_hip_trackScratchBytes block per table slot (grown on demand)
    emf_track_state_t* trackStatesHost = nullptr;  // pinned mirror
    std::map<int, TrackResult> trackResults;   // by model id
    DeviceImage<float, 3> points;
    DeviceImage<float> raylengths, bg_raylengths, associationNorm, bg_associationWeights,
        diffRaylengths, objPartialSum;
    DeviceImage<float, 3> vertices, normals, bg_vertices, bg_normals;
    DeviceImage<uint8_t> modelSegmentation, bg_mask, noObjMask, occludedMask;
    DeviceBuffer visCounts;      // int32 per object
    DeviceBuffer hitKeys;        // u64 W x H, multi-GPU composite merge
    DeviceBuffer raycastStatsDev;  // 2 x u64
    int32_t* visCountsHost = nullptr;  // pinned
    bool statsOn = false;

    // device-resident model table for the batched launches (slot 0 = background)
    bool batched = true;            // false: per-volume launches (see EMFusion.cpp)
    bool forceLegacy = false;
    bool sharded = false;           // objects sharded over ranks: use the cross-rank exchanges
    DeviceBuffer modelTable;        // 2 x emf_model_t[EMF_MAX_MODELS]: [1] has the background's two copies swapped
    int tableSel = 0;               // which of the two describes the background's current front copy
    const emf_model_t* currentTable() const { return modelTable.as<emf_model_t>() + tableSel * EMF_MAX_MODELS; }
    // A launch takes at most EMF_MAX_BATCH table slots (its poses travel by value in the kernel arguments); a longer
    // model list -- the reference loops over any number of objects, EMFusion.cpp:635-670, 726-795, 865-889 -- is served
    // in chunks of the table: f(first slot, count) for the slots [first, n), cut at multiples of EMF_MAX_BATCH, so
    // that the chunk holding slot 0 is the only one with the background in it.
    template <class F>
    static void forChunks(int first, int n, F&& f) {
        while (first < n) {
            const int end = std::min(n, (first / EMF_MAX_BATCH + 1) * EMF_MAX_BATCH);
            f(first, end - first);
            first = end;
        }
    }
    int launchChunks() const { return (static_cast<int>(modelsHost.size()) + EMF_MAX_BATCH - 1) / EMF_MAX_BATCH; }
    uint32_t chunkMask(const std::vector<uint8_t>& flags, int first, int count) const {
        uint32_t m = 0;
        for (int k = 0; k < count; ++k) m |= flags[first + k] ? 1u << k : 0u;
        return m;
    }
    // Background kept twice (TSDF::enableDoubleBuffer): its integration runs out of place on `aux`,
    // concurrently with the raycast of the same frame, and the copies are flipped at the join.
    // EMF_BG_OVERLAP=0 keeps the reference's sequence raycast -> integrate (A/B measurements).
    bool bgOverlap = true;
    bool bgInFlight = false;        // the out-of-place integration of this frame has been enqueued
    bool bgBackStale = false;       // the background was integrated in place: the copies differ
    // the background's sweep yields to the raycast when both have workgroups to place (its long chains should
    // start as early as they can): lowest queue priority for the second stream (+1 % frames/s)
    Stream aux{streamPriority("EMF_PRIO_AUX", -1)};
    // Raycast far bounds (emf_hip_raycastFarBounds): per model and 8x8-pixel cell, where a march may
    // stop because nothing can be hit any more.  EMF_FAR_BOUNDS=0 marches every ray to the end.
    int marchLanes = 1;         // lanes per background ray (EMF_MARCH_ROWS, read by the constructor)
    bool useFootprints = true;  // objects are marched only where their volume box projects to
    bool useFarBounds = true;
    DeviceBuffer farBounds;       // two halves, written alternately (computeFarBounds)
    int farSel = 0;
    float* farBoundsHalf() const { return farBounds.as<float>() + static_cast<size_t>(farSel) * (farBounds.bytes() / 2 / sizeof(float)); }
    int forkFrame = -2;           // frame whose integrateBackgroundAsync forked `aux` (and re-recorded `main`'s event)
    bool farBoundsReady = false;
    bool earlyFarBounds = true;    // far bounds wait for the previous raycast only (EMF_EARLY_FAR_BOUNDS=0: for `main`)
    bool peerFused = false;     // sharded over a direct peer-write transport: exchanges fused into the path's kernels
    int bandRowsPending = 0;    // background raycast bands waiting for the raycast's exchange
    Stream lists{streamPriority("EMF_PRIO_LISTS", -1)};  // relevant-tile list rebuilds: behind the integrations, waited for by the next far bounds
    bool bgListPending = false; // the background was forked; its list rebuild is not enqueued yet
    bool bgPrepared = false;    // bgCullScratch's counter and the next dirtyNext map are already cleared
    void rebuildBackgroundList();
    void computeFarBounds(const std::vector<emf_pose_t>& co);
    void joinFarBounds();
    DeviceBuffer bgCullScratch;     // box list of the background's own launch
    bool overlapUsable() const;
    void integrateBackgroundAsync();  // fork: enqueue on aux what integrateDepth() would do for slot 0
    void joinBackground();
    std::vector<emf_model_t> modelsHost;
    std::vector<int32_t> resHost;   // 3 per model
    std::vector<float> voxelHost;   // voxel size per model (object footprints of the batched raycast)
    std::vector<uint8_t> scanSlot, listSlot;  // far bounds, per table slot: sign maps scanned / keeps a relevant-tile list
    bool anyScan = false;
    DeviceBuffer visibleDev;        // int32 per model slot: integrate gate, written on the device
    DeviceBuffer integrateStatsDev; // u64: voxels swept by integrateBatched
    int32_t* visibleHost = nullptr; // pinned mirror of visCounts for visibleObjects()
    bool visPending = false;
    std::vector<int32_t> visIds;    // object ids in the order of the pending counts

    KernelTimers ktimers;
    bool timingsOn = false;
    FrameTimings timings;
    std::vector<hipEvent_t> stamps;
};

}  // namespace emf
