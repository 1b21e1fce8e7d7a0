// TSDF.cpp -- emf::TSDF over the emf_hip_* C ABI (see TSDF.hpp).
#include "TSDF.hpp"

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <mutex>

namespace emf {

TSDF::TSDF(Vec3i _volumeRes, float _voxelSize, float _truncdist, Affine3f _pose,
           TSDFParams _params, Size _frameSize, Gradients gradients)
    : params(_params),
      volumeRes(_volumeRes),
      voxelSize(_voxelSize),
      truncdist(_truncdist),
      gradMode(gradients),
      frameSize(_frameSize),
      tsdfVol(voxels() * sizeof(float)),
      tsdfWeights(voxels() * sizeof(float)),
      brickFlags(2 * static_cast<size_t>((_volumeRes[0] + 3) / 4) * ((_volumeRes[1] + 3) / 4) *
                 ((_volumeRes[2] + 3) / 4)) {  // raw flags + dilated flags
    if (gradMode == Gradients::Materialized) tsdfGrads = DeviceBuffer(voxels() * 3 * sizeof(float));
    obtainReciprocal();
    reset(_pose);
}

// ---- checked reciprocal of the voxel size ---------------------------------------------------------
// Is 1 / voxelSize usable in place of the march's divisions?  An exhaustive device check per distinct
// voxel size and process (emf_hip_voxelReciprocal*); EMF_VOXEL_RCP=0 keeps the divisions for A/B runs.
namespace {
std::atomic<bool> g_deferReciprocal{false};
std::mutex g_rcpSlotMutex;
unsigned long long* g_rcpSlots = nullptr;  // pinned host words the verdicts are COPIED to, allocated once, never freed
unsigned long long* g_rcpSlotsDev = nullptr;  // the device counters the check adds into (no atomics across PCIe)
constexpr int kRcpSlots = 256;
bool g_rcpSlotUsed[kRcpSlots] = {};
hipStream_t g_rcpStream = nullptr;         // lowest priority: the check must not delay a frame

int take_rcp_slot() {
    std::lock_guard<std::mutex> lock(g_rcpSlotMutex);
    if (!g_rcpSlots) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(unsigned long long) * kRcpSlots, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        g_rcpSlots = static_cast<unsigned long long*>(p);
        void* dp = nullptr;
        if (hipMalloc(&dp, sizeof(unsigned long long) * kRcpSlots) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        g_rcpSlotsDev = static_cast<unsigned long long*>(dp);
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&g_rcpStream, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            g_rcpStream = nullptr;
            return -1;
        }
    }
    if (!g_rcpStream || !g_rcpSlotsDev) return -1;
    for (int i = 0; i < kRcpSlots; ++i)
        if (!g_rcpSlotUsed[i]) {
            g_rcpSlotUsed[i] = true;
            return i;
        }
    return -1;
}
void give_rcp_slot(int i) {
    std::lock_guard<std::mutex> lock(g_rcpSlotMutex);
    g_rcpSlotUsed[i] = false;
}
}  // namespace

struct TSDF::PendingReciprocal {
    int slot = -1;
    hipEvent_t done = nullptr;
};
void TSDF::PendingDeleter::operator()(PendingReciprocal* p) const {
    if (!p) return;
    // the check may still be running and will write its slot: wait for THAT kernel (rare: a volume
    // destroyed within milliseconds of its creation), then hand the slot back
    if (p->done) {
        (void)hipEventSynchronize(p->done);
        (void)hipEventDestroy(p->done);
    }
    if (p->slot >= 0) give_rcp_slot(p->slot);
    delete p;
}

void TSDF::deferReciprocalChecks(bool on) { g_deferReciprocal = on; }

void TSDF::obtainReciprocal() {
    rcpVoxel = 0.f;
    const char* vr = std::getenv("EMF_VOXEL_RCP");
    if (vr && vr[0] == '0') return;
    const int known = emf_hip_voxelReciprocalCached(voxelSize, &rcpVoxel);
    if (known == EMF_OK) return;            // this size has been checked in this process
    if (known != EMF_E_NOTREADY) return;    // outside the checked range: the march divides
    if (g_deferReciprocal) {
        const int slot = take_rcp_slot();
        if (slot >= 0) {
            std::unique_ptr<PendingReciprocal, PendingDeleter> p(new PendingReciprocal);
            p->slot = slot;
            if (hipEventCreateWithFlags(&p->done, hipEventDisableTiming) == hipSuccess &&
                emf_hip_voxelReciprocalBegin(voxelSize, g_rcpSlotsDev + slot,
                                             reinterpret_cast<emf_stream_t>(g_rcpStream)) == EMF_OK &&
                hipMemcpyAsync(g_rcpSlots + slot, g_rcpSlotsDev + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               g_rcpStream) == hipSuccess &&
                hipEventRecord(p->done, g_rcpStream) == hipSuccess) {
                pendingRcp = std::move(p);
                return;  // rcpVoxel stays 0 until pollReciprocal() sees the verdict
            }
            (void)hipGetLastError();
        }
    }
    if (emf_hip_voxelReciprocal(voxelSize, &rcpVoxel) != EMF_OK) rcpVoxel = 0.f;
}

bool TSDF::pollReciprocal() {
    if (!pendingRcp) return false;
    if (hipEventQuery(pendingRcp->done) != hipSuccess) {
        (void)hipGetLastError();  // not ready
        return false;
    }
    const unsigned long long bad = g_rcpSlots[pendingRcp->slot];
    pendingRcp.reset();
    if (emf_hip_voxelReciprocalEnd(voxelSize, bad, &rcpVoxel) != EMF_OK) rcpVoxel = 0.f;
    return rcpVoxel != 0.f;
}

void TSDF::settleReciprocal() {
    if (pendingRcp && pendingRcp->done) (void)hipEventSynchronize(pendingRcp->done);
}

void TSDF::reset(const Affine3f& _pose) {
    Stream& s = Stream::Null();
    tsdfVol.setZero(s);
    tsdfWeights.setZero(s);
    if (!tsdfGrads.empty()) tsdfGrads.setZero(s);
    emfCheck(emf_hip_resetBrickFlags(brickFlags.as<uint8_t>(), volumeRes.val, s.abi()),
             "TSDF::reset");
    if (volumeRes[0] % 4 == 0) {  // an all-zero volume has neither sign anywhere
        if (signMaps.empty()) signMaps = DeviceBuffer(emf_hip_signMapBytes(volumeRes.val));
        signMaps.setZero(s);
        signMapsValid = true;
        if (relevantTiles.empty()) relevantTiles = DeviceBuffer(emf_hip_relevantTileBytes(volumeRes.val));
        relevantTiles.setZero(s);  // count 0: nothing can be hit in an empty volume
        // ... and every tile of it is unseen (allocation rounded up to whole words for the fill)
        if (unseenTiles.empty()) unseenTiles = DeviceBuffer((emf_hip_unseenTileBytes(volumeRes.val) + 3) / 4 * 4);
        unseenTiles.fill32(0x01010101u, s);
    } else {
        signMaps = DeviceBuffer();
        signMapsValid = false;
        relevantTiles = DeviceBuffer();
        unseenTiles = DeviceBuffer();
    }
    if (doubleBuffered()) {  // equal copies, clean maps
        tsdfBack.setZero(s);
        weightsBack.setZero(s);
        dirtyMaps[0].setZero(s);
        dirtyMaps[1].setZero(s);
    }
    s.waitForCompletion();  // per-volume streams are non-blocking: do not race the clears
    pose = _pose;
}

void TSDF::refreshSignMaps(Stream& stream) {
    if (signMapsValid || volumeRes[0] % 4 != 0) return;
    const size_t bytes = emf_hip_signMapBytes(volumeRes.val);
    if (signMaps.empty() || signMaps.bytes() != bytes) signMaps = DeviceBuffer(bytes);
    emfCheck(emf_hip_rebuildSignMaps(tsdfVol.as<float>(), volumeRes.val, signMaps.as<uint8_t>(), stream.abi()),
             "TSDF::refreshSignMaps");
    signMapsValid = true;
    const size_t rb = emf_hip_relevantTileBytes(volumeRes.val);
    if (relevantTiles.empty() || relevantTiles.bytes() != rb) relevantTiles = DeviceBuffer(rb);
    relevantTiles.setZero(stream);  // the owner rebuilds the list (emf_hip_updateRelevantTiles) before it is used
    const size_t ub = (emf_hip_unseenTileBytes(volumeRes.val) + 3) / 4 * 4;
    if (unseenTiles.empty() || unseenTiles.bytes() != ub) unseenTiles = DeviceBuffer(ub);
    emfCheck(emf_hip_rebuildUnseenTiles(tsdfVol.as<float>(), tsdfWeights.as<float>(), volumeRes.val,
                                        unseenTiles.as<uint8_t>(), stream.abi()),
             "TSDF::refreshSignMaps");
}

void TSDF::enableDoubleBuffer() {
    if (doubleBuffered()) return;
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    tsdfBack = DeviceBuffer(voxels() * sizeof(float));
    weightsBack = DeviceBuffer(voxels() * sizeof(float));
    hipCheck(hipMemcpy(tsdfBack.data(), tsdfVol.data(), voxels() * sizeof(float), hipMemcpyDeviceToDevice),
             "TSDF::enableDoubleBuffer");
    hipCheck(hipMemcpy(weightsBack.data(), tsdfWeights.data(), voxels() * sizeof(float), hipMemcpyDeviceToDevice),
             "TSDF::enableDoubleBuffer");
    const size_t bytes = emf_hip_integrateDirtyMapBytes(volumeRes.val);
    for (auto& d : dirtyMaps) {
        d = DeviceBuffer(bytes);
        d.setZero(Stream::Null());
    }
    Stream::Null().waitForCompletion();
    dirtyPrev = 0;
}

emf_volume_out_t TSDF::backBuffers() const {
    emf_volume_out_t o;
    o.tsdf = tsdfBack.as<float>();
    o.weights = weightsBack.as<float>();
    o.dirtyPrev = dirtyMaps[dirtyPrev].as<uint8_t>();
    o.dirtyNext = dirtyMaps[1 - dirtyPrev].as<uint8_t>();
    return o;
}

void TSDF::resyncBack() {
    if (!doubleBuffered()) return;
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    hipCheck(hipMemcpy(tsdfBack.data(), tsdfVol.data(), voxels() * sizeof(float), hipMemcpyDeviceToDevice),
             "TSDF::resyncBack");
    hipCheck(hipMemcpy(weightsBack.data(), tsdfWeights.data(), voxels() * sizeof(float), hipMemcpyDeviceToDevice),
             "TSDF::resyncBack");
    dirtyMaps[0].setZero(Stream::Null());
    dirtyMaps[1].setZero(Stream::Null());
    Stream::Null().waitForCompletion();
}

void TSDF::flip() {
    std::swap(tsdfVol, tsdfBack);
    std::swap(tsdfWeights, weightsBack);
    dirtyPrev = 1 - dirtyPrev;
}

void TSDF::getCorners(Vec3f& low, Vec3f& high) const {
    // (res - 1) * voxelSize / 2 (reference TSDF.cpp:84-89)
    const Vec3f corner(static_cast<float>(volumeRes[0] - 1) * voxelSize / 2,
                       static_cast<float>(volumeRes[1] - 1) * voxelSize / 2,
                       static_cast<float>(volumeRes[2] - 1) * voxelSize / 2);
    low = -corner;
    high = corner;
}

Vec3f TSDF::getVolumeSize() const {
    return Vec3f(static_cast<float>(volumeRes[0]) * voxelSize,
                 static_cast<float>(volumeRes[1]) * voxelSize,
                 static_cast<float>(volumeRes[2]) * voxelSize);
}

void TSDF::integrate(const emf_image_t& depth, const emf_image_t& weights,
                     const Affine3f& cam_pose, const Matx33f& intr, Stream& stream,
                     const emf_image_t* invLambda) {
    const Affine3f rel_pose_OC = cam_pose.inv() * pose;  // volume -> camera
    signMapsValid = false;  // the per-volume launch does not keep them
    emfCheck(emf_hip_updateTSDF(&depth, &weights, tsdfVol.as<float>(), tsdfWeights.as<float>(),
                                brickFlagMode() ? brickFlags.as<uint8_t>() : nullptr,
                                rel_pose_OC.rotation().val, rel_pose_OC.translation().val,
                                intr.val, volumeRes.val, voxelSize, truncdist,
                                params.maxTSDFWeight, invLambda, stream.abi()),
             "TSDF::integrate");
}

void TSDF::updateGradients(Stream& stream) {
    if (gradMode != Gradients::Materialized) return;
    emfCheck(emf_hip_computeTSDFGrads(tsdfVol.as<float>(), tsdfGrads.as<float>(), volumeRes.val,
                                      stream.abi()),
             "TSDF::updateGradients");
}

void TSDF::raycast(const Affine3f& cam_pose, const Matx33f& intr, const emf_image_t& raylengths,
                   const emf_image_t& vertices, const emf_image_t& normals,
                   const emf_image_t& mask, Stream& stream, uint64_t* stats) {
    pollReciprocal();  // a volume used outside an emf::EMFusion adopts its deferred verdict here
    const Affine3f rel_pose_CO = pose.inv() * cam_pose;  // camera -> volume
    emfCheck(emf_hip_raycastTSDF(tsdfVol.as<float>(), gradsPtr(), tsdfWeights.as<float>(), nullptr,
                                 brickFlagMode() ? brickFlags.as<uint8_t>() : nullptr, &raylengths, &vertices, &normals, &mask,
                                 rel_pose_CO.rotation().val, rel_pose_CO.translation().val,
                                 intr.val, volumeRes.val, voxelSize, truncdist, rcpVoxel, stats,
                                 stream.abi()),
             "TSDF::raycast");
}

void TSDF::computeAssociation(const emf_image_t& points, const Affine3f& cam_pose,
                              const emf_image_t& associationWeights, Stream& stream) {
    const Affine3f rel_pose_CO = pose.inv() * cam_pose;
    emfCheck(emf_hip_computeAssociation(tsdfVol.as<float>(), nullptr, &points,
                                        rel_pose_CO.rotation().val,
                                        rel_pose_CO.translation().val, volumeRes.val, voxelSize,
                                        truncdist, params.assocSigma, params.alpha,
                                        params.uniPrior, &associationWeights, stream.abi()),
             "TSDF::computeAssociation");
}

int TSDF::brickFlagMode() {  // (read on every call: an instance picks the switch up when it is constructed)
    const char* e = debugEnv("EMF_BRICK_FLAGS");
    return e ? (e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0)) : 0;
}

void TSDF::describe(emf_model_t& m) const {
    m.tsdf = tsdfVol.as<float>();
    m.weights = tsdfWeights.as<float>();
    m.grads = gradsPtr();
    m.fgProbs = nullptr;
    m.fgVolMask = nullptr;
    m.res[0] = volumeRes[0];
    m.res[1] = volumeRes[1];
    m.res[2] = volumeRes[2];
    m.id = 0;
    m.voxelSize = voxelSize;
    m.truncdist = truncdist;
    m.maxWeight = params.maxTSDFWeight;
    m.assocC1 = -truncdist / params.assocSigma;         // reference TSDF.cpp:151
    m.assocC2 = 1.f / (2.f * params.assocSigma);        // reference TSDF.cpp:154
    m.alpha = params.alpha;
    m.assocC3 = (1 - params.alpha) * params.uniPrior;   // reference TSDF.cpp:133
    // EMF_BRICK_FLAGS selects how the brick uniformity flags are used by the class-level path:
    //   0 (default) not maintained, not used -- on the bench scene the march time is set by a few
    //     hundred image-border rays that graze seen/unseen space through MIXED bricks, so skipping
    //     work elsewhere does not shorten the kernel while maintaining the flags costs ~0.2 ms
    //   1 maintained by integrate, raycast fast-forwards through deep-uniform bricks
    //   2 as 1, and uniform lookups are answered from the flags without gathering
    const int mode = brickFlagMode();
    m.brickFlags = mode ? brickFlags.as<uint8_t>() : nullptr;
    m.reserved = mode == 2 ? 2 : 0;
    m.rcpVoxel = rcpVoxel;
    m.signMaps = signMapsValid && !signMaps.empty() ? signMaps.as<uint8_t>() : nullptr;
    // a list pays for large volumes (the far bounds then project a few thousand tiles instead of scanning
    // 65 536 neighbourhoods); an object volume's thousand tiles are scanned faster than a list is kept
    m.relevantTiles = m.signMaps && !relevantTiles.empty() && emf_hip_signMapBytes(volumeRes.val) / 2 >= 8192
                          ? relevantTiles.as<uint32_t>()
                          : nullptr;
    // (kept valid together with the sign maps: the same launches maintain both)
    const char* ut = std::getenv("EMF_UNSEEN_TILES");
    const bool useUnseen = !(ut && ut[0] == '0');
    m.unseenTiles = useUnseen && m.signMaps && !unseenTiles.empty() ? unseenTiles.as<uint8_t>() : nullptr;
    m.pad_ = 0;
}

Mesh TSDF::getMesh() { return extractMesh(nullptr); }

// count -> read back two numbers -> emit; the gradient volume is used when it is materialised
Mesh TSDF::extractMesh(const uint8_t* fgVolMask) {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    Stream& s = Stream::Null();
    DeviceBuffer scratch(std::max<size_t>(emf_hip_meshScratchBytes(volumeRes.val), 8));
    DeviceBuffer countsDev(sizeof(emf_mesh_counts_t));
    emfCheck(emf_hip_meshCount(tsdfVol.as<float>(), tsdfWeights.as<float>(), fgVolMask, volumeRes.val,
                               scratch.data(), countsDev.as<emf_mesh_counts_t>(), s.abi()),
             "TSDF::getMesh");
    emf_mesh_counts_t counts{};
    countsDev.download(&counts, s);
    Mesh mesh;
    if (counts.vertices == 0) return mesh;
    DeviceBuffer v(counts.vertices * 3 * sizeof(float)), n(counts.vertices * 3 * sizeof(float)),
        t(std::max<size_t>(counts.triangles, 1) * 4 * sizeof(int32_t));
    emfCheck(emf_hip_meshEmit(tsdfVol.as<float>(), gradsPtr(), tsdfWeights.as<float>(), fgVolMask,
                              volumeRes.val, voxelSize, scratch.data(), v.as<float>(), n.as<float>(),
                              t.as<int32_t>(), s.abi()),
             "TSDF::getMesh");
    mesh.cloud.resize(counts.vertices * 3);
    mesh.normals.resize(counts.vertices * 3);
    mesh.polygons.resize(static_cast<size_t>(counts.triangles) * 4);
    v.download(mesh.cloud.data(), s);
    n.download(mesh.normals.data(), s);
    if (counts.triangles) {
        std::vector<int32_t> all(std::max<size_t>(counts.triangles, 1) * 4);
        t.download(all.data(), s);
        mesh.polygons.assign(all.begin(), all.begin() + static_cast<size_t>(counts.triangles) * 4);
    }
    return mesh;
}

std::vector<float> TSDF::getTSDF() const {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::vector<float> h(voxels());
    tsdfVol.download(h.data(), Stream::Null());
    return h;
}

std::vector<float> TSDF::getWeightsVol() const {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::vector<float> h(voxels());
    tsdfWeights.download(h.data(), Stream::Null());
    return h;
}

}  // namespace emf
