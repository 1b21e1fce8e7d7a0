// ObjTSDF.cpp -- emf::ObjTSDF over the emf_hip_* C ABI (see ObjTSDF.hpp).
#include "ObjTSDF.hpp"

#include <algorithm>
#include <cmath>

namespace emf {

int ObjTSDF::nextID = 0;

ObjTSDF::ObjTSDF(Vec3i _volumeRes, float _voxelSize, float _truncdist, Affine3f _pose,
                 TSDFParams _params, Size _frameSize, Gradients gradients)
    : ObjTSDF(++nextID, _volumeRes, _voxelSize, _truncdist, _pose, _params, _frameSize,
              gradients) {}

ObjTSDF::ObjTSDF(int _id, Vec3i _volumeRes, float _voxelSize, float _truncdist, Affine3f _pose,
                 TSDFParams _params, Size _frameSize, Gradients gradients)
    : TSDF(_volumeRes, _voxelSize, _truncdist, _pose, _params, _frameSize, gradients),
      id(_id),
      fgBgProbs(voxels() * 2 * sizeof(float)),
      fgProbs(voxels() * sizeof(float)),
      fgVolMask(voxels()) {
    if (_id > nextID) nextID = _id;
    reset(_pose);
}

void ObjTSDF::reset(const Affine3f& _pose) {
    TSDF::reset(_pose);
    Stream& s = Stream::Null();
    fgBgProbs.setZero(s);
    // The reference leaves fgProbs uninitialised and fgVolMask empty until the first
    // integrateMask(), which always runs in the frame that creates the object
    // (EMFusion.cpp:100-106).  Here they start as "no foreground seen yet".
    fgProbs.setZero(s);
    fgVolMask.setZero(s);
    s.waitForCompletion();
}

void ObjTSDF::integrateMask(const emf_image_t& mask, const emf_image_t& occluded_mask,
                            const Affine3f& cam_pose, const Matx33f& intr, Stream& stream) {
    const Affine3f rel_pose = cam_pose.inv() * pose;
    emfCheck(emf_hip_updateFgBgProbs(&mask, &occluded_mask, tsdfVol.as<float>(),
                                     tsdfWeights.as<float>(), fgBgProbs.as<float>(),
                                     rel_pose.rotation().val, rel_pose.translation().val,
                                     intr.val, volumeRes.val, voxelSize, stream.abi()),
             "ObjTSDF::integrateMask");
    computeFgProbs(stream);
}

void ObjTSDF::computeAssociation(const emf_image_t& points, const Affine3f& cam_pose,
                                 const emf_image_t& associationWeights, Stream& stream) {
    const Affine3f rel_pose_CO = pose.inv() * cam_pose;
    emfCheck(emf_hip_computeAssociation(tsdfVol.as<float>(), fgProbs.as<float>(), &points,
                                        rel_pose_CO.rotation().val,
                                        rel_pose_CO.translation().val, volumeRes.val, voxelSize,
                                        truncdist, params.assocSigma, params.alpha,
                                        params.uniPrior, &associationWeights, stream.abi()),
             "ObjTSDF::computeAssociation");
}

void ObjTSDF::raycast(const Affine3f& cam_pose, const Matx33f& intr, const emf_image_t& raylengths,
                      const emf_image_t& vertices, const emf_image_t& normals,
                      const emf_image_t& mask, Stream& stream, uint64_t* stats) {
    pollReciprocal();  // (see TSDF::raycast)
    const Affine3f rel_pose_CO = pose.inv() * cam_pose;
    emfCheck(emf_hip_raycastTSDF(tsdfVol.as<float>(), gradsPtr(), tsdfWeights.as<float>(),
                                 fgVolMask.as<uint8_t>(),
                                 brickFlagMode() ? brickFlags.as<uint8_t>() : nullptr, &raylengths, &vertices, &normals, &mask,
                                 rel_pose_CO.rotation().val, rel_pose_CO.translation().val,
                                 intr.val, volumeRes.val, voxelSize, truncdist, rcpVoxel, stats,
                                 stream.abi()),
             "ObjTSDF::raycast");
}

void ObjTSDF::computeFgProbs(Stream& stream) {
    emfCheck(emf_hip_computeFgProbs(fgBgProbs.as<float>(), fgProbs.as<float>(),
                                    fgVolMask.as<uint8_t>(), volumeRes.val, stream.abi()),
             "ObjTSDF::computeFgProbs");
}

Vec3f ObjTSDF::resize(const Vec3f& p10, const Vec3f& p90, float volPad, Stream& stream) {
    Vec3f half;  // corners of the voxel-centre box, ObjTSDF.cpp:82-87
    for (int i = 0; i < 3; ++i) half[i] = (static_cast<float>(volumeRes[i]) - 1.f) * .5f * voxelSize;
    bool contained = true;
    for (int i = 0; i < 3 && contained; ++i) contained = !(p10[i] < -half[i] || p90[i] > half[i]);
    if (contained) return Vec3f::all(0.f);

    Vec3f newCenter = (p10 + p90) / 2.f;
    const Vec3f centerVox = newCenter / voxelSize;
    Vec3i pixOffset;  // cv::Vec3f -> cv::Vec3i rounds to nearest even (cvRound)
    for (int i = 0; i < 3; ++i) pixOffset[i] = static_cast<int>(std::lrintf(centerVox[i]));
    for (int i = 0; i < 3; ++i) newCenter[i] = static_cast<float>(pixOffset[i]) * voxelSize;
    pose = pose.translate(pose.rotation() * newCenter);

    const Vec3f dims = p90 - p10;
    const float newVolSize = volPad * std::max(dims[0], std::max(dims[1], dims[2])) / voxelSize;
    const int n = (static_cast<int>(std::ceil(newVolSize)) + 1) / 2 * 2;  // next even >= size
    const Vec3i newRes = Vec3i::all(n);
    if (n < 2) throw HipError("ObjTSDF::resize: degenerate extent", EMF_E_ARG);
    for (int i = 0; i < 3; ++i) pixOffset[i] -= (newRes[i] - volumeRes[i]) / 2;

    const size_t nv = static_cast<size_t>(n) * n * n;
    DeviceBuffer newVol(nv * sizeof(float)), newWeights(nv * sizeof(float)),
        newFgBg(nv * 2 * sizeof(float));
    auto shift = [&](const DeviceBuffer& src, DeviceBuffer& dst, int channels) {
        emfCheck(emf_hip_copyValues(src.as<float>(), dst.as<float>(), channels, pixOffset.val,
                                    volumeRes.val, newRes.val, stream.abi()),
                 "ObjTSDF::resize");
    };
    shift(tsdfVol, newVol, 1);
    shift(tsdfWeights, newWeights, 1);
    shift(fgBgProbs, newFgBg, 2);
    DeviceBuffer newGrads;
    if (gradMode == Gradients::Materialized) {
        newGrads = DeviceBuffer(nv * 3 * sizeof(float));
        shift(tsdfGrads, newGrads, 3);
    }
    stream.waitForCompletion();  // the old buffers are released below
    tsdfVol = std::move(newVol);
    tsdfWeights = std::move(newWeights);
    tsdfGrads = std::move(newGrads);
    fgBgProbs = std::move(newFgBg);
    volumeRes = newRes;
    fgProbs = DeviceBuffer(nv * sizeof(float));
    fgVolMask = DeviceBuffer(nv);
    const size_t nb = static_cast<size_t>((n + 3) / 4) * ((n + 3) / 4) * ((n + 3) / 4);
    brickFlags = DeviceBuffer(2 * nb);
    brickFlags.setZero(stream);  // every brick "mixed": always correct; integrate() refines them
    signMaps = DeviceBuffer();   // new resolution, shifted values: rebuilt by refreshSignMaps()
    signMapsValid = false;
    relevantTiles = DeviceBuffer();
    unseenTiles = DeviceBuffer();
    computeFgProbs(stream);
    return newCenter;
}

void ObjTSDF::updateClassProbs(const std::vector<double>& scores) {
    if (scores.empty()) return;  // a mask that came without scores
    if (classProbs.empty()) {
        classProbs = scores;
        return;
    }
    if (classProbs.size() != scores.size())
        throw HipError("ObjTSDF::updateClassProbs: score vectors of different lengths", EMF_E_ARG);
    for (size_t i = 0; i < scores.size(); ++i) classProbs[i] += scores[i];
}

int ObjTSDF::getClassID() const {
    return static_cast<int>(std::max_element(classProbs.begin(), classProbs.end()) - classProbs.begin());
}

Mesh ObjTSDF::getMesh() { return extractMesh(fgVolMask.as<uint8_t>()); }

void ObjTSDF::describe(emf_model_t& m) const {
    TSDF::describe(m);
    m.fgProbs = fgProbs.as<float>();
    m.fgVolMask = fgVolMask.as<uint8_t>();
    m.id = id;
}

std::vector<float> ObjTSDF::getFgProbVol() {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::vector<float> h(voxels());
    fgProbs.download(h.data(), Stream::Null());
    return h;
}

std::vector<uint8_t> ObjTSDF::getFgVolMask() {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::vector<uint8_t> h(voxels());
    fgVolMask.download(h.data(), Stream::Null());
    return h;
}

std::vector<float> ObjTSDF::getFgBgCounts() const {
    hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::vector<float> h(voxels() * 2);
    fgBgProbs.download(h.data(), Stream::Null());
    return h;
}

}  // namespace emf
