// TSDF.cpp -- emf::TSDF over the emf_hip_* C ABI (see TSDF.hpp).
#include "TSDF.hpp"

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
      tsdfWeights(voxels() * sizeof(float)) {
    if (gradMode == Gradients::Materialized) tsdfGrads = DeviceBuffer(voxels() * 3 * sizeof(float));
    reset(_pose);
}

void TSDF::reset(const Affine3f& _pose) {
    Stream& s = Stream::Null();
    tsdfVol.setZero(s);
    tsdfWeights.setZero(s);
    if (!tsdfGrads.empty()) tsdfGrads.setZero(s);
    s.waitForCompletion();  // per-volume streams are non-blocking: do not race the clears
    pose = _pose;
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
                     const Affine3f& cam_pose, const Matx33f& intr, Stream& stream) {
    const Affine3f rel_pose_OC = cam_pose.inv() * pose;  // volume -> camera
    emfCheck(emf_hip_updateTSDF(&depth, &weights, tsdfVol.as<float>(), tsdfWeights.as<float>(),
                                rel_pose_OC.rotation().val, rel_pose_OC.translation().val,
                                intr.val, volumeRes.val, voxelSize, truncdist,
                                params.maxTSDFWeight, stream.abi()),
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
    const Affine3f rel_pose_CO = pose.inv() * cam_pose;  // camera -> volume
    emfCheck(emf_hip_raycastTSDF(tsdfVol.as<float>(), gradsPtr(), tsdfWeights.as<float>(), nullptr,
                                 &raylengths, &vertices, &normals, &mask,
                                 rel_pose_CO.rotation().val, rel_pose_CO.translation().val,
                                 intr.val, volumeRes.val, voxelSize, truncdist, stats,
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
