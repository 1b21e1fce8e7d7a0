// ObjTSDF.cpp -- emf::ObjTSDF over the emf_hip_* C ABI (see ObjTSDF.hpp).
#include "ObjTSDF.hpp"

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
