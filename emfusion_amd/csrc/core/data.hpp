// data.hpp -- processing parameters.  Field names, meaning and default values follow the
// reference's emf::TSDFParams / emf::Params (reference include/EMFusion/core/data.h:32-199 and
// config/default.cfg) so a configuration written for the reference maps 1:1; OpenCV types are
// replaced by the PODs of types.hpp.  Parameters of subsystems outside this build's scope
// (Mask R-CNN class filters, tracking, bilateral filter) are kept so the struct stays a drop-in,
// and are marked "unused here".
#pragma once

#include <string>
#include <vector>

#include "types.hpp"

namespace emf {

class TSDFParams {
public:
    // tracking (LM) parameters -- unused here (tracking is the next scope row)
    float tau = 1e3f;
    float eps1 = 1e-8f;
    float eps2 = 1e-8f;
    float nu_init = 2.0f;
    float huberThresh = 0.2f;

    /** Weight capping for TSDF fusion. */
    float maxTSDFWeight = 64.f;
    /** Sigma of the Laplace data likelihood (association weights). */
    float assocSigma = 0.02f;
    /** Mixture parameter of the association likelihood with the uniform prior. */
    float alpha = 0.8f;
    /** Value of the uniform prior. */
    float uniPrior = 1.0f;
};

class Params {
public:
    Params() {
        frameSize = Size(640, 480);
        setDefaultIntrinsics();
        globalVolumeDims = Vec3i::all(512);
        const float volSize = 5.12f;
        globalVoxelSize = volSize / globalVolumeDims[0];
        volumePose = Affine3f().translate(Vec3f(0, 0, volSize / 2));
    }

    /** fx = fy = 525 * W / 640, principal point at the image centre minus half a pixel. */
    void setDefaultIntrinsics() {
        const float f = 525.f * static_cast<float>(frameSize.width) / 640.f;
        intr = Matx33f(f, 0, frameSize.width / 2 - .5f, 0, f, frameSize.height / 2 - .5f, 0, 0, 1);
    }

    Size frameSize;
    Matx33f intr;

    // depth pre-processing (bilateral filter), EMFusion::preprocessDepth
    float bilateral_sigma_depth = 0.04f;
    float bilateral_sigma_spatial = 4.5f;
    int bilateral_kernel_size = 7;

    /** Background model voxel resolution / voxel size [m] / truncation distance [voxels]. */
    Vec3i globalVolumeDims;
    float globalVoxelSize;
    float globalRelTruncDist = 10.f;
    /** Initial object model voxel resolution / truncation distance [voxels]. */
    Vec3i objVolumeDims = Vec3i::all(64);
    float objRelTruncDist = 10.f;

    /** Initial pose of the background volume centre relative to the first camera. */
    Affine3f volumePose;

    float volPad = 2.f;
    int maxTrackingIter = 100;
    /** Mask frames: fg/bg probabilities are integrated every maskRCNNFrames-th frame. */
    int maskRCNNFrames = 30;
    float existenceThresh = 0.1f;
    float volIOUThresh = 0.5f;
    float matchIOUThresh = 0.2f;
    float distanceThresh = 5.f;
    /** Minimum number of segmentation pixels for an object to count as visible. */
    int visibilityThresh = 40 * 40;
    float assocThresh = 0.1f;
    /** Image border (pixels) ignored by the visibility count. */
    int boundary = 20;

    TSDFParams tsdfParams;

    std::vector<std::string> FILTER_CLASSES;  // unused here (Mask R-CNN)
    std::vector<std::string> STATIC_OBJECTS;  // unused here
    bool ignore_person = false;               // unused here
};

}  // namespace emf
