// ObjTSDF.hpp -- emf::ObjTSDF: an object volume = TSDF + per-voxel foreground probability.
//
// Keeps the per-frame surface of the reference's emf::ObjTSDF (reference
// include/EMFusion/core/ObjTSDF.h:40-216, src/core/ObjTSDF.cpp:167-226): integrateMask,
// computeAssociation (foreground-weighted), raycast (foreground-masked weights), computeFgProbs,
// the getters, resize() (ObjTSDF.cpp:80-165), the existence and class-probability bookkeeping.
#pragma once

#include "TSDF.hpp"

namespace emf {

class ObjTSDF : public TSDF {
public:
    ObjTSDF(Vec3i volumeRes, float voxelSize, float truncdist, Affine3f pose, TSDFParams params,
            Size frameSize, Gradients gradients = Gradients::OnTheFly);
    /** Same, with a caller-chosen ID (multi-GPU runs number objects globally). */
    ObjTSDF(int id, Vec3i volumeRes, float voxelSize, float truncdist, Affine3f pose,
            TSDFParams params, Size frameSize, Gradients gradients = Gradients::OnTheFly);

    bool operator==(const ObjTSDF& o) const { return id == o.id; }
    bool operator!=(const ObjTSDF& o) const { return id != o.id; }
    int getID() const { return id; }
    /** Existence bookkeeping of the reference (ObjTSDF.cpp:61-68). */
    void updateExProb(bool exists) { exCount += exists; nonExCount += 1 - exists; }
    float getExProb() const { return static_cast<float>(exCount) / (exCount + nonExCount); }
    /** Accumulate the class scores of a matched Mask R-CNN detection (reference ObjTSDF.cpp:70-78). */
    void updateClassProbs(const std::vector<double>& scores);
    /** Index of the largest accumulated score (reference ObjTSDF.cpp:242-245); 0 before any score. */
    int getClassID() const;

    /** Also clears the fg/bg counts (reference ObjTSDF.cpp:58-61) and the derived volumes. */
    void reset(const Affine3f& pose) override;

    /**
     * Accumulate fg/bg counts from a 0/1 mask and refresh the foreground probability
     * (reference ObjTSDF::integrateMask, ObjTSDF.cpp:167-179).  mask, occluded: u8 W x H.
     */
    void integrateMask(const emf_image_t& mask, const emf_image_t& occluded_mask,
                       const Affine3f& cam_pose, const Matx33f& intr,
                       Stream& stream = Stream::Null());

    /** Association likelihood times interpolated foreground probability (ObjTSDF.cpp:181-201). */
    void computeAssociation(const emf_image_t& points, const Affine3f& cam_pose,
                            const emf_image_t& associationWeights,
                            Stream& stream = Stream::Null());

    /**
     * Raycast seeing only foreground voxels (reference ObjTSDF::raycast, ObjTSDF.cpp:203-216).
     * The foreground mask is applied inside the weight gather; no raycastWeights volume is built.
     */
    void raycast(const Affine3f& cam_pose, const Matx33f& intr, const emf_image_t& raylengths,
                 const emf_image_t& vertices, const emf_image_t& normals, const emf_image_t& mask,
                 Stream& stream = Stream::Null(), uint64_t* stats = nullptr) override;

    /** fgProbs = fg / (fg + bg), fgVolMask = fgProbs > 0.5 (reference ObjTSDF.cpp:218-226). */
    void computeFgProbs(Stream& stream = Stream::Null());

    /**
     * Grow / recentre the volume when the 10th / 90th percentile box [p10, p90] (volume frame) leaves
     * it (reference ObjTSDF::resize, ObjTSDF.cpp:80-165): new centre = box centre snapped to the
     * voxel grid, new cubic resolution = next even integer >= volPad * longest side / voxelSize,
     * contents shifted with emf_hip_copyValues, pose translated by R * newCentre.  Returns the
     * centre shift in the old volume frame (0 if the box was contained and nothing changed).
     */
    Vec3f resize(const Vec3f& p10, const Vec3f& p90, float volPad, Stream& stream = Stream::Null());

    /** Mesh of the foreground part only (reference ObjTSDF::getMesh, ObjTSDF.cpp:247-268). */
    Mesh getMesh() override;

    std::vector<float> getFgProbVol();
    std::vector<uint8_t> getFgVolMask();
    std::vector<float> getFgBgCounts() const;

    void describe(emf_model_t& m) const override;
    const float* fgProbsPtr() const { return fgProbs.as<float>(); }
    const uint8_t* fgVolMaskPtr() const { return fgVolMask.as<uint8_t>(); }

private:
    int exCount = 0, nonExCount = 0;
    std::vector<double> classProbs;
    static int nextID;
    int id;
    DeviceBuffer fgBgProbs;  // N^3 x 2 f32 counts
    DeviceBuffer fgProbs;    // N^3 f32
    DeviceBuffer fgVolMask;  // N^3 u8 (0/255)
};

}  // namespace emf
