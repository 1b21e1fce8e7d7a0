// TSDF.hpp -- emf::TSDF: one TSDF volume with its per-frame operations.
//
// Keeps the method surface of the reference's emf::TSDF for the per-frame volumetric path
// (reference include/EMFusion/core/TSDF.h:39-330, src/core/TSDF.cpp:28-168): constructor
// arguments, integrate / updateGradients / raycast / computeAssociation / reset and the getters.
// Every method enqueues hand-written gfx950 kernels through the emf_hip_* C ABI on the given
// stream and returns; nothing here computes on the host.  Tracking (prepareTracking ...
// syncTrack) and meshing are outside this build's scope.
#pragma once

#include <memory>
#include <vector>

#include "data.hpp"
#include "types.hpp"

namespace emf {

class TSDF {
public:
    /** How raycast() obtains the surface normal. */
    enum class Gradients {
        OnTheFly,     // blend forward differences of the TSDF at the hit (no gradient volume)
        Materialized  // keep the reference's N^3 x 3 gradient volume, rebuilt by updateGradients()
    };

    TSDF(Vec3i volumeRes, float voxelSize, float truncdist, Affine3f pose, TSDFParams params,
         Size frameSize, Gradients gradients = Gradients::OnTheFly);
    virtual ~TSDF() = default;
    TSDF(TSDF&&) = default;
    TSDF& operator=(TSDF&&) = default;

    /** Zero the volume (tsdf, weights, gradients) and set its pose (reference TSDF.cpp:77-82). */
    virtual void reset(const Affine3f& pose);

    void getCorners(Vec3f& low, Vec3f& high) const;
    Vec3f getVolumeSize() const;
    Vec3i getVolumeRes() const { return volumeRes; }
    float getVoxelSize() const { return voxelSize; }
    float getTruncDist() const { return truncdist; }
    Affine3f getPose() const { return pose; }
    /** Externally supplied pose (stands in for the tracking update of syncTrack()). */
    void setPose(const Affine3f& p) { pose = p; }
    size_t voxels() const {
        return static_cast<size_t>(volumeRes[0]) * volumeRes[1] * volumeRes[2];
    }

    /**
     * Fuse one depth frame, weighted per pixel by the association weights
     * (reference TSDF::integrate, TSDF.cpp:108-118).  depth, weights: f32 W x H device images.
     * invLambda: optional emf_hip_computeInvLambda table for `intr` (same results, fewer
     * instructions per voxel).
     */
    void integrate(const emf_image_t& depth, const emf_image_t& weights, const Affine3f& cam_pose,
                   const Matx33f& intr, Stream& stream = Stream::Null(),
                   const emf_image_t* invLambda = nullptr);

    /**
     * Refresh the gradient volume (reference TSDF::updateGradients, TSDF.cpp:120-123).  A no-op in
     * Gradients::OnTheFly mode, where raycast() differences the TSDF directly.
     */
    void updateGradients(Stream& stream = Stream::Null());

    /**
     * Ray-march the volume from cam_pose (reference TSDF::raycast, TSDF.cpp:158-168).  The four
     * outputs must be zeroed by the caller beforehand, as in the reference.
     */
    virtual void raycast(const Affine3f& cam_pose, const Matx33f& intr,
                         const emf_image_t& raylengths, const emf_image_t& vertices,
                         const emf_image_t& normals, const emf_image_t& mask,
                         Stream& stream = Stream::Null(), uint64_t* stats = nullptr);

    /**
     * Un-normalised association likelihood of the camera-frame points with this volume
     * (reference TSDF::computeAssociation + computeLaplace, TSDF.cpp:125-156), one fused kernel.
     */
    void computeAssociation(const emf_image_t& points, const Affine3f& cam_pose,
                            const emf_image_t& associationWeights,
                            Stream& stream = Stream::Null());

    /**
     * Iso-surface of the observed part of the volume (weights > 0), marching cubes on the device
     * (reference TSDF::getMesh, TSDF.cpp:356-373).  Synchronises.
     */
    virtual Mesh getMesh();

    /** Host copies in the reference layout, (Nz*Ny) rows x Nx cols (TSDF.cpp:398-408). */
    std::vector<float> getTSDF() const;
    std::vector<float> getWeightsVol() const;

    const float* tsdfPtr() const { return tsdfVol.as<float>(); }
    const float* weightsPtr() const { return tsdfWeights.as<float>(); }
    const float* gradsPtr() const { return tsdfGrads.empty() ? nullptr : tsdfGrads.as<float>(); }
    uint8_t* brickFlagsPtr() const { return brickFlags.as<uint8_t>(); }
    Gradients gradientMode() const { return gradMode; }

    /**
     * Keep the volume twice (MI355X has the memory: 1 GB more for a 512^3 background) so that a
     * frame's integration can run out of place, concurrently with the same frame's raycast
     * (emf_hip_integrateBatchedCulledOut; EMFusion::integrateBatched).  Everything else keeps reading
     * tsdfPtr() / weightsPtr(), the FRONT copy; after an out-of-place integration into backBuffers()
     * the owner calls flip().  Allocates the second copy (equal to the first) and the two dirty maps.
     */
    void enableDoubleBuffer();
    bool doubleBuffered() const { return !tsdfBack.empty(); }
    /** Second copy + dirty maps in the roles emf_hip_integrateBatchedCulledOut expects now. */
    emf_volume_out_t backBuffers() const;
    /** The back copy holds the newly integrated state: make it the front. */
    void flip();
    /** After an IN-PLACE integration of a double-buffered volume: make the copies equal again. */
    void resyncBack();

    /**
     * Sign maps (emf_model_t.signMaps, include/emf_hip.h): kept by the tile integration launches through
     * the model table, read by emf_hip_raycastFarBounds.  Anything else that writes the tsdf volume
     * calls invalidateSignMaps(); refreshSignMaps() rebuilds them from the values when they are stale
     * (the owner calls it before describe()).  Volumes with Nx % 4 != 0 have none (no tile launches).
     */
    void invalidateSignMaps() { signMapsValid = false; }
    void refreshSignMaps(Stream& stream = Stream::Null());

    /**
     * Static part of this volume's entry in the device model table used by the batched launches
     * (emf_model_t, include/emf_hip.h); image pointers are filled in by the owner of the images.
     */
    virtual void describe(emf_model_t& m) const;
    /** 0 / 1 / 2 from the environment variable EMF_BRICK_FLAGS, see TSDF.cpp. */
    static int brickFlagMode();

    /**
     * The checked reciprocal of the voxel size (emf_hip_voxelReciprocal) is a device-side verdict
     * (3 x 2^23 inputs, ~20 microseconds) the first time a size is seen in the process.  With deferral on,
     * a constructor that meets a new size does not wait for it: the check is enqueued on a stream of
     * its own, the march divides (same results) and pollReciprocal() adopts the verdict once it is in.
     * emf::EMFusion turns this on after its background exists, so objects created inside a frame
     * (reference EMFusion.cpp:495-560) never stall it.
     */
    static void deferReciprocalChecks(bool on);
    /** True if the verdict arrived with this call (the owner refreshes its model table). */
    bool pollReciprocal();
    /** Wait for a deferred check in flight (it ends here); pollReciprocal() then adopts it. */
    void settleReciprocal();
    float reciprocal() const { return rcpVoxel; }

protected:
    Mesh extractMesh(const uint8_t* fgVolMask);
    TSDFParams params;
    Vec3i volumeRes;
    float voxelSize;
    float truncdist;
    float rcpVoxel = 0.f;  // emf_hip_voxelReciprocal(voxelSize): checked stand-in for x / voxelSize
    Affine3f pose;  // volume-centre frame -> world
    Gradients gradMode;
    Size frameSize;

    DeviceBuffer tsdfVol;      // N^3 f32
    DeviceBuffer tsdfWeights;  // N^3 f32
    DeviceBuffer tsdfGrads;    // N^3 x 3 f32, only in Materialized mode
    DeviceBuffer brickFlags;   // ceil(N/8)^3 u8 uniformity flags kept by integrate(), read by raycast()
    // double buffering (enableDoubleBuffer): the other copy, and per 32x8x8 tile "the copies differ"
    DeviceBuffer tsdfBack, weightsBack;
    DeviceBuffer dirtyMaps[2];
    DeviceBuffer signMaps;      // per 32x8x8 tile: holds a positive tsdf / holds a negative tsdf
    bool signMapsValid = false;
    DeviceBuffer relevantTiles; // count + indices of the tiles in which a raycast hit can be completed
    DeviceBuffer unseenTiles;   // per tile: every weight is 0 (emf_model_t.unseenTiles); valid with the sign maps
    int dirtyPrev = 0;  // index of the map the last out-of-place integration wrote

private:
    struct PendingReciprocal;
    struct PendingDeleter {
        void operator()(PendingReciprocal* p) const;
    };
    std::unique_ptr<PendingReciprocal, PendingDeleter> pendingRcp;  // a deferred check in flight
    void obtainReciprocal();
};

}  // namespace emf
