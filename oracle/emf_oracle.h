/*
 * emf_oracle.h -- CPU restatement of EM-Fusion's per-frame volumetric hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The product path (emfusion_amd/, include/) never links,
 * imports or calls it and fails loudly when the HIP library is missing.
 *
 * PARITY UNPINNED.  The reference (EmbodiedVision/emfusion) ships no tests, golden vectors or
 * known-answer fixtures for this path, has no CPU implementation, and cannot
 * be built in this image: its kernels need the CUDA toolkit headers and OpenCV >= 4.3 with the
 * contrib CUDA modules, neither of which is present, and building it against hand-written stand-ins
 * for those headers is not allowed.  Every function below is therefore our own restatement of the
 * reference algorithm, written from reading the cited lines; it is pinned only by analytic
 * known-answer properties (tests/test_oracle_properties.py), not by reference outputs.
 *
 * Conventions shared by all functions:
 *   - volumes are continuous (Nz*Ny) rows x Nx cols float arrays, element (z*Ny + y, x)
 *     (TSDF.cpp:35-42, TSDF.cu:342-343, TSDF.cuh:76-83); `res` = {Nx, Ny, Nz}
 *   - images are continuous row-major H x W, channels interleaved
 *   - R[9] row-major 3x3, t[3]; K[9] row-major intrinsics (cv::Matx33f reinterpretation,
 *     TSDF.cu:417-422)
 *   - all arithmetic is IEEE single precision in the reference's operation order; build with
 *     -ffp-contract=off (see oracle/Makefile) so no a*b+c is fused
 */
#ifndef EMF_ORACLE_H
#define EMF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* number of OpenMP threads used by the sweeps below (1 = scalar port); returns the value in effect */
int orc_set_threads(int n);

/* kernel_computePoints + points.setTo(0): EMFusion.cu:29-61 */
void orc_computePoints(const float* depth, float* points, int w, int h, const float K[9]);

/* kernel_updateTSDF: TSDF.cu:327-401 */
void orc_updateTSDF(const float* depth, const float* assoc, int w, int h, float* tsdf,
                    float* weights, const float R_OC[9], const float t_OC[3], const float K[9],
                    const int res[3], float voxelSize, float truncdist, float maxWeight);

/* TSDF::updateGradients = setTo(0) + kernel_computeTSDFGrads: TSDF.cpp:120-123, TSDF.cu:429-448.
 * grads is N^3 x 3 floats. */
void orc_computeTSDFGrads(const float* tsdf, float* grads, const int res[3]);

/* kernel_raycastTSDF: TSDF.cu:466-573 (+ enterVolStep/exitVolStep TSDF.cuh:31-63,
 * interpolateTrilinear TSDF.cuh:65-97).  Outputs are read-modify-write exactly like the
 * reference: the caller pre-zeroes them (EMFusion.cpp:727-743).  `grads` may be NULL: the
 * gradient at a hit is then blended from forward differences of `tsdf` (identical values, see
 * DESIGN.md).  `fgmask` may be NULL; when given (u8, 0 = background voxel) the weights seen by the
 * march are `fgmask ? w : 0` (ObjTSDF.cpp:209-210).  `steps` (optional, per pixel) receives the
 * number of main-loop iterations that sampled the volume (used for the gather-byte model). */
void orc_raycastTSDF(const float* tsdf, const float* grads, const float* weights,
                     const uint8_t* fgmask, float* raylengths, float* vertices, float* normals,
                     uint8_t* mask, int w, int h, const float R_CO[9], const float t_CO[3],
                     const float K[9], const int res[3], float voxelSize, float truncdist,
                     uint32_t* steps);

/* getVolumeVals = vals.setTo(0) + kernel_getVolumeVals<T>: TSDF.cu:662-726.  channels in 1..3 */
void orc_getVolumeVals(const float* vol, int channels, const float* points, int w, int h,
                       const float R_CO[9], const float t_CO[3], const int res[3],
                       float voxelSize, float* vals);

/* kernel_updateFgBgProbs: ObjTSDF.cu:29-80.  mask/occluded are u8 read as bool; fgbg is N^3 x 2 */
void orc_updateFgBgProbs(const uint8_t* mask, const uint8_t* occluded, int w, int h,
                         const float* tsdf, const float* weights, float* fgbg, const float R[9],
                         const float t[3], const float K[9], const int res[3], float voxelSize);

/* ObjTSDF::computeFgProbs: ObjTSDF.cpp:218-226.  fgProbs = fg/(fg+bg) with x/0 := 0,
 * fgVolMask = fgProbs > 0.5 ? 255 : 0 */
void orc_computeFgProbs(const float* fgbg, float* fgProbs, uint8_t* fgVolMask, const int res[3]);

/* ObjTSDF::raycast weight masking: ObjTSDF.cpp:209-210 (setTo(0) + masked copyTo) */
void orc_maskRaycastWeights(const float* weights, const uint8_t* fgVolMask, float* raycastWeights,
                            const int res[3]);

/* TSDF::computeAssociation / ObjTSDF::computeAssociation incl. computeLaplace:
 * TSDF.cpp:125-156, ObjTSDF.cpp:181-201.  fgProbs == NULL -> background variant.
 * out: un-normalised association weights (W x H). */
void orc_computeAssociation(const float* tsdf, const float* fgProbs, const float* points, int w,
                            int h, const float R_CO[9], const float t_CO[3], const int res[3],
                            float voxelSize, float truncdist, float assocSigma, float alpha,
                            float uniPrior, float* out);

/* EMFusion::computeAssociationWeights normalisation: EMFusion.cpp:653-665.
 * maps[0] = background, maps[1..n-1] = objects in std::map (ID) order.  In place.
 * norm (W x H) receives associationNorm. */
void orc_normalizeAssociation(float* const* maps, int nmaps, int w, int h, float* norm);

/* EMFusion::raycast compositing: EMFusion.cpp:760-794.  Per-object inputs in list order.
 * diff is the persistent diffRaylengths buffer (Q12: untouched where bg_mask == 0).
 * visCounts[k] receives the count of seg == ids[k] inside the boundary-inset rectangle. */
void orc_compositeRaycast(int nobj, const int* ids, const float* const* objRay,
                          const float* const* objVert, const float* const* objNorm,
                          const uint8_t* const* objSeg, const float* bgRay, const float* bgVert,
                          const float* bgNorm, const uint8_t* bgMask, float* ray, float* vert,
                          float* norm, uint8_t* seg, float* diff, uint8_t* noObj, int w, int h,
                          int boundary, int* visCounts);

/* EMFusion::integrateMasks occlusion mask: EMFusion.cpp:897-900.
 * occluded = saturate_u8(objSeg(0/1) - (seg == id ? 255 : 0)) */
void orc_occludedMask(const uint8_t* objSeg, const uint8_t* seg, int id, uint8_t* occluded, int w,
                      int h);

/* ---- f-1: weighted LM-ICP tracking --------------------------------------------------------- */

/* kernel_computePoseGradients (TSDF.cu:603-660).  grads6: (W*H) x 6.  gradsVol may be NULL. */
void orc_computePoseGradients(const float* tsdf, const float* gradsVol, const float* points, int w,
                              int h, const float R_CO[9], const float t_CO[3], const int res[3],
                              float voxelSize, float* grads6);
/* Huber weights, max-normalised integration weights, association (TSDF.cpp:218-252). */
void orc_trackingWeights(const float* tsdfVals, const float* intWeightsRaw, const float* assoc,
                         int n, float huberThresh, float maxWeight, float* trackWeights,
                         float* intWeights);
/* computeAb + weighting + column sums (TSDF.cu:729-766, TSDF.cpp:254-262, 375-388). */
void orc_reduceAb(const float* grads6, const float* tsdfVals, const float* intWeights, int n,
                  float A[36], float b[6]);
/* TSDF::computeError (TSDF.cpp:390-394). */
double orc_trackingError(const float* tsdfVals, const float* intWeights, int n);

/* ---- f-2: EMFusion::preprocessDepth (EMFusion.cpp:294-305): bilateral filter + patches ------ */
void orc_preprocessDepth(const float* raw, int w, int h, int ksz, float sigmaDepth,
                         float sigmaSpatial, float* out);

/* ---- f-4: cuda::TSDF::marchingCubes (TSDF.cu:855-1152) behind TSDF::getMesh / ObjTSDF::getMesh ---
 * mask = weights > 0 [& fg != 0]; vertices / normals: 3 floats per vertex, triangles: 4 ints each. */
void orc_marchingCubesCount(const float* tsdf, const float* weights, const uint8_t* fg,
                            const int res[3], int* numVerts, int* numTris);
void orc_marchingCubes(const float* tsdf, const float* grads, const float* weights,
                       const uint8_t* fg, const int res[3], float voxelSize, float* vertices,
                       float* normals, int* triangles);

/* kernel_renderPhong + the colour lookup of renderGPU (EMFusion.cu:100-186); image: H x W x 3 u8. */
void orc_renderPhong(const float* points, const float* normals, const uint8_t* seg,
                     const uint8_t* colorMap, const float lightPos[3], int w, int h, uint8_t* image);

#ifdef __cplusplus
}
#endif
#endif
