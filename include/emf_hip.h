/*
 * emf_hip.h -- C ABI of the MI355X-native (gfx950) EM-Fusion volumetric hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces either one of
 * the reference's kernel-wrapper free functions (namespace emf::cuda::{TSDF,ObjTSDF,EMFusion},
 * declared in include/EMFusion/core/cuda/{TSDF,ObjTSDF,EMFusion}.cuh) or one chain of
 * OpenCV-CUDA element-wise launches inside emf::TSDF / emf::ObjTSDF / emf::EMFusion methods; the
 * reference interface each one replaces is cited as file:line (relative to the reference root).
 * INTEGRATION.md shows the C++ stub a maintainer adds on the reference side to bind GpuMat
 * arguments to these calls.
 *
 * Contract for every function
 *   - all pointers are DEVICE pointers unless the parameter is named *_host or documented so;
 *     R / t / K / res are small HOST arrays copied into the launch arguments
 *   - never allocates, frees or synchronises: work is enqueued on `stream` (0 = null stream) and
 *     the call returns; launch errors surface as the return value of this or a later call
 *   - returns EMF_OK (0), a negative EMF_E_* for rejected arguments (nothing enqueued), or a
 *     positive hipError_t; never throws; re-entrant, no global state except a thread-local
 *     message buffer read by emf_hip_last_error_string()
 *   - volumes: continuous (Nz*Ny) rows x Nx cols float arrays, element (z*Ny + y, x), i.e. what
 *     cv::cuda::createContinuous(Ny*Nz, Nx, CV_32FCn) allocates (TSDF.cpp:35-42); res = {Nx,Ny,Nz}
 *   - images: emf_image_t = pointer + row pitch in bytes + width/height, the PtrStepSz of a GpuMat
 *   - R row-major 3x3, t 3-vector, K row-major 3x3 intrinsics, exactly the memory of
 *     cv::Matx33f / cv::Vec3f (TSDF.cu:417-422)
 *   - arithmetic is IEEE binary32 in the reference's operation order with a*b+c contraction
 *     disabled; see DESIGN.md "Numerics"
 */
#ifndef EMF_HIP_H
#define EMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMF_HIP_ABI_VERSION 8

/* hipStream_t without dragging HIP headers into C callers */
typedef struct ihipStream_t* emf_stream_t;

enum {
    EMF_OK = 0,
    EMF_E_NULL = -1,      /* a required pointer is NULL */
    EMF_E_SHAPE = -2,     /* non-positive or inconsistent width/height/resolution */
    EMF_E_PITCH = -3,     /* pitch smaller than a row or not a multiple of the element alignment */
    EMF_E_ARG = -4,       /* scalar argument out of domain (voxelSize <= 0, channels not in 1..3 ...) */
    EMF_E_LIMIT = -5,     /* count exceeds a documented limit (EMF_MAX_*) */
    EMF_E_NODEVICE = -6,  /* no HIP device / library built without device code for this GPU */
    EMF_E_NOTREADY = -7,  /* the answer is not known yet (emf_hip_voxelReciprocalCached: size never checked) */
    EMF_E_PEER_TIMEOUT = -8 /* a peer's flag did not arrive within the bound of a direct peer-write exchange (reported by
                             the host classes; the kernels raise the group's error word and skip the exchange's consumer) */
};

/* GpuMat-like image view: `data` device pointer, `pitch` bytes per row (>= width * elemsize) */
typedef struct emf_image {
    void* data;
    size_t pitch;
    int32_t width;
    int32_t height;
} emf_image_t;

#define EMF_MAX_MODELS 256 /* background + objects handled by one call (seg ids are u8) */
#define EMF_MAX_BATCH 32   /* models per batched (model-table) launch */
#define EMF_MAX_PEERS 8    /* ranks of a direct peer-write exchange group (one MI355X node) */

/* Brick uniformity flags: one byte per 4x4x4 brick of a TSDF volume, B = ceil(Nx/4) * ceil(Ny/4) *
 * ceil(Nz/4) bricks, x fastest.  0 = mixed; 1 / 2 / 4 = every voxel of the brick is exactly
 * 0 / +1 / -1.  A flag BUFFER holds 2 * B bytes: the raw flags, then the dilated flags
 * (class | D << 3, D in 1..3 = every brick within Chebyshev distance D shares the class; 0 else).  emf_hip_updateTSDF maintains both;
 * emf_hip_raycastTSDF uses the dilated half to evaluate lookups in uniform regions without
 * gathering and to fast-forward through them (bit-identical results).  A buffer starts as all 1
 * for a zeroed volume (emf_hip_resetBrickFlags) and must be passed to EVERY integration of that
 * volume. */
#define EMF_BRICK 4
#define EMF_BRICK_MIXED 0
#define EMF_BRICK_ALL_ZERO 1
#define EMF_BRICK_ALL_ONE 2
#define EMF_BRICK_ALL_NEG_ONE 4

int emf_hip_abi_version(void);
/* message for the most recent non-zero return on this thread ("" if none) */
const char* emf_hip_last_error_string(void);
/* 0 if a HIP device is usable, EMF_E_NODEVICE otherwise; fills name/arch (may be NULL) */
int emf_hip_device_info(char* name, size_t name_len, char* arch, size_t arch_len, int* num_cus);

/* ------------------------------------------------------------------------------------------------
 * Level 1: one call per reference kernel wrapper (reference memory layout in, same layout out)
 * ---------------------------------------------------------------------------------------------- */

/* Replaces emf::cuda::EMFusion::computePoints (EMFusion.cuh:39-40, EMFusion.cu:29-61).
 * depth: f32 W x H; points: f32x3 W x H, every pixel is written (the reference's setTo(0) is
 * therefore redundant).  Deviation: no cudaDeviceSynchronize -- stream ordered. */
int emf_hip_computePoints(const emf_image_t* depth, const emf_image_t* points, const float K[9],
                          emf_stream_t stream);

/* Replaces emf::cuda::TSDF::updateTSDF (TSDF.cuh:115-122, TSDF.cu:327-427).
 * depth, assocWeights: f32 W x H (same size); tsdf, weights: N^3 f32 read-modify-write.
 * brickFlags: NULL, or the volume's brick uniformity flags, kept consistent by this call.
 * invLambda : NULL, or the table written by emf_hip_computeInvLambda for the same K and image size;
 *             identical results, about a quarter fewer instructions per fused voxel. */
int emf_hip_updateTSDF(const emf_image_t* depth, const emf_image_t* assocWeights, float* tsdf,
                       float* weights, uint8_t* brickFlags, const float R_OC[9],
                       const float t_OC[3], const float K[9], const int32_t res[3],
                       float voxelSize, float truncdist, float maxWeight,
                       const emf_image_t* invLambda, emf_stream_t stream);

/* The pixel-only factor of the integration (TSDF.cu:374-380): invLambda(x, y) =
 * 1 / |((x - cx) / fx, (y - cy) / fy, 1)|, f32 W x H, the very floats updateTSDF computes inline
 * per voxel from the rounded pixel.  Depends on K and the image size only: compute once. */
int emf_hip_computeInvLambda(const float K[9], emf_image_t* invLambda, emf_stream_t stream);

/* Replaces TSDF::updateGradients = tsdfGrads.setTo(0) + emf::cuda::TSDF::computeTSDFGrads
 * (TSDF.cpp:120-123, TSDF.cuh:132-134, TSDF.cu:429-464).  grads: N^3 x 3 f32; the last index
 * planes are written as zero by this call (no separate memset). */
int emf_hip_computeTSDFGrads(const float* tsdf, float* grads, const int32_t res[3],
                             emf_stream_t stream);

/* Replaces emf::cuda::TSDF::raycastTSDF (TSDF.cuh:154-162, TSDF.cu:466-601).
 * raylengths f32, vertices f32x3, normals f32x3, mask u8 (0/1), all W x H, must be PRE-ZEROED by
 * the caller exactly as the reference does (EMFusion.cpp:727-743): pixels without a hit are not
 * written, and a non-zero incoming raylength clips the march (TSDF.cu:496-500).
 *   grads   : N^3 x 3 gradient volume, or NULL -> the normal is blended from forward differences
 *             of `tsdf` on the fly (bit-identical to a volume made by emf_hip_computeTSDFGrads)
 *   fgVolMask: NULL, or N^3 u8 -- the march then sees weights `fgVolMask ? w : 0`, which replaces
 *             ObjTSDF::raycast's per-frame raycastWeights sweep (ObjTSDF.cpp:209-210)
 *   brickFlags: NULL, or the brick uniformity flags of `tsdf` (see EMF_BRICK): lookups whose
 *             eight corners lie in equally-uniform bricks are computed without touching `tsdf`
 *   rcpVoxel: 0, or the value emf_hip_voxelReciprocal returned for `voxelSize`: the march then
 *             divides by the voxel size in 3 instructions instead of 11, same results
 *   stats   : NULL, or 4 x u64 device counters this call ADDS to: [0] volume samples taken by the
 *             main march loop (the S of SURVEY.md section 8d), [1] hits, [2] samples that read
 *             the volume (the rest were answered by the brick flags), [3] samples fast-forwarded */
int emf_hip_raycastTSDF(const float* tsdf, const float* grads, const float* weights,
                        const uint8_t* fgVolMask, const uint8_t* brickFlags,
                        const emf_image_t* raylengths,
                        const emf_image_t* vertices, const emf_image_t* normals,
                        const emf_image_t* mask, const float R_CO[9], const float t_CO[3],
                        const float K[9], const int32_t res[3], float voxelSize, float truncdist,
                        float rcpVoxel, uint64_t* stats, emf_stream_t stream);

/* Plain device-to-device copy kernel (16 B per lane per iteration): the "attainable HBM bandwidth"
 * yardstick bench.py times beside the hot path.  16-byte aligned. */
int emf_hip_streamCopy(void* dst, const void* src, size_t bytes, emf_stream_t stream);

/* Reciprocal of a voxel size, CHECKED for use in place of the division x / voxelSize:
 * runs float inputs x through  q = x * r; q = fma(fma(-q, d, x), r, q)  (r = 1 / d) and through the
 * IEEE division on the device, and stores r in *rcp only if the two agree bit for bit for all x
 * with 1e-30 <= |x| <= 1e30 (0 otherwise; the march keeps its arguments inside that range, see
 * march_wave.hpp).  Swept: every mantissa of the binade [1, 2), both signs, and of the binade
 * that holds 1e-30 -- which decides every binade of the range because both forms commute with
 * scaling by 2^k there (argument and its device-checked premise: abi_common.hip,
 * k_check_reciprocal); ~20 microseconds instead of the 2.3 ms of all 2^32 inputs.
 * The verdict depends on the bit pattern of voxelSize alone and is remembered for the life of
 * the process: the first call for a size runs the check on a stream of its own and waits for THAT
 * (no allocation, no device-wide synchronisation), later calls for the same size return at once
 * without touching the device. */
int emf_hip_voxelReciprocal(float voxelSize, float* rcp);

/* The same check without any wait, for volumes created inside a frame (reference
 * EMFusion.cpp:495-560, initNewObjVolume): the march divides (rcpVoxel = 0, same results) until the
 * verdict is in.
 *   ...Cached: *rcp and EMF_OK if this size has been checked before in this process, else
 *              EMF_E_NOTREADY (nothing is enqueued);
 *   ...Begin : enqueues the check on `stream`; it first stores 0 to *mismatches, then adds the
 *              number of disagreeing inputs to it.  `mismatches` is the caller's (device memory, or
 *              host memory the device can write); nothing is allocated, nothing waited for;
 *   ...End   : the caller has seen the check complete (event, stream query) and read the count:
 *              records the verdict for the process and returns the reciprocal (0 if any input
 *              disagreed).  Pure host code. */
int emf_hip_voxelReciprocalCached(float voxelSize, float* rcp);
int emf_hip_voxelReciprocalBegin(float voxelSize, unsigned long long* mismatches, emf_stream_t stream);
int emf_hip_voxelReciprocalEnd(float voxelSize, unsigned long long mismatches, float* rcp);
/* Test aid: the same comparison over ALL 2^32 bit patterns (2.3 ms of the whole chip, blocking, nothing cached):
 * *mismatches_host = inputs of the guarded range on which the two forms differ.  The short form's verdict must be
 * "usable" exactly when this is 0 (tests/test_gpu_parity.py). */
int emf_hip_voxelReciprocalExhaustive(float voxelSize, unsigned long long* mismatches_host);

/* Measurement aid (bench.py): `iterations` independent 8-byte gather loads per lane from a footprint that stays in every
 * CU's vector L1, `workgroups` x 256 lanes; linesPerInstruction = distinct 128-byte lines one 64-lane instruction touches
 * (64, 4 or 1).  Calibrates the L1 rate the raycast's `roofline.frac` is quoted against.  buf: footprintBytes (power of
 * two) of readable device memory, sink: 8 writable bytes (never written in practice). */
int emf_hip_l1GatherProbe(const void* buf, size_t footprintBytes, int linesPerInstruction, int iterations, int workgroups,
                          void* sink, emf_stream_t stream);

/* ---- direct peer-write exchanges ("Collective implementation"; new design, the
 * reference is single-GPU) ---------------------------------------------------------------------------
 * The exchanges of the sharded path move 1-5 MB: latency decides.  Instead of a library collective
 * each rank stores its contribution straight into a slot of every peer's receive buffer (all xGMI
 * links at once), raises a flag on every peer, waits for the peers' flags and reduces the slots locally
 * in rank order.  emf_peer_t is what ONE rank knows about the group; the host maps the peers' buffers
 * once (same process: the pointers themselves; one process per GPU: hipIpcOpenMemHandle) --
 * emf::makePeerCommunicator.  Buffers: emf_hip_peerBufferBytes(world, slotBytes) bytes (two parities x
 * world slots) and `world` u32 flags per rank, zeroed before the first exchange; slotBytes % 16 == 0.
 * An exchange with sequence number seq (1, 2, 3 ... identical on all ranks) is
 *     scatter (ranks that contribute) -> signal + wait (every rank) -> reduce / copy out of the slots
 * enqueued on the caller's stream; nothing allocates or synchronises.  A wait that exceeds the group's bound
 * stores seq to *error (host-visible word of the caller's) and the exchange's consumer kernel leaves its outputs
 * untouched; the host classes turn a non-zero error word into EMF_E_PEER_TIMEOUT at their next synchronisation.
 * Three forms, same slots and flags:
 *   three launches    peerScatter -> peerSignalWait -> peerReduce* / peerCopyFromSlot            (round 3)
 *   two launches      peerScatter -> peerWaitReduce* / peerWaitCopyFromSlots: the consumer's first workgroup
 *                     signals, every workgroup waits, then reduces / copies                     (round 4)
 *   fused             the path's own kernels scatter and consume: emf_hip_estepBatchedPeer ->
 *                     peerNormalizeAssociation (E-step), emf_hip_packHitKeysPeer ->
 *                     emf_hip_compositeFromKeysPeer (raycast): one extra launch per exchange     (round 4) */
typedef struct emf_peer {
    int32_t rank, world;
    void* slots[EMF_MAX_PEERS];      /* receive buffer of peer p as addressable from THIS device */
    uint32_t* flags[EMF_MAX_PEERS];  /* flag page of peer p: 4096 bytes; words 0..world-1 are the flags */
    size_t slotBytes;                /* capacity of one sender's slot */
    uint32_t* error;                 /* this rank's error word */
    uint32_t timeoutMs;              /* bound of a wait in milliseconds (0: 5000) */
    uint32_t waitInFront;            /* != 0 (what emf::make*PeerCommunicator* set): every peerWait* / *Peer consumer entry
                                        enqueues the one-wave peerSignalWait in front of its kernel, which then does not
                                        poll.  0: the consumer's workgroups signal and poll themselves -- one launch less,
                                        but the polls of a whole grid on the same uncached words cost more than that
                                        launch (k_peer_normalize at 640 x 480, one rank: 5.4 + 5.9 us against 40 us with
                                        1200 polling workgroups, 20 us with 256), and when ranks share a GPU (rehearsals)
                                        grids of spinning workgroups can keep a lagging rank's producer from starting */
    uint32_t systemFences;           /* != 0: every flag store follows a system-scope release and every poll is followed by a
                                        system-scope acquire, in the wait launch and in consumers that wait themselves;
                                        peerScatter ends with one (6.6 us per exchange beside the background's sweep,
                                        measured).  The communicator sets it for ranks on DISTINCT devices; ranks sharing a
                                        device run without (slots and flags are fine-grained memory, peer_core.hpp
                                        "Memory ordering") */
} emf_peer_t;
size_t emf_hip_peerBufferBytes(int world, size_t slotBytes);
int emf_hip_peerScatter(const emf_peer_t* group, const void* src, size_t bytes, size_t dstOffset, uint32_t seq,
                        emf_stream_t stream);
int emf_hip_peerSignalWait(const emf_peer_t* group, uint32_t seq, uint32_t timeoutMs, emf_stream_t stream);
int emf_hip_peerReduceSumF32(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream);
int emf_hip_peerReduceMinU64(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream);
int emf_hip_peerCopyFromSlot(const emf_peer_t* group, uint32_t seq, int sender, size_t srcOffset, void* dst,
                             size_t bytes, emf_stream_t stream);
/* signal + wait + reduce in one launch (out := sum / min over the world slots in rank order) */
int emf_hip_peerWaitReduceSumF32(const emf_peer_t* group, uint32_t seq, size_t count, float* out, emf_stream_t stream);
int emf_hip_peerWaitReduceMinU64(const emf_peer_t* group, uint32_t seq, size_t count, uint64_t* out, emf_stream_t stream);
/* signal + wait + nparts copies dst[k][0, bytes[k]) := slot[senders[k]] + srcOffsets[k] in one launch (a broadcast:
 * one part on the receivers, none on the root; a band gather: one part per peer and image); nparts <= 16 */
int emf_hip_peerWaitCopyFromSlots(const emf_peer_t* group, uint32_t seq, int nparts, const int32_t* senders_host,
                                  const size_t* srcOffsets_host, void* const* dsts_host, const size_t* bytes_host,
                                  emf_stream_t stream);
/* Diagnostics behind the two arithmetic shortcuts of the tiled integration (device_core.hpp, on by
 * default, EMF_INT_FAST): the pixel of a voxel as round(x * rcp(z)) unless that lies next to a rounding
 * tie, and the side of the truncation band from the hardware square root unless the distance lies
 * next to +-truncdist.  Both rest on "v_rcp_f32 / v_sqrt_f32 are accurate to 1 ulp".
 *   ...sweepFastPathPremises: all 2^32 float bit patterns through both instructions against the
 *      correctly rounded double results; out4 (device, 4 x u64, zeroed by the call): [0] inputs with
 *      2^-126 <= |z| <= 2^126 whose reciprocal is off by more than 2^-23 relative, [1] inputs
 *      n >= 2^-126 whose square root is, [2] / [3] the largest relative errors seen (float bits);
 *   ...debugPixelRounding: fast[i] / exact[i] = the rounded quotient num[i] / den[i] by the shortcut
 *      and by the IEEE division -- the very device functions the kernels call;
 *   ...debugBandDecision: likewise for the branch (bits 0..3: 2 = behind the band, 3 = fuse; bit 4:
 *      inside the band) and the clamped sample of a voxel with depth d, pixel factor invLambda and
 *      squared distance n2. */
int emf_hip_sweepFastPathPremises(unsigned long long* out4_dev, emf_stream_t stream);
int emf_hip_debugPixelRounding(const float* num_dev, const float* den_dev, int n, int32_t* fast_dev,
                               int32_t* exact_dev, emf_stream_t stream);
int emf_hip_debugBandDecision(const float* d_dev, const float* invLambda_dev, const float* n2_dev, int n,
                              float truncdist, int32_t* kindFast_dev, float* sampleFast_dev,
                              int32_t* kindExact_dev, float* sampleExact_dev, emf_stream_t stream);

/* Diagnostic: one wave that stays resident on `stream` until *release != 0 (host memory the device
 * can read) or maxMilliseconds have passed.  While it runs the stream is "not ready", so a host call
 * that synchronised with the whole device cannot have returned before it ended: the tests use it
 * to show that a frame contains no device-wide synchronisation. */
int emf_hip_spinProbe(const volatile uint32_t* release, uint32_t maxMilliseconds, emf_stream_t stream);
/* Diagnostic: keeps `stream` busy for `microseconds` (one sleeping wave): the latency of a small
 * collective in single-GPU measurements of the exchange path (emf::makeDelayedCommunicator). */
int emf_hip_spinDelay(uint32_t microseconds, emf_stream_t stream);

/* Replaces emf::cuda::TSDF::getVolumeVals (TSDF.cuh:197-203, TSDF.cu:662-726).
 * vol: N^3 x channels f32 (channels 1..3, interleaved); points f32x3; vals f32 x channels.
 * The callee zero-fills vals for pixels without a lookup (vals.setTo(0), TSDF.cu:705). */
int emf_hip_getVolumeVals(const float* vol, int channels, const emf_image_t* points,
                          const float R_CO[9], const float t_CO[3], const int32_t res[3],
                          float voxelSize, const emf_image_t* vals, emf_stream_t stream);

/* Replaces emf::cuda::ObjTSDF::updateFgBgProbs (ObjTSDF.cuh:49-56, ObjTSDF.cu:29-107).
 * mask, occluded: u8 W x H read as bool; fgBgProbs: N^3 x 2 f32 read-modify-write. */
int emf_hip_updateFgBgProbs(const emf_image_t* mask, const emf_image_t* occluded,
                            const float* tsdf, const float* weights, float* fgBgProbs,
                            const float R_OC[9], const float t_OC[3], const float K[9],
                            const int32_t res[3], float voxelSize, emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Level 2: one call per OpenCV-CUDA launch chain in the volume / orchestrator classes
 * ---------------------------------------------------------------------------------------------- */

/* Replaces ObjTSDF::computeFgProbs (ObjTSDF.cpp:218-226): split, add, divide (x/0 := 0),
 * compare(NE)+setTo(0), compare(GT 0.5).  fgProbs N^3 f32, fgVolMask N^3 u8 (0/255). */
int emf_hip_computeFgProbs(const float* fgBgProbs, float* fgProbs, uint8_t* fgVolMask,
                           const int32_t res[3], emf_stream_t stream);

/* Literal replacement of ObjTSDF::raycast's weight masking (ObjTSDF.cpp:209-210):
 * raycastWeights = fgVolMask ? weights : 0.  The native path passes fgVolMask to
 * emf_hip_raycastTSDF instead and never materialises this volume. */
int emf_hip_maskRaycastWeights(const float* weights, const uint8_t* fgVolMask,
                               float* raycastWeights, const int32_t res[3], emf_stream_t stream);

/* Replaces TSDF::computeAssociation (TSDF.cpp:125-136) incl. TSDF::computeLaplace
 * (TSDF.cpp:138-156) when fgProbs == NULL, and ObjTSDF::computeAssociation (ObjTSDF.cpp:181-201)
 * otherwise: trilinear SDF lookup, Laplace likelihood, optional foreground-probability factor,
 * mixture with the uniform prior, zero where the lookup is exactly 0.  out: f32 W x H,
 * UN-normalised, every pixel written. */
int emf_hip_computeAssociation(const float* tsdf, const float* fgProbs, const emf_image_t* points,
                               const float R_CO[9], const float t_CO[3], const int32_t res[3],
                               float voxelSize, float truncdist, float assocSigma, float alpha,
                               float uniPrior, const emf_image_t* out, emf_stream_t stream);

/* Replaces the normalisation half of EMFusion::computeAssociationWeights (EMFusion.cpp:653-665).
 * maps_host: HOST array of `nmaps` image views (device data), [0] = background then objects in
 * std::map (ascending ID) order; every map is divided in place by the normaliser; x/0 := 0.
 *   nsum    : the normaliser is the sequential sum of maps[0 .. nsum-1] (single GPU: nsum = nmaps)
 *   extraSum: NULL, or f32 W x H added LAST into the normaliser.  Multi-GPU: every rank
 *             passes nsum = 1 (its background replica) and extraSum = the all-reduced sum of all
 *             ranks' object maps (emf_hip_sumAssociation + RCCL all-reduce)
 *   norm    : NULL, or f32 W x H receiving associationNorm (required when nsum > 16)
 * 1 <= nmaps <= EMF_MAX_MODELS, 0 <= nsum <= nmaps, nsum == 0 requires extraSum. */
int emf_hip_normalizeAssociation(const emf_image_t* maps_host, int nmaps, int nsum,
                                 const emf_image_t* extraSum, const emf_image_t* norm,
                                 emf_stream_t stream);

/* Sum of `nmaps` association maps into `sum` (sequential order, maps_host[0] first): the local
 * partial a rank contributes to the normaliser all-reduce.  sum: f32 W x H, overwritten. */
int emf_hip_sumAssociation(const emf_image_t* maps_host, int nmaps, const emf_image_t* sum,
                           emf_stream_t stream);

/* Replaces the compositing part of EMFusion::raycast (EMFusion.cpp:760-794) in one pass.
 * Per-object inputs are HOST arrays of `nobj` image views in std::list (creation) order; ids_host
 * holds the object IDs written to the segmentation (saturated to u8 like cv::Scalar->uchar).
 *   diff     : f32 W x H, the persistent diffRaylengths buffer, only updated where bgMask != 0
 *   visCounts: device int32[nobj], overwritten with the number of pixels with seg == id inside
 *              [boundary, W-boundary) x [boundary, H-boundary)  (EMFusion.cpp:778-791)
 * Outputs ray/vert/norm/seg/noObj are fully overwritten (no pre-zeroing needed). */
int emf_hip_compositeRaycast(int nobj, const int32_t* ids_host, const emf_image_t* objRay_host,
                             const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                             const emf_image_t* objSeg_host, const emf_image_t* bgRay,
                             const emf_image_t* bgVert, const emf_image_t* bgNorm,
                             const emf_image_t* bgMask, const emf_image_t* ray,
                             const emf_image_t* vert, const emf_image_t* norm,
                             const emf_image_t* seg, const emf_image_t* diff,
                             const emf_image_t* noObj, int boundary, int32_t* visCounts,
                             emf_stream_t stream);

/* emf_hip_compositeRaycast + emf_hip_visibilityFlags in two launches instead of three: the launch that
 * writes the composite also counts (its last chunk, on the segmentation values it writes), the flag launch
 * clears the counts behind itself.  visCounts is scratch of the pair here: ZERO on entry, zero on return
 * (the numbers go to countsMirror, if given, and into the flags). */
int emf_hip_compositeVisibility(int nobj, const int32_t* ids_host, const emf_image_t* objRay_host,
                                const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                                const emf_image_t* objSeg_host, const emf_image_t* bgRay,
                                const emf_image_t* bgVert, const emf_image_t* bgNorm,
                                const emf_image_t* bgMask, const emf_image_t* ray,
                                const emf_image_t* vert, const emf_image_t* norm,
                                const emf_image_t* seg, const emf_image_t* diff,
                                const emf_image_t* noObj, int boundary, int32_t* visCounts,
                                int visibilityThresh, int32_t* visible_dev, int32_t* countsMirror,
                                emf_stream_t stream);

/* Replaces the occlusion mask of EMFusion::integrateMasks (EMFusion.cpp:897-900):
 * occluded = saturate_u8(objSeg - (seg == id ? 255 : 0)).  All u8 W x H. */
int emf_hip_occludedMask(const emf_image_t* objSeg, const emf_image_t* seg, int id,
                         const emf_image_t* occluded, emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Level 3: batched, model-table driven launches used by the host classes (emf::EMFusion).
 * One launch covers the background and every object volume of this rank, replacing the
 * reference's one-stream-per-volume fan-out (EMFusion.h:471, EMFusion.cpp:636-668, 727-758,
 * 866-888).  Results are identical to calling the level-1/2 functions model by model.
 * ---------------------------------------------------------------------------------------------- */

/* Static description of one model, an array of which lives in DEVICE memory (models_dev).
 * Slot 0 is the background, slots 1.. the objects in creation order.  Image pointers are
 * continuous W x H buffers (pitch = width * elemsize). */
typedef struct emf_model {
    float* tsdf;            /* N^3 f32 */
    float* weights;         /* N^3 f32 */
    const float* grads;     /* N^3 x 3 f32 or NULL (on-the-fly differences) */
    const float* fgProbs;   /* N^3 f32, objects only, else NULL */
    const uint8_t* fgVolMask; /* N^3 u8, objects only, else NULL */
    uint8_t* brickFlags;    /* flag buffer (2 * B bytes, see EMF_BRICK) or NULL */
    float* assoc;           /* f32 association map of this model */
    float* raylengths;      /* f32   raycast outputs of this model */
    float* vertices;        /* f32x3 */
    float* normals;         /* f32x3 */
    uint8_t* hitMask;       /* u8 0/1 */
    uint8_t* signMaps;      /* emf_hip_signMapBytes(res) bytes or NULL: per 32x8x8 tile "holds a positive
                             * tsdf", then per tile "holds a negative tsdf"; kept by the tile integration
                             * launches (sticky), read by emf_hip_raycastFarBounds */
    uint32_t* relevantTiles; /* emf_hip_relevantTileBytes(res) bytes or NULL (emf_hip_updateRelevantTiles) */
    uint8_t* unseenTiles;   /* emf_hip_unseenTileBytes(res) bytes or NULL: per 32x8x8 tile "every weight is 0 (and
                             * every tsdf finite)"; set by the owner (a cleared volume: all 1;
                             * emf_hip_rebuildUnseenTiles), cleared by the tile integration launches, which
                             * integrate such a tile without reading it */
    int32_t res[3];
    int32_t id;             /* 0 = background */
    float voxelSize, truncdist, maxWeight;
    float assocC1;          /* -truncdist / assocSigma  (TSDF.cpp:151) */
    float assocC2;          /* 1 / (2 assocSigma)       (TSDF.cpp:154) */
    float alpha;            /*                          (TSDF.cpp:131) */
    float assocC3;          /* (1 - alpha) * uniPrior   (TSDF.cpp:133) */
    int32_t reserved;
    float rcpVoxel;         /* emf_hip_voxelReciprocal(voxelSize), or 0 = divide */
    int32_t pad_;
} emf_model_t;

/* Rigid transform passed by value with each launch (poses change every frame). */
typedef struct emf_pose {
    float R[9];
    float t[3];
} emf_pose_t;

/* E-step for all models in one launch (TSDF.cpp:125-156, ObjTSDF.cpp:181-201, EMFusion.cpp:635-670):
 * each (pixel, model) lane computes its likelihood, partial sums meet in LDS.
 *   poseCO_host[m]: camera -> volume m.  points: f32x3 W x H.
 *   normalize != 0: maps are written normalised, norm (f32 W x H, may be NULL) gets the
 *                   normaliser = sequential sum of all maps (+ nothing else): single-GPU form
 *   normalize == 0: maps are written UN-normalised and objSum (f32 W x H, or NULL: no sum) receives
 *                   the sequential sum of the object maps (slots 1..): the partial a rank feeds to
 *                   the all-reduce; finish with emf_hip_normalizeAssociation(nsum = 1, extraSum)
 * 1 <= nmodels <= EMF_MAX_BATCH.  A model list longer than that is served in chunks of the table
 * (models_dev + k, poseCO_host + k; normalize == 0, objSum NULL) followed by ONE
 * emf_hip_normalizeAssociation over all maps (nsum = nmaps): the same sequential sum, the same bits. */
int emf_hip_estepBatched(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                         const emf_image_t* points, int normalize, const emf_image_t* norm,
                         const emf_image_t* objSum, emf_stream_t stream);

/* emf_hip_normalizeAssociation(nsum = nmaps) over the `assoc` maps of a whole model table (continuous W x H, slot 0 first)
 * in ONE launch: the sequential sum of all maps in table order, every map divided by it (x / 0 := 0), the sum to
 * norm_dev (continuous W x H f32, or NULL).  Finishes the chunked E-step of a model list longer than EMF_MAX_BATCH
 * (emf_hip_estepBatched per chunk with normalize == 0).  1 <= nmodels <= EMF_MAX_MODELS.  Same bits. */
int emf_hip_normalizeAssociationTable(const emf_model_t* models_dev, int nmodels, int width, int height, float* norm_dev,
                                      emf_stream_t stream);

/* emf_hip_computePoints + emf_hip_estepBatched in one launch, for the first E-step of a frame
 * (EMFusion.cpp:73, 79): each pixel's point is formed from `depth` with computePoints' arithmetic,
 * used, and stored to `points` (f32x3 W x H, every pixel written) for the frame's later stages. */
int emf_hip_estepBatchedFromDepth(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                                  const emf_image_t* depth, const float K[9], const emf_image_t* points,
                                  int normalize, const emf_image_t* norm, const emf_image_t* objSum,
                                  emf_stream_t stream);

/* Raycast of all models in one launch (TSDF.cu:466-601 per model, ObjTSDF.cpp:203-216).
 * Unlike emf_hip_raycastTSDF the outputs need no pre-zeroing: every pixel of every model's
 * raylengths / vertices / normals / hitMask is written (zeros where there is no hit), which is
 * what the reference's setTo(0) + kernel leave behind (EMFusion.cpp:727-758).
 *   res_host: HOST int32[nmodels * 3], the resolutions stored in the table
 *   useBrickFlags != 0: march with the models' brick flags (fast-forward through uniform bricks);
 *   0: ignore them and use the wave-scheduled march (default; faster on the bench scene).
 * Both produce the same images.  Models whose table entry carries rcpVoxel != 0 divide by the
 * voxel size with the checked reciprocal (emf_hip_voxelReciprocal), same results.
 *   bgBandRow0, bgBandRows: multi-GPU split of the REPLICATED background (table slot 0): only the
 *   image rows [bgBandRow0, bgBandRow0 + bgBandRows) of slot 0 are marched and written, the others
 *   are left untouched for an all-gather of the ranks' bands (multiples of 16; 0, 0 = all rows). */
int emf_hip_raycastBatched(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                           const int32_t* res_host, int nmodels, int width, int height,
                           const float K[9], int useBrickFlags, int bgBandRow0, int bgBandRows,
                           const float* farBounds_dev, const float* voxelSizes_host, uint64_t* stats,
                           emf_stream_t stream);
/* emf_hip_raycastBatched with the BACKGROUND's rays marched by lanesPerBgRay = 1, 2 or 4 lanes each (march_quad,
 * march_wave.hpp: the lanes of a ray take consecutive samples speculatively; same images, same sample count; measured
 * slower beside the background's integration, hence not the default).  Ignored (1) with brick flags or volumes above
 * 4 GiB. */
int emf_hip_raycastBatchedLanes(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                                const int32_t* res_host, int nmodels, int width, int height,
                                const float K[9], int useBrickFlags, int bgBandRow0, int bgBandRows,
                                const float* farBounds_dev, const float* voxelSizes_host, int lanesPerBgRay,
                                uint64_t* stats, emf_stream_t stream);
/* The same launch for a table chunk that holds OBJECTS ONLY (every slot, slot 0 included, is marched over its
 * footprint and zero-filled outside it): the later chunks of a model list longer than EMF_MAX_BATCH
 * (reference EMFusion.cpp:745-758 loops over any number of objects).  farBounds_dev: this chunk's part of the
 * bounds (emf_hip_raycastFarBounds called with the same chunk). */
int emf_hip_raycastBatchedObjects(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                                  const int32_t* res_host, int nmodels, int width, int height,
                                  const float K[9], int useBrickFlags, const float* farBounds_dev,
                                  const float* voxelSizes_host, uint64_t* stats, emf_stream_t stream);
/* voxelSizes_host: NULL, or HOST float[nmodels], the voxel sizes stored in the table.  With them the
 * objects (slots 1..) get marching workgroups only for the 16x16-pixel tiles their volume box can project
 * to under poseCO_host; the rest of their images is zero-filled sixteen tiles per workgroup -- same
 * output, but four objects no longer put 4 x 1200 nearly empty workgroups in front of the background's
 * (whose dispatch alone took the first 200 us of the launch). */

/* Far bounds for emf_hip_raycastBatched (farBounds_dev; NULL = none).  A hit of the reference's march
 * (TSDF.cu:533-568) is a negative sample following a positive one, so it needs a negative and a positive
 * voxel close together.  From the models' sign maps (emf_model_t.signMaps) this call computes, per model
 * and per 8x8-pixel cell of the image, the largest raylength at which a ray of the cell can still
 * complete a hit (0: it cannot at all; +inf for models without sign maps); the march of a ray is cut
 * there.  Every sample still taken is taken exactly as before, the ones dropped could not have written
 * an output: results are bit-identical, the rays that used to run on through unseen space to the far
 * side of the volume -- the longest of the image -- stop behind the last surface.
 *   bounds_dev: emf_hip_raycastFarBoundBytes(nmodels, width, height) bytes
 * emf_hip_rebuildSignMaps recomputes a volume's maps from its values (needed after anything but the
 * tile integration launches wrote the tsdf: uploads, emf_hip_copyValues, the one-voxel-per-lane
 * kernels, emf_hip_updateTSDF). */
size_t emf_hip_signMapBytes(const int32_t res[3]);
int emf_hip_rebuildSignMaps(const float* tsdf, const int32_t res[3], uint8_t* signMaps, emf_stream_t stream);
/* Unseen-tile map (emf_model_t.unseenTiles).  A voxel nobody has fused into (weight 0) is set to 0, to
 * -1 or to its first sample by kernel_updateTSDF whatever it held (TSDF.cu:352-355, 369-372, 392-400), or
 * left alone (pixel outside the image, association weight 0): for a tile of such voxels the tile
 * integration launches compute the new values without loading the old ones (they load what a voxel
 * keeps) and store them -- behind surfaces and outside the truncation band, where depth drop-outs flip
 * unseen voxels between -1 and 0 from frame to frame, that is two thirds of the bytes the sweep moved. */
size_t emf_hip_unseenTileBytes(const int32_t res[3]);
int emf_hip_rebuildUnseenTiles(const float* tsdf, const float* weights, const int32_t res[3], uint8_t* unseenTiles,
                               emf_stream_t stream);
size_t emf_hip_raycastFarBoundBytes(int nmodels, int width, int height);
/* scanMask: bit m set = model m has every tile of its sign maps examined (a neighbourhood scan per tile:
 * fine for an object volume, ~50 us for a 512^3 one); clear = its relevant-tile list is walked
 * (emf_model_t.relevantTiles; a model with neither bit nor list keeps its whole range). */
int emf_hip_raycastFarBounds(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                             const int32_t* res_host, int nmodels, int width, int height, const float K[9],
                             uint32_t scanMask, float* bounds_dev, emf_stream_t stream);
/* The tiles in which a hit can be completed, as a list per model (emf_model_t.relevantTiles:
 * emf_hip_relevantTileBytes(res) bytes: a count, then tile indices): rebuilt from the sign maps after an
 * integration -- off the frame's critical path -- so that emf_hip_raycastFarBounds, which needs the
 * camera pose of the frame and therefore sits right in front of the raycast, only has to project a
 * few thousand tiles.  Models whose relevantTiles is NULL are skipped. */
size_t emf_hip_relevantTileBytes(const int32_t res[3]);
int emf_hip_updateRelevantTiles(const emf_model_t* models_dev, const int32_t* res_host, int nmodels,
                                emf_stream_t stream);

/* Integration of all models in one launch (TSDF.cu:327-427 per model, EMFusion.cpp:865-875).
 *   poseOC_host[m]: volume m -> camera
 *   res_host    : HOST int32[nmodels * 3], the resolutions stored in the table (the launch grid is
 *                 sized from them).  Models with Nx % 4 == 0 run on 32x8x8 float4 tiles; others (an
 *                 object after ObjTSDF::resize is only guaranteed an even Nx) run one voxel per
 *                 lane in a second launch of the same call -- same arithmetic, same gate
 *   visible_dev : NULL, or device int32[nmodels]; models with visible_dev[m] == 0 are skipped
 *                 (EMFusion.cpp:869-872) -- evaluated on the device, no host round trip
 * Each model's `assoc` map weights its fusion; brickFlags are kept consistent when present.
 *   invLambda   : NULL, or emf_hip_computeInvLambda's table for K and the depth size
 *   maintainBrickFlags: 0 if no model of the table carries brick flags (skips the launch that
 *                 refreshes the dilated flags); non-zero otherwise
 *   stats       : NULL, or one u64 device counter this call ADDS the voxel count of every model
 *                 it actually sweeps to (work accounting for the byte model)
 */
int emf_hip_integrateBatched(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                             const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                             const emf_image_t* depth, const emf_image_t* invLambda,
                             const float K[9], int maintainBrickFlags, uint64_t* stats,
                             emf_stream_t stream);

/* The same integration with a two-level launch (models with Nx % 4 == 0 only, no brick flags upkeep):
 * boxes of 1x2x2 tiles (32 x 16 x 16 voxels) that lie outside the view cone are culled first (one lane per box), and only
 * the tiles of the surviving boxes get a workgroup -- the culled tiles of a large volume otherwise
 * cost a workgroup dispatch each.  Results are identical to emf_hip_integrateBatched.
 *   scratch_dev      : emf_hip_integrateCullScratchBytes(res_host, nmodels) bytes
 *   launchBoxes      : how many boxes the tile launch is sized for -- the survivor count of an earlier
 *                      frame plus a margin; 0 = all boxes.  Too few: the grid strides over the rest;
 *                      too many: the extra workgroups exit.
 *   survivors_out_dev: NULL, or where this call's survivor count (u32) is copied at the end */
size_t emf_hip_integrateCullScratchBytes(const int32_t* res_host, int nmodels);
int emf_hip_integrateBatchedCulled(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                                   const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                                   const emf_image_t* depth, const emf_image_t* invLambda,
                                   const float K[9], void* scratch_dev, uint32_t launchBoxes,
                                   uint32_t* survivors_out_dev, uint64_t* stats, emf_stream_t stream);

/* Out-of-place form of the same launch, for volumes that are kept TWICE (double-buffered) so that the
 * integration of a frame can run concurrently with the raycast of the same frame -- both read the state
 * the previous frame left; the reference runs them back to back (EMFusion.cpp:94, 103) -- and the
 * raycast never sees a half-integrated volume.  Model m is READ from models_dev[m].tsdf / .weights
 * and the integrated state is written to out_host[m].tsdf / .weights, the volume's second copy.  The
 * two copies are equal wherever the previous integration changed nothing, which the dirty maps track
 * per array at tile granularity (one byte per 32 x 8 x 8 voxel tile for the tsdf, then one per tile for
 * the weights: emf_hip_integrateDirtyMapBytes(res) bytes in all).  A set byte of dirtyPrev -- the
 * copies of that array differ in that tile -- makes this call write every voxel of the array in the
 * tile (integrated or copied), elsewhere it writes only what changes; dirtyNext (cleared by the call)
 * receives what this call changed and is the next call's dirtyPrev, with the roles of the copies
 * swapped.  A model whose visible_dev gate is closed is not integrated but still brought up to date.
 * Start with equal copies and clean maps.  Values are those of the in-place launch, bit for bit.
 * out_host == NULL: in place (= emf_hip_integrateBatchedCulled); otherwise every model of the call
 * needs its four pointers. */
typedef struct emf_volume_out {
    float* tsdf;
    float* weights;
    const uint8_t* dirtyPrev;
    uint8_t* dirtyNext;
} emf_volume_out_t;
size_t emf_hip_integrateDirtyMapBytes(const int32_t res[3]);
int emf_hip_integrateBatchedCulledOut(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                                      const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                                      const emf_image_t* depth, const emf_image_t* invLambda,
                                      const float K[9], const emf_volume_out_t* out_host, int prepared,
                                      void* scratch_dev, uint32_t launchBoxes, uint32_t* survivors_out_dev,
                                      uint64_t* stats, emf_stream_t stream);
/* prepared != 0: the caller has cleared the survivor counter (first word of scratch_dev) and the
 * dirtyNext maps already -- emf_hip_integratePrepareOut does exactly that and can be enqueued as soon as
 * the previous call on that scratch / those maps has run, i.e. off the frame's critical path (two fill
 * commands in front of the launch cost it ~40 us while the raycast's workgroups are being dispatched).
 * out_host may be NULL here (counter only). */
int emf_hip_integratePrepareOut(const emf_volume_out_t* out_host, const int32_t* res_host, int nmodels,
                                void* scratch_dev, emf_stream_t stream);

/* visible_dev[slot] = (slot == 0) ? 1 : (visCounts[slot - 1] > visibilityThresh)  for
 * slot < nmodels (EMFusion.cpp:778-791): turns compositeRaycast's counts into the gate above.
 * countsMirror: NULL, or nmodels - 1 int32 in device-visible HOST memory (hipHostMalloc) that receive
 * the counts as well -- the host's view of the visible set without a copy command in the stream. */
int emf_hip_visibilityFlags(const int32_t* visCounts, int nmodels, int visibilityThresh,
                            int32_t* visible_dev, int32_t* countsMirror, emf_stream_t stream);

/* Fill a brick flag buffer (2 * B bytes) for a freshly zeroed volume (all EMF_BRICK_ALL_ZERO). */
int emf_hip_resetBrickFlags(uint8_t* brickFlags, const int32_t res[3], emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Cross-GPU compositing (object volumes sharded over ranks).  The reference
 * is single-GPU; these three calls bracket the ONE all-reduce(min, u64, W*H) that merges the
 * nearest raycast hit over the objects of all ranks with the reference's tie rule (first object in
 * creation order keeps the pixel, EMFusion.cpp:760-771):
 *   key = (float_bits(raylength) << 32) | listPosition,   no hit = all ones.
 * ---------------------------------------------------------------------------------------------- */

/* keys[pixel] = min key over this rank's objects.  listPos_host[k] = position of local object k in
 * the global creation-order list; objRay / objSeg: its raycast raylengths (f32) and hit mask (u8).
 * 0 <= nlocal <= EMF_MAX_BATCH; keys: u64 W x H, overwritten. */
int emf_hip_packHitKeys(int nlocal, const int32_t* listPos_host, const emf_image_t* objRay_host,
                        const emf_image_t* objSeg_host, uint64_t* keys, int width, int height,
                        emf_stream_t stream);

/* Finish the composite from all-reduced keys: segmentation id = ids_host[listPosition], raylength
 * from the key, vertex / normal from the winner's images when it lives on this rank (zeros
 * otherwise: they only feed rendering), then the background override, noObj mask, background
 * vertices / normals and per-object visibility counts exactly as emf_hip_compositeRaycast
 * (EMFusion.cpp:773-794).  ids_host: ids of ALL nall objects in creation order; visCounts: device
 * int32[nall], overwritten. */
int emf_hip_compositeFromKeys(const uint64_t* keys, int nall, const int32_t* ids_host, int nlocal,
                              const int32_t* listPos_host, const emf_image_t* objRay_host,
                              const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                              const emf_image_t* bgRay, const emf_image_t* bgVert,
                              const emf_image_t* bgNorm, const emf_image_t* bgMask,
                              const emf_image_t* ray, const emf_image_t* vert,
                              const emf_image_t* norm, const emf_image_t* seg,
                              const emf_image_t* diff, const emf_image_t* noObj, int boundary,
                              int32_t* visCounts, emf_stream_t stream);

/* visible_dev[slot] = (slot == 0) ? 1 : (visCounts[countIndex_host[slot]] > visibilityThresh):
 * emf_hip_visibilityFlags for a rank whose model slots map to arbitrary entries of a global count
 * array.  1 <= nmodels <= EMF_MAX_BATCH + 1 (the background + the objects one rank may own). */
int emf_hip_visibilityFlagsIndexed(const int32_t* visCounts, int nmodels,
                                   const int32_t* countIndex_host, int visibilityThresh,
                                   int32_t* visible_dev, emf_stream_t stream);

/* Depth pre-processing: replaces EMFusion::preprocessDepth
 * (EMFusion.cpp:294-305) = cv::cuda::bilateralFilter(kernelSize, sigmaDepth [m], sigmaSpatial [px],
 * reflected borders) + NaN -> 0 + "0 wherever the raw depth is 0", one launch.  f32 W x H images,
 * not in place; kernelSize odd, <= 15. */
int emf_hip_preprocessDepth(const emf_image_t* depthRaw, const emf_image_t* depth, int kernelSize,
                            float sigmaDepth, float sigmaSpatial, emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Object creation / matching from instance masks (mesh-free part)
 * ---------------------------------------------------------------------------------------------- */

/* What EMFusion::initNewObjVolume needs of the masked points (EMFusion.cpp:498-535):
 * count = pixels with mask != 0 and a valid point (computeValidPoints: any coordinate != 0);
 * p10 / p90 = per axis, the elements at index int(count * .1f) and int(count * .9f) of the sorted
 * coordinates of those points after x' = R x + t (computePercentiles, EMFusion.cu:77-98). */
typedef struct emf_point_stats {
    uint32_t count;
    float p10[3];
    float p90[3];
} emf_point_stats_t;

size_t emf_hip_pointStatsScratchBytes(void);

/* filterPoints + transformPoints + computePercentiles (EMFusion.cu:63-98, EMFusion.cpp:408-415)
 * fused into a masked radix select: no compaction, no sort, bit-identical order statistics.
 * points f32x3, mask u8 (W x H); stats_dev and scratch_dev in device memory; 9 small launches. */
int emf_hip_maskedPointStats(const emf_image_t* points, const emf_image_t* mask, const float R[9],
                             const float t[3], void* scratch_dev, emf_point_stats_t* stats_dev,
                             emf_stream_t stream);

/* The statistics of EMFusion::updateObj (EMFusion.cpp:827-863): as emf_hip_maskedPointStats, but
 * over the masked points (already transformed into the object's frame by R, t) TOGETHER WITH the
 * vertex cloud of the object's marching-cubes mesh (TSDF.cu:855-1152, ObjTSDF.cpp:247-268): one
 * vertex per sign-changing edge of every cube whose 8 voxels have weight > 0 (and fgVolMask != 0 if
 * given), interpolated by vertexInterp.  The cloud does not depend on the triangle table, so no
 * mesh is built: the vertices are streamed into the same radix select. */
int emf_hip_objectExtentStats(const emf_image_t* points, const emf_image_t* mask, const float R[9],
                              const float t[3], const float* tsdf, const float* weights,
                              const uint8_t* fgVolMask, const int32_t res[3], float voxelSize,
                              void* scratch_dev, emf_point_stats_t* stats_dev, emf_stream_t stream);

/* Replaces emf::cuda::TSDF::copyValues (TSDF.cu:768-819) used by ObjTSDF::resize: dst (dstRes,
 * `channels` floats per voxel) receives src shifted by `offset` voxels -- dst(v) = src(v + offset)
 * inside the source, 0 elsewhere (the reference clears dst first; this call writes every voxel). */
int emf_hip_copyValues(const float* src, float* dst, int channels, const int32_t offset[3],
                       const int32_t srcRes[3], const int32_t dstRes[3], emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Meshes: cuda::TSDF::marchingCubes (TSDF.cu:855-1152) behind
 * TSDF::getMesh / ObjTSDF::getMesh (TSDF.cpp:356-373, ObjTSDF.cpp:247-268)
 * ---------------------------------------------------------------------------------------------- */

typedef struct emf_mesh_counts {
    uint32_t vertices;  /* = mesh.cloud.cols */
    uint32_t triangles; /* = mesh.polygons.cols / 4 */
} emf_mesh_counts_t;

/* Bytes of device scratch the two calls below share for a volume of this resolution (about
 * 36 bytes per 1000 voxels; the reference keeps 9 bytes per cube in cubeClasses / vertIdxBuffer / triIdxBuffer). */
size_t emf_hip_meshScratchBytes(const int32_t res[3]);

/* Pass 1 -- kernel_classifyCubes + the two sums + the two exclusive scans: counts the vertices and
 * triangles of the iso-surface over the cubes whose 8 voxels all have weights > 0 (and
 * fgVolMask != 0 when given: ObjTSDF::getMesh), and leaves the per-workgroup offsets in scratch.
 * counts_dev: device memory; read it back, allocate the outputs, then call emf_hip_meshEmit with
 * the same volume and the same scratch. */
int emf_hip_meshCount(const float* tsdf, const float* weights, const uint8_t* fgVolMask,
                      const int32_t res[3], void* scratch_dev, emf_mesh_counts_t* counts_dev,
                      emf_stream_t stream);

/* Pass 2 -- kernel_createTriangles: vertices and normals (3 floats each per vertex, volume frame)
 * and triangles (4 int32 each: 3, i0, i1, i2), element for element what the reference's mesh
 * holds.  grads: the N^3 x 3 gradient volume, or NULL to take the same forward differences on the
 * fly.  Normals are the interpolated gradients as they are -- the reference's normalisations are
 * no-ops (common.cuh:170-173). */
int emf_hip_meshEmit(const float* tsdf, const float* grads, const float* weights,
                     const uint8_t* fgVolMask, const int32_t res[3], float voxelSize,
                     const void* scratch_dev, float* vertices, float* normals, int32_t* triangles,
                     emf_stream_t stream);

/* The ignore_person block of EMFusion::render (EMFusion.cpp:139-150: compare, setTo, two masked
 * copyTo) in one launch: pixels labelled `id` get label 0 and the background's vertex / normal. */
int emf_hip_hideLabel(const emf_image_t* segmentation, int id, const emf_image_t* vertices,
                      const emf_image_t* normals, const emf_image_t* bgVertices, const emf_image_t* bgNormals,
                      emf_stream_t stream);

/* Replaces cuda::EMFusion::renderGPU (EMFusion.cu:100-186): Phong shading of the composited raycast
 * (vertices, normals f32x3; segmentation u8) into image (u8x3, RGB), coloured per label through
 * colorMap (256 x RGB, HOST memory, passed by value to the kernel).  Pixels without a vertex are
 * written as 0, so the reference's image.setTo(0) is not needed.  lightPos: the translation of
 * renderGPU's lightPose (EMFusion::render passes the identity, i.e. the camera centre). */
int emf_hip_renderPhong(const emf_image_t* vertices, const emf_image_t* normals,
                        const emf_image_t* segmentation, const uint8_t colorMap[768],
                        const float lightPos[3], const emf_image_t* image, emf_stream_t stream);

/* EMFusion::initObjsFromUnmatched's carving step (EMFusion.cpp:462-478): removes from the unmatched
 * instance mask `seg` (in place) the pixels the object `id` already claims -- its footprint in the
 * model segmentation, plus `matchMask` if a mask was matched to it (may be NULL) -- and counts the
 * mask's pixels before ([0]) and after ([1]).  The caller zeroes the mask if after / before < 0.5. */
int emf_hip_carveMask(const emf_image_t* seg, const emf_image_t* modelSeg, int id,
                      const emf_image_t* matchMask, uint32_t* counts_dev, emf_stream_t stream);

/* The two numbers of EMFusion::cleanUpObjs' association test (EMFusion.cpp:936-949):
 * count = |objSeg OR matchMask| (matchMask may be NULL), sum = sum of `assoc` over those pixels
 * (double accumulation like cv::cuda::sum, fixed order).  The object is spurious if
 * assocThresh * count > sum.  out_dev: emf_hip_maskAssociationMassBytes() bytes of device memory -- the answer in
 * out_dev[0], behind it the partials of the row bands (two launches; ABI 8: one emf_mask_mass_t used to suffice). */
typedef struct emf_mask_mass {
    double sum;
    uint32_t count;
    uint32_t pad_;
} emf_mask_mass_t;
size_t emf_hip_maskAssociationMassBytes(void);
int emf_hip_maskAssociationMass(const emf_image_t* objSeg, const emf_image_t* matchMask,
                                const emf_image_t* assoc, emf_mask_mass_t* out_dev,
                                emf_stream_t stream);

/* The counts behind EMFusion::matchSegmentation (EMFusion.cpp:797-825) for ALL objects at once:
 * counts_dev[0] = pixels of `seg`; counts_dev[1 + id] = |seg AND (modelSeg == id)|;
 * counts_dev[257 + id] = |modelSeg == id|, id = 1..255 (513 uint32, cleared by the call).
 * IoU(id) = inter / (counts[0] + area - inter). */
int emf_hip_maskOverlap(const emf_image_t* seg, const emf_image_t* modelSeg, uint32_t* counts_dev,
                        emf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Tracking: weighted Levenberg-Marquardt ICP on the TSDF
 * (TSDF.cpp:170-344, 375-395; EMFusion.cpp:672-724).  All LM state lives in device memory.
 * ---------------------------------------------------------------------------------------------- */

/* TSDFParams fields of the tracker (data.h:32-66) */
typedef struct emf_track_params {
    float huberThresh; /* 0.2  */
    float maxWeight;   /* maxTSDFWeight, 64 */
    float tau;         /* 1e3  */
    float eps1;        /* 1e-8 */
    float eps2;        /* 1e-8 */
    float nuInit;      /* 2    */
} emf_track_params_t;

/* Per-model Levenberg-Marquardt state (device memory; read it back after synchronising).
 * R, t = rel_pose_CO (camera -> volume), the quantity TSDF::prepareTracking sets up and
 * TSDF::syncTrack converts back: cam_pose = pose * rel_pose_CO. */
typedef struct emf_track_state {
    float R[9], t[3];
    float Rtrial[9], ttrial[3];
    float A[36], b[6], x[6];
    float mu, nu, rho, err, errNew;
    uint32_t maxIwBits;        /* float bits of max |min(intWeights, maxWeight)| at the current pose */
    uint32_t maxIwTrialBits;   /* ... at the trial pose */
    int32_t converged;         /* trackingConverged */
    int32_t firstIteration;
    int32_t evaluateGradient;
    int32_t haveTrial;
    int32_t iterations;        /* trial steps evaluated */
    int32_t accepted;          /* ... of which accepted (rho > 0) */
    int32_t iwSel;             /* which of the two weight images belongs to the current pose */
    /* bookkeeping of the fused step kernel (one launch per LM iteration, see emf_hip_trackIterate) */
    int32_t wSel;              /* which of the two per-pixel weight images belongs to the current pose */
    int32_t needAccum;         /* A, b, err must be summed at the current pose before the next solve */
    int32_t haveSpec;          /* spec[] holds them already (summed speculatively at the accepted trial pose) */
    float spec[28];            /* upper triangle of A (21), b (6), err -- moved into A, b, err by the next solve */
    int32_t checkB;            /* A, b are fresh: the max|b| < eps1 test is still to be made */
    int32_t pending;           /* what the last launch left in the partial sums: 0 nothing, 1 A/b/err at the
                                  current pose, 2 the trial step's error + A/b/err at the trial pose */
    int32_t body;              /* what the current launch does per pixel (same codes) */
    int32_t iterTarget;        /* `iterations` at which the current trackIterate call stops */
    /* |log| of the two poses (the step-size test of TSDF.cpp:292-296 needs the current pose's): kept with
     * the poses, so that no launch has to make them in front of its solve */
    float logCur, logTrial;
    /* factor on the current pose's per-pixel weight image: 1, or -- when a step was accepted whose weights had been
     * normalised by the previous pose's maximum although the accepted pose's is another -- the ratio of the two
     * maxima (see emf_hip_trackIterate) */
    float wFac;
} emf_track_state_t;

/* bytes of scratch per model for emf_hip_trackIterate on a width x height image */
size_t emf_hip_trackScratchBytes(int width, int height);

/* TSDF::prepareTracking for all models (TSDF.cpp:170-191): states[m] <- initial LM state with
 * rel_pose_CO = poseCO_host[m], which the caller has re-orthonormalised (the reference runs a
 * Householder QR of the rotation block on the host). */
int emf_hip_trackPrepare(emf_track_state_t* states_dev, const emf_pose_t* poseCO_host, int nmodels,
                         float nuInit, emf_stream_t stream);

/* `iterations` more LM iterations of every model in lock-step, as EMFusion::performTracking runs them
 * (EMFusion.cpp:673-684, 692-720), without host synchronisation: ONE launch per iteration.  Every
 * workgroup of a launch first finishes the previous launch in its own LDS copy of the state (adds the
 * partial sums in a fixed order, judges the pending trial step, solves for the next one -- the same
 * arithmetic in every workgroup, workgroup 0 stores the state), then evaluates the new trial pose per
 * pixel: its error under the current weights AND, speculatively, the Hessian sums the next iteration
 * needs if the step is accepted.  That speculation normalises the weights by the maximum integration
 * weight of the CURRENT pose (the trial pose's is known after the pass; it is the weight cap after a few
 * frames); when the two differ, the sums -- linear in the normaliser -- are multiplied by the ratio of the
 * maxima, and the accepted pose's weight image by the same factor where it is read (`wFac`): 2e-7 relative
 * from weights made anew.  (EMF_TRACK_RESCALE=0 in the environment: the sums are re-made at the accepted
 * pose by one extra launch, rounds 1-3.)  The call enqueues iterations + 3 launches (rounded up to even):
 * one spare for such an extra launch -- read `iterations` / `haveTrial` from the state to see how far a
 * model got; launches with nothing left to do return at once, as do converged models.  Each model's
 * `assoc` map supplies the association weights.
 * scratch_dev: nmodels * scratchBytesPerModel bytes (>= emf_hip_trackScratchBytes). */
int emf_hip_trackIterate(const emf_model_t* models_dev, emf_track_state_t* states_dev, int nmodels,
                         const emf_image_t* points, const emf_track_params_t* params,
                         void* scratch_dev, size_t scratchBytesPerModel, int iterations,
                         emf_stream_t stream);

/* The same loop, launch by launch, for a host that does not want to guess how many iterations a
 * stage needs: emf_hip_trackStep enqueues launch number `launch` (0, 1, 2, ... since the stage's
 * emf_hip_trackPrepare or since the last call of emf_hip_trackIterate; launch 0 also runs the weight-
 * maximum pass and lets every model do `iterations` more iterations) and has the device report to
 * `watch`, which must be host memory the device can write (hipHostMalloc, coherent):
 *   watch[0]      <- seq when the launch has begun (any nonzero number the caller increases per launch),
 *   watch[1 + m]  <- 1 when model m has converged, 2 when it has done its iterations, else 0 -- in the lower half;
 *                    the upper half is seq's (a stage tag, if the caller puts one there: launches of the previous stage
 *                    that are still queued write too).
 * These are hints to stop enqueuing (written with system-scope stores while the stream runs: keep a
 * few launches ahead of watch[0], stop when every watch[1 + m] != 0); the states are read as usual.
 * The number of launches of a stage must be even (the state alternates between the caller's array
 * and a shadow in the scratch): end with one more launch if it is not -- a launch with nothing to
 * do returns at once.  watch may be NULL.
 * finalStates (host memory the device can write, nmodels entries, or NULL): the launch that finds model m done
 * stores its state to finalStates[m] IN FRONT OF watch[1 + m] (system scope, release): a host that has seen the
 * word (and an acquire fence) reads the stage's result there -- no copy command, no wait for the stream. */
int emf_hip_trackStep(const emf_model_t* models_dev, emf_track_state_t* states_dev, int nmodels,
                      const emf_image_t* points, const emf_track_params_t* params,
                      void* scratch_dev, size_t scratchBytesPerModel, int launch, int iterations,
                      uint32_t* watch, uint32_t seq, emf_track_state_t* finalStates, emf_stream_t stream);

/* The two weight images a stage leaves behind, as the reference's debug output reads them at the end of a frame
 * (TSDF::getHuberWeights / getTrackingWeights, TSDF.cpp:346-354: `trackWeights` = min(huberThresh / |tsdf value|, 1)
 * with x / 0 := 0, TSDF.cpp:222-231; `intWeights` = that x the clamped, NORM_INF-normalised integration weights x the
 * association weights, TSDF.cpp:233-256) -- evaluated at the pose the stage ended with (states_dev[m].R / t), with the
 * body's own arithmetic (k_track_step): the tracker itself never materialises the Huber image and keeps only the
 * product.  Call it behind the stage's last launch and before anything overwrites the models' `assoc` maps.
 * huber_dev / track_dev: nmodels dense width x height float images each (either may be NULL). */
int emf_hip_trackWeightImages(const emf_model_t* models_dev, const emf_track_state_t* states_dev, int nmodels,
                              const emf_image_t* points, const emf_track_params_t* params,
                              const void* scratch_dev, size_t scratchBytesPerModel, float* huber_dev,
                              float* track_dev, emf_stream_t stream);

/* Level 1: replaces emf::cuda::TSDF::computePoseGradients (TSDF.cuh, TSDF.cu:603-660).
 * grads6: (W*H) x 6 f32, every row written (zeros where the reference leaves its setTo(0));
 * grads: N^3 x 3 gradient volume or NULL (forward differences blended on the fly, same values). */
int emf_hip_computePoseGradients(const float* tsdf, const float* grads, const emf_image_t* points,
                                 const float R_CO[9], const float t_CO[3], const int32_t res[3],
                                 float voxelSize, float* grads6, emf_stream_t stream);

/* ---- direct peer-write exchanges fused into the path's kernels (types: see the peer section above) ---- */
/* The E-step's exchange fused into the path (reference EMFusion.cpp:653-665; SURVEY 8e exchange 1):
 *   estepBatchedPeer          = emf_hip_estepBatched / ...FromDepth (depth != NULL) with normalize = 0 whose per-pixel
 *                               sum of the OBJECT maps goes straight into slot[me] (offset 0, W*H floats) of every peer;
 *   peerNormalizeAssociation  waits, sums the slots in rank order (-> objSum, optional), norm := maps[0] + that sum,
 *                               maps[k] := maps[k] / norm (x / 0 := 0): the values of peerWaitReduceSumF32 followed
 *                               by emf_hip_normalizeAssociation(maps, nmaps, 1, objSum, norm).  nmaps <= 16. */
int emf_hip_estepBatchedPeer(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                             const emf_image_t* depth, const float K[9], const emf_image_t* points,
                             const emf_peer_t* group, uint32_t seq, emf_stream_t stream);
int emf_hip_peerNormalizeAssociation(const emf_peer_t* group, uint32_t seq, const emf_image_t* maps_host, int nmaps,
                                     const emf_image_t* objSum, const emf_image_t* norm, emf_stream_t stream);
/* The raycast's exchange fused into the path (reference EMFusion.cpp:760-794; SURVEY 8e exchange 2 + the row bands
 * of the replicated background's raycast).  Slot layout per sender: [W*H u64 hit keys][W*H f32 background
 * raylengths][W*H u8 background hit mask] -- emf_hip_peerRaycastSlotBytes(W, H).
 *   packHitKeysPeer          = emf_hip_packHitKeys whose keys go into slot[me] of every peer, together with rows
 *                              [bandRow0, bandRow0 + bandRows) of bgRay / bgMask (bandRows = 0: no bands);
 *   compositeFromKeysPeer    waits, takes the minimum key over the slots, fetches the foreign bands of bgRay / bgMask
 *                              from their owners' slots (rows of owner r: [r * bandRowsPerRank, ...); also stored into
 *                              the local images), then composites exactly as emf_hip_compositeFromKeys and counts
 *                              visibility into visCounts (which must be zero at the launch's start);
 *   visibilityFlagsMirror    the integrate gate from those counts (as visibilityFlagsIndexed), the nall counts mirrored
 *                              to host-visible memory, visCounts left cleared for the next frame. */
size_t emf_hip_peerRaycastSlotBytes(int width, int height);
int emf_hip_packHitKeysPeer(int nlocal, const int32_t* listPos_host, const emf_image_t* objRay_host,
                            const emf_image_t* objSeg_host, const emf_image_t* bgRay, const emf_image_t* bgMask,
                            int bandRow0, int bandRows, const emf_peer_t* group, uint32_t seq, emf_stream_t stream);
int emf_hip_compositeFromKeysPeer(const emf_peer_t* group, uint32_t seq, int bandRowsPerRank, int nall,
                                  const int32_t* ids_host, int nlocal, const int32_t* listPos_host,
                                  const emf_image_t* objRay_host, const emf_image_t* objVert_host,
                                  const emf_image_t* objNorm_host, const emf_image_t* bgRay, const emf_image_t* bgVert,
                                  const emf_image_t* bgNorm, const emf_image_t* bgMask, const emf_image_t* ray,
                                  const emf_image_t* vert, const emf_image_t* norm, const emf_image_t* seg,
                                  const emf_image_t* diff, const emf_image_t* noObj, int boundary, int32_t* visCounts,
                                  emf_stream_t stream);
int emf_hip_visibilityFlagsMirror(int32_t* visCounts, int nall, int nmodels, const int32_t* countIndex_host,
                                  int visibilityThresh, int32_t* visible_dev, int32_t* countsMirror, emf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EMF_HIP_H */
