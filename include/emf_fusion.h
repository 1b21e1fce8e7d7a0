/*
 * emf_fusion.h -- C handle API over the C++ host classes (emf::EMFusion / emf::TSDF / emf::ObjTSDF,
 * emfusion_amd/csrc/core) for callers that cannot include C++ headers: the Python harness
 * (tests/, bench.py) and FFI users.  C++ callers -- such as a port of the reference's
 * apps/EM-Fusion.cpp main loop -- use the classes directly (see apps/emfusion_synth.cpp).
 *
 * The handle wraps one emf::EMFusion: one background volume + N object volumes, driven frame by
 * frame with externally supplied poses and masks (tracking and Mask R-CNN are outside this
 * build's scope).  Every function returns 0 on success, a negative EMF_E_* code or a positive
 * hipError_t / ncclResult_t; emf_fusion_last_error_string() describes the last failure on the
 * calling thread.  Nothing throws across this boundary.
 */
#ifndef EMF_FUSION_H
#define EMF_FUSION_H

#include "emf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct emf_fusion emf_fusion_t;
typedef struct emf_comm emf_comm_t;
typedef struct emf_synth emf_synth_t;

/* Mirror of the fields of emf::Params / emf::TSDFParams the volumetric path reads
 * (reference include/EMFusion/core/data.h; defaults = reference defaults). */
typedef struct emf_fusion_params {
    int32_t width, height;
    float K[9];
    int32_t bg_res[3];
    float bg_voxel_size;
    float bg_rel_truncdist;
    float volume_pose_t[3]; /* background volume centre relative to the first camera */
    int32_t obj_res[3];
    float obj_rel_truncdist;
    float max_tsdf_weight, assoc_sigma, alpha, uni_prior;
    int32_t visibility_thresh, boundary, mask_frames;
    int32_t materialize_gradients; /* 0: normals from on-the-fly differences; 1: gradient volume */
    int32_t max_tracking_iter;     /* LM iterations per tracking stage (data.h:106: 100) */
} emf_fusion_params_t;

/* per-stage GPU milliseconds of the last frame (HIP events on the main stream) */
typedef struct emf_frame_timings {
    float points, estep, raycast, composite, integrate, masks, total;
} emf_frame_timings_t;

enum emf_fusion_image {
    EMF_IMG_POINTS = 0,          /* f32x3 */
    EMF_IMG_BG_ASSOC = 1,        /* f32   normalised background association weights */
    EMF_IMG_OBJ_ASSOC = 2,       /* f32   per object (obj_id) */
    EMF_IMG_ASSOC_NORM = 3,      /* f32 */
    EMF_IMG_RAYLENGTHS = 4,      /* f32   composite */
    EMF_IMG_VERTICES = 5,        /* f32x3 composite */
    EMF_IMG_NORMALS = 6,         /* f32x3 composite */
    EMF_IMG_SEGMENTATION = 7,    /* u8    composite model segmentation */
    EMF_IMG_BG_RAYLENGTHS = 8,   /* f32 */
    EMF_IMG_OBJ_RAYLENGTHS = 9   /* f32   per object (obj_id) */
};

enum emf_fusion_volume {
    EMF_VOL_TSDF = 0,     /* f32 */
    EMF_VOL_WEIGHTS = 1,  /* f32 */
    EMF_VOL_FGPROBS = 2,  /* f32, objects only */
    EMF_VOL_FGMASK = 3,   /* u8,  objects only */
    EMF_VOL_BRICKS = 4    /* u8,  brick uniformity flags, ceil(N/4) per axis (res = brick grid) */
};

const char* emf_fusion_last_error_string(void);

void emf_fusion_default_params(emf_fusion_params_t* p);
/* comm may be NULL (single GPU).  The handle shares ownership of comm. */
int emf_fusion_create(const emf_fusion_params_t* p, emf_comm_t* comm, emf_fusion_t** out);
/* The instance apps/emfusion_synth --configfile builds: EVERY key of one of the reference's configuration files (config/default.cfg, tum.cfg ...;
 * apps/EM-Fusion.cpp:268-371) and, optionally, a Co-Fusion calibration.txt -- including the ones emf_fusion_params_t has no
 * field for (ignore_person, FILTER_CLASSES, STATIC_OBJECTS, huberThresh, tau, eps1, eps2, nu_init, bilateral_*, volPad,
 * existenceThresh, volIOUThresh, matchIOUThresh, distanceThresh, assocThresh).  path NULL or "": the defaults of data.h.
 * params_out (may be NULL) receives the subset the struct does carry. */
int emf_fusion_create_from_config(const char* path, const char* calibration, int materialize_gradients, emf_comm_t* comm,
                                  emf_fusion_params_t* params_out, emf_fusion_t** out);
void emf_fusion_destroy(emf_fusion_t* h);
int emf_fusion_reset(emf_fusion_t* h);
/* Device buffers released by destroyed / resized volumes wait in a process-wide pool instead of going through
 * hipFree (which synchronises the device; EMF_POOL_MIB caps the pool, default 16 GiB).  This really frees them
 * -- it waits for the device -- and reports how many bytes were held (bytes_freed may be NULL). */
int emf_fusion_trim_pool(uint64_t* bytes_freed);
/* emf::EMFusion::processFrame(const RGBD&) -- the reference's entry (EMFusion.h:66): a HOST depth image in metres
 * (width x height floats), uploaded, bilateral-filtered and run through the schedule with the frame inputs set by the
 * emf_fusion_set_* / queue_* calls; with emf_fusion_use_preproc_masks the instance masks of every mask frame come
 * from <path>/Mask%04d.plk (EMFusion::usePreprocMasks, EMFusion.h:98).  emf_fusion_get_last_masks: EMFusion::
 * getLastMasks (EMFusion.h:83) -- W x H x 3 bytes (may be NULL), *instances of the last mask frame. */
int emf_fusion_process_rgbd(emf_fusion_t* h, const float* depth_host, int32_t width, int32_t height);
int emf_fusion_use_preproc_masks(emf_fusion_t* h, const char* path);
int emf_fusion_get_last_masks(emf_fusion_t* h, uint8_t* rgb, size_t capacity, int32_t* instances);

/* Create an object volume (edge vol_size metres, obj_res voxels) centred at `center` in world
 * coordinates; every rank issues the same calls.  *id_out = object id (1-based). */
int emf_fusion_add_object(emf_fusion_t* h, const float center[3], float vol_size, int32_t* id_out);

/* Run one frame of the schedule (emf::EMFusion::processFrame) on a depth map resident in device
 * memory.  obj_R / obj_t: nposes x 9 / nposes x 3 floats for the object ids in pose_ids.
 * masks: device u8 0/1 images for the ids in mask_ids; used when run_masks != 0. */
int emf_fusion_process_frame(emf_fusion_t* h, const emf_image_t* depth_dev, const float cam_R[9],
                             const float cam_t[3], int nposes, const int32_t* pose_ids,
                             const float* obj_R, const float* obj_t, int nmasks,
                             const int32_t* mask_ids, const emf_image_t* masks, int run_masks);

/* Tracking (SURVEY f-1).  set_tracking: from the next frame on, process_frame ignores the supplied
 * camera pose / object poses and tracks them instead (EMFusion::performTracking); frame 0 always
 * takes the supplied poses.  get_pose: id 0 = camera -> world, else object volume -> world.
 * track_result: iterations / accepted / converged / error of the last run of model id. */
int emf_fusion_set_tracking(emf_fusion_t* h, int track_camera, int track_objects);
/* Object creation / matching from a device instance mask (u8 W x H, non-zero = inside) of the
 * current frame (EMFusion::initNewObjVolume / matchSegmentation, SURVEY f-3).  *id = new / matched
 * object id, or -1.  *iou is in/out for match (start it at 0). */
int emf_fusion_create_object_from_mask(emf_fusion_t* h, const emf_image_t* mask, int32_t* id);
/* In-frame creation, as the reference does it (initOrMatchObjs before integrateDepth): the masks
 * queued here are run through initNewObjVolume by the NEXT process_frame, after its raycast and
 * before its integration; last_created returns the ids (-1 = rejected), in queue order. */
int emf_fusion_queue_new_object_masks(emf_fusion_t* h, int n, const emf_image_t* masks);
/* The instance masks of a Mask R-CNN frame for the NEXT process_frame: it runs the reference's
 * initOrMatchObjs on them after its raycast (match, resolve double matches, carve and spawn the
 * unmatched -- the masks are MODIFIED in place), integrates the matched masks and, if clean-up is
 * on, deletes spurious objects.  last_mask_assignment: the object id each mask ended up with. */
int emf_fusion_queue_instance_masks(emf_fusion_t* h, int n, const emf_image_t* masks);
/* The class scores that go with the queued instance masks (n x num_classes doubles, mask-major;
 * MaskRCNN::getScores): a matched object accumulates them.  object_class: index of its largest
 * accumulated score (0 before any).  set_ignore_person: Params.ignore_person (config/tum.cfg) --
 * "person" objects (COCO class 1) stay out of renderings and mesh files. */
int emf_fusion_queue_instance_scores(emf_fusion_t* h, int n, int num_classes, const double* scores);
int emf_fusion_object_class(emf_fusion_t* h, int id, int32_t* class_id);
int emf_fusion_set_ignore_person(emf_fusion_t* h, int on);
/* geometry of an object volume as it is now (objects are created and resized inside frames): resolution, voxel size,
 * truncation distance [m], existence probability (ObjTSDF::getExProb); any output pointer may be NULL */
int emf_fusion_object_info(emf_fusion_t* h, int id, int32_t res[3], float* voxel_size, float* truncdist, float* existence);
int emf_fusion_last_mask_assignment(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count);
int emf_fusion_last_created(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count);
int emf_fusion_match_mask(emf_fusion_t* h, const emf_image_t* mask, int32_t* id, float* iou);
/* EMFusion::updateObj + ObjTSDF::resize (EMFusion.cpp:827-863, ObjTSDF.cpp:80-165) for one object and
 * one mask of the current frame's points: the volume grows / recentres if the 10th..90th percentile
 * box of (surface vertices + masked points) leaves it.  offset: the centre shift in the old volume
 * frame (all 0: nothing changed).  process_frame does this for every matched instance mask. */
int emf_fusion_update_object(emf_fusion_t* h, int id, const emf_image_t* mask, float offset[3]);
/* From the next frame on, run the reference's cleanUpObjs at the end of every frame (delete objects
 * that are not visible, whose association mass does not fit their mask, or -- on mask frames --
 * whose existence probability is low); last_deleted lists the ids the last frame removed. */
int emf_fusion_set_cleanup(emf_fusion_t* h, int on);
int emf_fusion_last_deleted(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count);
/* Results in the reference's formats (SURVEY f-4, mesh-free part): enable the per-frame pose log
 * before processing, then write <dir>/poses-cam.txt, poses-<id>.txt (TUM: "frame tx ty tz qx qy qz
 * qw") and, if volumes != 0, <dir>/tsdfs/{bg_tsdf,tsdf_<id>,weights_<id>,fgProbs_<id>}.bin
 * (int32 res[3], uint64 element size, float voxel size, voxels).  emf_io_* are host-only helpers. */
/* TSDF::getMesh / ObjTSDF::getMesh of model `id` (0 = background): extract runs marching cubes and
 * keeps the result in the handle, copy hands it out (vertices, normals: 3 floats per vertex;
 * triangles: 4 int32 per triangle = 3, i0, i1, i2).  emf_io_write_mesh writes the reference's PLY. */
int emf_fusion_extract_mesh(emf_fusion_t* h, int id, uint32_t* num_vertices, uint32_t* num_triangles);
int emf_fusion_copy_mesh(emf_fusion_t* h, float* vertices, float* normals, int32_t* triangles);
int emf_io_write_mesh(const char* filename, uint32_t num_vertices, const float* vertices,
                      const float* normals, uint32_t num_triangles, const int32_t* triangles);
/* EMFusion::render (EMFusion.cpp:131-160): Phong-shaded RGB view of the models, width*height*3 bytes
 * into host memory; color_map (may be NULL) receives the 256 x RGB label colours. */
int emf_fusion_render(emf_fusion_t* h, uint8_t* rgb, uint8_t* color_map);
/* Multi-GPU: broadcast the depth image of every frame from rank `root` (whose process_frame argument
 * is the source; on the other ranks it is the destination and must have the same size and pitch)
 * before anything else runs.  root < 0 (default): every rank is handed the frame itself. */
int emf_fusion_set_depth_broadcast(emf_fusion_t* h, int root);
int emf_fusion_enable_pose_log(emf_fusion_t* h, int on);
/* EMFusion::setupOutput (EMFusion.cpp:243-247): log on; exp_vols != 0 also keeps the volumes of
 * objects deleted during the run for write_results.  exp_frame_meshes is accepted and ignored. */
int emf_fusion_setup_output(emf_fusion_t* h, int exp_frame_meshes, int exp_vols);
/* EMFusion::writeResults (EMFusion.cpp:248-292): pose files, mesh_bg.ply and mesh_<id>.ply always;
 * tsdfs/ *.bin only if volumes != 0 or setup_output asked for them. */
int emf_fusion_write_results(emf_fusion_t* h, const char* dir, int volumes);
int emf_io_write_volume(const char* filename, const float* voxels, const int32_t res[3], float voxel_size);
int emf_io_write_pose_file(const char* filename, int n, const int32_t* frames, const float* R,
                           const float* t);
/* Undo the PNG scan-line filters (PNG specification 9.2; what cv::imread does inside
 * TUMRGBDReader.cpp for the depth images): rows = height x (1 + stride) bytes, each line preceded by
 * its filter type 0..4; out = height x stride reconstructed bytes; bpp = bytes per pixel (1 or 2). */
int emf_io_png_unfilter(const uint8_t* rows, int height, int stride, int bpp, uint8_t* out);
/* The C++ dataset readers behind apps/emfusion_synth --sequence (core/Readers.hpp; reference
 * src/utils/TUMRGBDReader.cpp, src/core/MaskRCNN.cpp:250-282), exposed for tests and FFI users.
 *   read_depth_png     an 8/16-bit grayscale PNG as float = raw * scale (TUM: 1 / 5000); out may be NULL to ask
 *                      for the size only; capacity in floats
 *   tum_associations   entry `index` of <file>: depth file name and time stamp; *count = number of entries
 *   load_preproc_masks a Mask%04d.plk of the reference's preprocessing: *n instances of *width x *height; masks
 *                      (n * height * width bytes, 0/1), boxes (n * 4) and scores (n * *nscores) are filled when
 *                      not NULL and large enough (mask_capacity in bytes, score_capacity in doubles) */
int emf_io_read_depth_png(const char* path, float scale, float* out, size_t capacity, int32_t* width, int32_t* height);
/*   read_exr           one channel (NULL / "": the only one, else the first of Z, Y, R) of a single-part scan-line
 *                      OpenEXR file as float (emf::readExr; reference src/utils/ImageReader.cpp:105-110 reads its
 *                      depth files with cv::imread); out may be NULL to ask for the size only
 *   image_reader       emf::ImageReader on <base><colordir> / <base><depthdir> (ColorNNNN.png / DepthNNNN.exr):
 *                      number of frames and first index; the reference's error messages for unusable directories */
/*   load_config        emf::loadConfigFile (core/Config.hpp): the reference's config files (config/default.cfg ...;
 *                      apps/EM-Fusion.cpp:268-371) applied to the reference defaults, then -- if calibration is not
 *                      NULL and the file exists -- <dir>/calibration.txt (EM-Fusion.cpp:399-410); the fields the
 *                      volumetric path reads come back in *p (may be NULL), every configurable field as
 *                      "Section.key = value" lines in dump (may be NULL; truncated to dump_capacity - 1 characters) */
int emf_io_load_config(const char* path, const char* calibration, emf_fusion_params_t* p, char* dump, size_t dump_capacity);
int emf_io_read_exr(const char* path, const char* channel, float* out, size_t capacity, int32_t* width, int32_t* height);
int emf_io_image_reader(const char* base, const char* colordir, const char* depthdir, int32_t* num_frames, int32_t* first);
int emf_io_tum_associations(const char* file, int index, char* depth_name, int name_capacity, double* stamp, int32_t* count);
/* capacities in elements (bytes of masks, doubles of boxes and scores); an output whose capacity is too small is not written */
int emf_io_load_preproc_masks(const char* path, int32_t* n, int32_t* width, int32_t* height, uint8_t* masks,
                              size_t mask_capacity, double* boxes, size_t box_capacity, double* scores, size_t score_capacity,
                              int32_t* nscores);
/* from the next frame on, filter the incoming depth (EMFusion::preprocessDepth, SURVEY f-2) */
int emf_fusion_set_preprocess(emf_fusion_t* h, int on);
int emf_fusion_get_pose(emf_fusion_t* h, int id, float R[9], float t[3]);
int emf_fusion_track_result(emf_fusion_t* h, int id, int32_t* iterations, int32_t* accepted,
                            int32_t* converged, float* error);

/* Individual stages (emf::EMFusion::{computeAssociationWeights, raycast, integrateDepth}) acting
 * on the state left by the last process_frame; for stage-level tests and profiling. */
int emf_fusion_stage_estep(emf_fusion_t* h);
int emf_fusion_stage_raycast(emf_fusion_t* h);
int emf_fusion_stage_integrate(emf_fusion_t* h);

int emf_fusion_synchronize(emf_fusion_t* h);
int emf_fusion_enable_timings(emf_fusion_t* h, int on);
int emf_fusion_last_timings(emf_fusion_t* h, emf_frame_timings_t* out);
/* counters: [0] march samples, [1] hits, [2] samples that read the volume, [3] samples
 * fast-forwarded inside uniform bricks -- accumulated by raycast while enabled */
int emf_fusion_enable_raycast_stats(emf_fusion_t* h, int on);
int emf_fusion_raycast_stats(emf_fusion_t* h, uint64_t counters[4]);

/* Per-launch HIP-event timers: every kernel launch of the schedule is bracketed by an event pair
 * on the stream it is launched on.  enable(max_launches) allocates the pool (0 = off); collect()
 * must be called with the device idle (after emf_fusion_synchronize). */
enum emf_kernel_kind {
    EMF_K_POINTS = 0, EMF_K_ASSOC, EMF_K_NORMALIZE, EMF_K_RAYCAST, EMF_K_COMPOSITE,
    EMF_K_INTEGRATE, EMF_K_GRADS, EMF_K_FGBG, EMF_K_TRACK, EMF_K_INTEGRATE_BG, EMF_K_NUM_KINDS
};
typedef struct emf_kernel_summary {
    uint64_t launches;
    double total_ms;
    double units; /* voxels (volume sweeps) or pixels (image kernels), summed over launches */
} emf_kernel_summary_t;
int emf_fusion_kernel_timers_enable(emf_fusion_t* h, uint64_t max_launches);
int emf_fusion_kernel_timers_clear(emf_fusion_t* h);
/* restrict the event pairs to the kinds whose bit (1 << emf_kernel_kind) is set; default all */
int emf_fusion_kernel_timers_select(emf_fusion_t* h, uint32_t kind_mask);
/* Bracket only every `every`-th launch of a kind (1 = all): the event records themselves cost the frame. */
int emf_fusion_kernel_timers_stride(emf_fusion_t* h, uint32_t every);
int emf_fusion_kernel_timers_collect(emf_fusion_t* h, emf_kernel_summary_t out[EMF_K_NUM_KINDS],
                                     uint64_t* dropped);

/* Device views of per-frame images / volumes (valid until the next frame / destroy).
 * obj_id is ignored unless the selector is per object; 0 selects the background volume. */
int emf_fusion_get_image(emf_fusion_t* h, int which, int obj_id, emf_image_t* view);
int emf_fusion_get_volume(emf_fusion_t* h, int which, int obj_id, void** dev_ptr, int32_t res[3]);
/* ids of the objects classified visible by the last raycast; returns count in *n (<= cap) */
int emf_fusion_visible_objects(emf_fusion_t* h, int32_t* ids, int cap, int* n);
/* ids of all live objects of the job in creation order (deleted ones are gone); count in *n (<= cap) */
int emf_fusion_object_ids(emf_fusion_t* h, int32_t* ids, int cap, int* n);
int emf_fusion_frame_index(emf_fusion_t* h);
/* 1 if the background's integration runs out of place beside the raycast (double-buffered background) */
int emf_fusion_background_overlap(emf_fusion_t* h);
/* Which path the frames run on: 0 = per-volume (one stream per volume, host visibility gate: EMF_PER_VOLUME=1, materialised
 * gradients), k >= 1 = batched with k launches per stage (1 up to EMF_MAX_BATCH models, then one per chunk of the table) */
int emf_fusion_batched_chunks(emf_fusion_t* h);
/* Host time emf_fusion_process_rgbd has spent so far handing depth maps to the device -- staging memcpy + enqueue with the
 * double-buffered pinned upload (default), the blocking pageable copy with EMF_ASYNC_UPLOAD=0 -- and the frames it covers */
int emf_fusion_upload_host_time(emf_fusion_t* h, double* seconds, uint64_t* frames);
/* 1 if this rank holds object id's volume */
int emf_fusion_owns_object(emf_fusion_t* h, int obj_id);

/* ---- RCCL communicator for the object-sharded multi-GPU path (one process per GPU) ---- */
#define EMF_COMM_UNIQUE_ID_BYTES 128
int emf_comm_unique_id(void* out128);
int emf_comm_create(const void* unique_id128, int rank, int world, emf_comm_t** out);
void emf_comm_destroy(emf_comm_t* c);
/* Rehearsal backend: `world` communicators of one process sharing one GPU, one per host thread (RCCL
 * refuses two ranks on a device).  N emf_fusion handles driven from N threads then run the code path
 * of an N-GPU job; collectives are staged through host memory.  out: array of `world` handles. */
int emf_comm_create_local_group(int world, emf_comm_t** out);
/* Direct peer-write exchanges (include/emf_hip.h "direct peer-write exchanges"): no library collective,
 * three small launches per exchange.  slot_bytes bounds one message (W * H * 8 for the hit keys).
 *   ..._peer_local_group: `world` (<= 8) ranks of one process sharing a GPU, for N handles on N threads;
 *   ..._peer            : one process per rank; the receive buffers are mapped into every peer through
 *                         hipIpc*, the 128 handle bytes per rank travel through `all_gather` (0 = success;
 *                         all[r * bytes ...] := rank r's block), e.g. torch.distributed over gloo.
 * Untested on xGMI (no multi-GPU box in the build environment); exercised on one GPU. */
typedef int (*emf_allgather_fn)(void* user, const void* mine, size_t bytes, void* all);
int emf_comm_create_peer_local_group(int world, size_t slot_bytes, emf_comm_t** out);
int emf_comm_create_peer(int rank, int world, size_t slot_bytes, emf_allgather_fn all_gather, void* user,
                         emf_comm_t** out);
/* The four exchanges of a communicator, callable on their own (tests of a transport without a frame around
 * it).  dev pointers on the current device, `stream` a hipStream_t or NULL. */
int emf_comm_all_reduce_sum_f32(emf_comm_t* c, float* dev, size_t count, void* stream);
int emf_comm_all_reduce_min_u64(emf_comm_t* c, uint64_t* dev, size_t count, void* stream);
int emf_comm_broadcast(emf_comm_t* c, void* dev, size_t bytes, int root, void* stream);
int emf_comm_gather_row_bands(emf_comm_t* c, void* dev, size_t bytes_per_row, int band_rows, int total_rows,
                              void* stream);
/* Latency model around another communicator (which must outlive the new handle's users but may be
 * destroyed after it): every exchange -- a grouped one counts once -- first keeps its stream busy for
 * `microseconds`.  With a 1-rank RCCL communicator and EMF_FORCE_SHARDED=1 it measures, on one GPU, how
 * much per-collective latency the frame's schedule hides.  emf_comm_exchanges: exchanges issued so far
 * through a delayed communicator (0 for the others). */
int emf_comm_create_delayed(emf_comm_t* inner, int microseconds, emf_comm_t** out);
int emf_comm_exchanges(emf_comm_t* c, uint64_t* out);
/* What the transport reports about this rank (asked of RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice /
 * ncclGetVersion, + the device's PCI bus id), as a JSON object in `json` (cap >= 512): {"transport", "ranks", "rank",
 * "device", "pci_bus_id", "version"}.  bench.py --gpus N gathers one per rank into its line. */
int emf_comm_describe(emf_comm_t* c, char* json, size_t cap);
/* Rehearsal backend, one process per rank: collectives are staged through host memory and handed to
 * the caller's functions (0 = success), e.g. torch.distributed over gloo -- lets the N-rank job run on
 * a box with fewer than N GPUs (bench.py --comm gloo). */
typedef struct emf_comm_callbacks {
    int32_t rank, world;
    int (*all_reduce_sum_f32)(void* user, float* host, size_t count);
    int (*all_reduce_min_u64)(void* user, uint64_t* host, size_t count);
    int (*broadcast)(void* user, void* host, size_t bytes, int root);
    void* user;
} emf_comm_callbacks_t;
int emf_comm_create_host_staged(const emf_comm_callbacks_t* cb, emf_comm_t** out);

/* ---- synthetic RGB-D stream (host side, replaces the dataset readers) ---- */
int emf_synth_create(int width, int height, const float K[9], int num_spheres, uint64_t seed,
                     float noise_sigma, float dropout, emf_synth_t** out);
void emf_synth_destroy(emf_synth_t* s);
/* depth: HOST float[w*h]; ids: HOST u8[w*h] or NULL */
int emf_synth_render(emf_synth_t* s, int frame, float* depth, uint8_t* ids);
int emf_synth_camera_pose(emf_synth_t* s, int frame, float R[9], float t[3]);
int emf_synth_sphere(emf_synth_t* s, int k, int frame, float center[3], float* radius,
                     float* volume_size);

#ifdef __cplusplus
}
#endif
#endif /* EMF_FUSION_H */
