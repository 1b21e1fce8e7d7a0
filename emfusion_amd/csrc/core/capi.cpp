// capi.cpp -- extern "C" handle API of include/emf_fusion.h over the C++ host classes.
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>

#include "EMFusion.hpp"
#include "Config.hpp"
#include "Readers.hpp"
#include "Output.hpp"
#include "SyntheticScene.hpp"
#include "emf_fusion.h"

using namespace emf;

struct emf_comm {
    std::shared_ptr<Communicator> impl;
};
struct emf_fusion {
    std::unique_ptr<EMFusion> impl;
    bool trackCamera = false, trackObjects = false, preprocess = false, cleanUp = false;
    std::vector<emf_image_t> queuedMasks, queuedInstances;
    std::vector<std::vector<double>> queuedScores;
    emf::Mesh mesh;  // result of the last emf_fusion_extract_mesh
};
struct emf_synth {
    std::unique_ptr<SyntheticScene> impl;
};

namespace {

thread_local char g_err[512] = {0};

int report(const std::exception& e) {
    std::snprintf(g_err, sizeof(g_err), "%s", e.what());
    if (const auto* he = dynamic_cast<const HipError*>(&e)) return he->code() ? he->code() : -100;
    return -100;
}

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return EMF_OK;
    } catch (const std::exception& e) {
        return report(e);
    } catch (...) {
        std::snprintf(g_err, sizeof(g_err), "unknown exception");
        return -100;
    }
}

int nullArg(const char* fn, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s is NULL", fn, what);
    return EMF_E_NULL;
}

#define REQ(p)                                  \
    do {                                        \
        if (!(p)) return nullArg(__func__, #p); \
    } while (0)

Matx33f m33(const float* a) {
    return Matx33f(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
}

}  // namespace

extern "C" {

const char* emf_fusion_last_error_string(void) { return g_err; }

void emf_fusion_default_params(emf_fusion_params_t* p) {
    if (!p) return;
    const Params d;
    p->width = d.frameSize.width;
    p->height = d.frameSize.height;
    std::memcpy(p->K, d.intr.val, sizeof(p->K));
    std::memcpy(p->bg_res, d.globalVolumeDims.val, sizeof(p->bg_res));
    p->bg_voxel_size = d.globalVoxelSize;
    p->bg_rel_truncdist = d.globalRelTruncDist;
    std::memcpy(p->volume_pose_t, d.volumePose.translation().val, sizeof(p->volume_pose_t));
    std::memcpy(p->obj_res, d.objVolumeDims.val, sizeof(p->obj_res));
    p->obj_rel_truncdist = d.objRelTruncDist;
    p->max_tsdf_weight = d.tsdfParams.maxTSDFWeight;
    p->assoc_sigma = d.tsdfParams.assocSigma;
    p->alpha = d.tsdfParams.alpha;
    p->uni_prior = d.tsdfParams.uniPrior;
    p->visibility_thresh = d.visibilityThresh;
    p->boundary = d.boundary;
    p->mask_frames = d.maskRCNNFrames;
    p->materialize_gradients = 0;
    p->max_tracking_iter = d.maxTrackingIter;
}

int emf_fusion_create(const emf_fusion_params_t* p, emf_comm_t* comm, emf_fusion_t** out) {
    REQ(p);
    REQ(out);
    return guarded([&] {
        Params q;
        q.frameSize = Size(p->width, p->height);
        q.intr = m33(p->K);
        q.globalVolumeDims = Vec3i(p->bg_res[0], p->bg_res[1], p->bg_res[2]);
        q.globalVoxelSize = p->bg_voxel_size;
        q.globalRelTruncDist = p->bg_rel_truncdist;
        q.volumePose = Affine3f().translate(
            Vec3f(p->volume_pose_t[0], p->volume_pose_t[1], p->volume_pose_t[2]));
        q.objVolumeDims = Vec3i(p->obj_res[0], p->obj_res[1], p->obj_res[2]);
        q.objRelTruncDist = p->obj_rel_truncdist;
        q.tsdfParams.maxTSDFWeight = p->max_tsdf_weight;
        q.tsdfParams.assocSigma = p->assoc_sigma;
        q.tsdfParams.alpha = p->alpha;
        q.tsdfParams.uniPrior = p->uni_prior;
        q.visibilityThresh = p->visibility_thresh;
        q.boundary = p->boundary;
        q.maskRCNNFrames = p->mask_frames;
        if (p->max_tracking_iter > 0) q.maxTrackingIter = p->max_tracking_iter;
        auto h = std::make_unique<emf_fusion>();
        h->impl = std::make_unique<EMFusion>(
            q, p->materialize_gradients ? TSDF::Gradients::Materialized : TSDF::Gradients::OnTheFly,
            comm ? comm->impl : nullptr);
        *out = h.release();
    });
}

int emf_fusion_create_from_config(const char* path, const char* calibration, int materialize_gradients, emf_comm_t* comm,
                                  emf_fusion_params_t* params_out, emf_fusion_t** out) {
    REQ(out);
    return guarded([&] {
        // every field of the reference's Params, as apps/emfusion_synth --configfile builds them (apps/EM-Fusion.cpp:268-371):
        // emf_fusion_params_t carries a subset only (no ignore_person, LM / Huber / bilateral / lifecycle thresholds)
        Params q;
        if (path && path[0]) loadConfigFile(q, path);
        if (calibration && calibration[0]) loadCalibrationFile(q, calibration);
        auto h = std::make_unique<emf_fusion>();
        h->impl = std::make_unique<EMFusion>(q, materialize_gradients ? TSDF::Gradients::Materialized : TSDF::Gradients::OnTheFly,
                                             comm ? comm->impl : nullptr);
        if (params_out) {
            const int rc = emf_io_load_config(path, calibration, params_out, nullptr, 0);
            if (rc != 0) throw HipError("create_from_config: the configuration could not be re-read", rc);
            params_out->materialize_gradients = materialize_gradients;
        }
        *out = h.release();
    });
}

void emf_fusion_destroy(emf_fusion_t* h) { delete h; }

int emf_fusion_trim_pool(uint64_t* bytes_freed) {
    return guarded([&] {
        if (bytes_freed) *bytes_freed = emf::DeviceBuffer::pooledBytes();
        emf::DeviceBuffer::trimPool();
    });
}

int emf_fusion_reset(emf_fusion_t* h) {
    REQ(h);
    return guarded([&] { h->impl->reset(); });
}

int emf_fusion_add_object(emf_fusion_t* h, const float center[3], float vol_size,
                          int32_t* id_out) {
    REQ(h);
    REQ(center);
    return guarded([&] {
        const int id = h->impl->addObject(Vec3f(center[0], center[1], center[2]), vol_size);
        h->impl->settleReciprocals();  // an explicit call, outside any frame: the check ends here, not beside the first frames
        if (id_out) *id_out = id;
    });
}

int emf_fusion_process_frame(emf_fusion_t* h, const emf_image_t* depth_dev, const float cam_R[9],
                             const float cam_t[3], int nposes, const int32_t* pose_ids,
                             const float* obj_R, const float* obj_t, int nmasks,
                             const int32_t* mask_ids, const emf_image_t* masks, int run_masks) {
    REQ(h);
    REQ(depth_dev);
    REQ(cam_R);
    REQ(cam_t);
    if (nposes > 0) {
        REQ(pose_ids);
        REQ(obj_R);
        REQ(obj_t);
    }
    if (nmasks > 0) {
        REQ(mask_ids);
        REQ(masks);
    }
    return guarded([&] {
        FrameInputs in;
        in.cam_pose = Affine3f(m33(cam_R), Vec3f(cam_t[0], cam_t[1], cam_t[2]));
        for (int i = 0; i < nposes; ++i)
            in.obj_poses[pose_ids[i]] = Affine3f(
                m33(obj_R + 9 * i), Vec3f(obj_t[3 * i], obj_t[3 * i + 1], obj_t[3 * i + 2]));
        for (int i = 0; i < nmasks; ++i) in.masks[mask_ids[i]] = masks[i];
        in.runMasks = run_masks != 0;
        in.trackCamera = h->trackCamera;
        in.trackObjects = h->trackObjects;
        in.preprocessDepth = h->preprocess;
        in.cleanUp = h->cleanUp;
        in.newObjectMasks.swap(h->queuedMasks);
        in.instanceMasks.swap(h->queuedInstances);
        in.instanceScores.swap(h->queuedScores);
        h->impl->processFrame(*depth_dev, in);
    });
}

int emf_fusion_process_rgbd(emf_fusion_t* h, const float* depth_host, int32_t width, int32_t height) {
    REQ(h);
    REQ(depth_host);
    return guarded([&] {
        FrameInputs in;  // poses stay as tracked (or identity): the reference's entry takes nothing but the frame
        in.cam_pose = h->impl->getCameraPose();
        in.trackCamera = h->trackCamera;
        in.trackObjects = h->trackObjects;
        in.cleanUp = h->cleanUp;
        in.newObjectMasks.swap(h->queuedMasks);
        in.instanceMasks.swap(h->queuedInstances);
        in.instanceScores.swap(h->queuedScores);
        h->impl->setFrameInputs(in);
        RGBD frame;
        frame.size = Size(width, height);
        frame.depth = depth_host;
        h->impl->processFrame(frame);
    });
}

int emf_fusion_use_preproc_masks(emf_fusion_t* h, const char* path) {
    REQ(h);
    REQ(path);
    return guarded([&] { h->impl->usePreprocMasks(path); });
}

int emf_fusion_get_last_masks(emf_fusion_t* h, uint8_t* rgb, size_t capacity, int32_t* instances) {
    REQ(h);
    return guarded([&] {
        std::vector<uint8_t> img;
        const int n = h->impl->getLastMasks(img);
        if (instances) *instances = n;
        if (rgb && capacity >= img.size() && !img.empty()) std::memcpy(rgb, img.data(), img.size());
    });
}

int emf_io_read_depth_png(const char* path, float scale, float* out, size_t capacity, int32_t* width, int32_t* height) {
    REQ(path);
    return guarded([&] {
        std::vector<uint16_t> px;
        int w = 0, hgt = 0;
        readPngGray(path, px, w, hgt);
        if (width) *width = w;
        if (height) *height = hgt;
        if (out) {
            if (capacity < px.size()) throw HipError("emf_io_read_depth_png: buffer too small", EMF_E_ARG);
            for (size_t i = 0; i < px.size(); ++i) out[i] = static_cast<float>(px[i]) * scale;
        }
    });
}

int emf_io_load_config(const char* path, const char* calibration, emf_fusion_params_t* p, char* dump, size_t dump_capacity) {
    return guarded([&] {
        Params q;
        if (path && path[0]) loadConfigFile(q, path);
        if (calibration && calibration[0]) loadCalibrationFile(q, calibration);
        if (p) {
            emf_fusion_default_params(p);
            p->width = q.frameSize.width;
            p->height = q.frameSize.height;
            for (int k = 0; k < 9; ++k) p->K[k] = q.intr.val[k];
            for (int k = 0; k < 3; ++k) {
                p->bg_res[k] = q.globalVolumeDims[k];
                p->obj_res[k] = q.objVolumeDims[k];
                p->volume_pose_t[k] = q.volumePose.translation()[k];
            }
            p->bg_voxel_size = q.globalVoxelSize;
            p->bg_rel_truncdist = q.globalRelTruncDist;
            p->obj_rel_truncdist = q.objRelTruncDist;
            p->max_tsdf_weight = q.tsdfParams.maxTSDFWeight;
            p->assoc_sigma = q.tsdfParams.assocSigma;
            p->alpha = q.tsdfParams.alpha;
            p->uni_prior = q.tsdfParams.uniPrior;
            p->visibility_thresh = q.visibilityThresh;
            p->boundary = q.boundary;
            p->mask_frames = q.maskRCNNFrames;
            p->max_tracking_iter = q.maxTrackingIter;
        }
        if (dump && dump_capacity) std::snprintf(dump, dump_capacity, "%s", dumpConfig(q).c_str());
    });
}

int emf_io_read_exr(const char* path, const char* channel, float* out, size_t capacity, int32_t* width, int32_t* height) {
    REQ(path);
    return guarded([&] {
        std::vector<float> px;
        const Size s = readExr(path, px, channel ? channel : "");
        if (width) *width = s.width;
        if (height) *height = s.height;
        if (out) {
            if (capacity < px.size()) throw HipError("emf_io_read_exr: buffer too small", EMF_E_ARG);
            std::copy(px.begin(), px.end(), out);
        }
    });
}

int emf_io_image_reader(const char* base, const char* colordir, const char* depthdir, int32_t* num_frames, int32_t* first) {
    REQ(base);
    REQ(colordir);
    REQ(depthdir);
    return guarded([&] {
        const ImageReader r(base, colordir, depthdir);
        if (num_frames) *num_frames = static_cast<int32_t>(r.getNumFrames());
        if (first) *first = r.firstIndex();
    });
}

int emf_io_tum_associations(const char* file, int index, char* depth_name, int name_capacity, double* stamp, int32_t* count) {
    REQ(file);
    return guarded([&] {
        std::vector<std::string> rgb, depth;
        std::vector<double> stamps;
        TUMRGBDReader::readFileAssociations(file, rgb, depth, &stamps);
        if (count) *count = static_cast<int32_t>(depth.size());
        if (index >= 0 && index < static_cast<int>(depth.size())) {
            if (depth_name && name_capacity > 0) std::snprintf(depth_name, static_cast<size_t>(name_capacity), "%s", depth[index].c_str());
            if (stamp) *stamp = stamps[index];
        }
    });
}

int emf_io_load_preproc_masks(const char* path, int32_t* n, int32_t* width, int32_t* height, uint8_t* masks,
                              size_t mask_capacity, double* boxes, size_t box_capacity, double* scores, size_t score_capacity,
                              int32_t* nscores) {
    REQ(path);
    return guarded([&] {
        PreprocMasks pm;
        const int k = loadPreprocessedMasks(path, pm);
        if (n) *n = k;
        if (width) *width = pm.width;
        if (height) *height = pm.height;
        const size_t per = static_cast<size_t>(pm.width) * pm.height;
        const size_t ns = pm.scores.empty() ? 0 : pm.scores[0].size();
        for (const auto& row : pm.scores)  // (a list of lists may be ragged; the flat output is not)
            if (row.size() != ns) throw HipError("emf_io_load_preproc_masks: class-score rows of different lengths", EMF_E_ARG);
        if (nscores) *nscores = static_cast<int32_t>(ns);
        if (masks && mask_capacity >= per * k)
            for (int i = 0; i < k; ++i) std::memcpy(masks + per * i, pm.masks[i].data(), per);
        if (boxes && box_capacity >= 4 * static_cast<size_t>(k))
            for (int i = 0; i < k; ++i) std::memcpy(boxes + 4 * i, pm.boxes[i].data(), 4 * sizeof(double));
        if (scores && score_capacity >= ns * pm.scores.size())
            for (size_t i = 0; i < pm.scores.size(); ++i) std::memcpy(scores + ns * i, pm.scores[i].data(), ns * sizeof(double));
    });
}

int emf_fusion_set_tracking(emf_fusion_t* h, int track_camera, int track_objects) {
    REQ(h);
    h->trackCamera = track_camera != 0;
    h->trackObjects = track_objects != 0;
    return EMF_OK;
}

int emf_fusion_create_object_from_mask(emf_fusion_t* h, const emf_image_t* mask, int32_t* id) {
    REQ(h);
    REQ(mask);
    REQ(id);
    return guarded([&] { *id = h->impl->initNewObjVolume(*mask); });
}

int emf_fusion_queue_new_object_masks(emf_fusion_t* h, int n, const emf_image_t* masks) {
    REQ(h);
    if (n > 0) REQ(masks);
    return guarded([&] { h->queuedMasks.assign(masks, masks + (n > 0 ? n : 0)); });
}

int emf_fusion_queue_instance_masks(emf_fusion_t* h, int n, const emf_image_t* masks) {
    REQ(h);
    if (n > 0) REQ(masks);
    return guarded([&] { h->queuedInstances.assign(masks, masks + (n > 0 ? n : 0)); });
}

int emf_fusion_queue_instance_scores(emf_fusion_t* h, int n, int num_classes, const double* scores) {
    REQ(h);
    if (n > 0) REQ(scores);
    return guarded([&] {
        h->queuedScores.clear();
        for (int i = 0; i < n; ++i)
            h->queuedScores.emplace_back(scores + static_cast<size_t>(i) * num_classes,
                                         scores + static_cast<size_t>(i + 1) * num_classes);
    });
}

int emf_fusion_object_class(emf_fusion_t* h, int id, int32_t* class_id) {
    REQ(h);
    REQ(class_id);
    return guarded([&] {
        const ObjTSDF* o = h->impl->getObject(id);
        if (!o) throw HipError("object_class: no object " + std::to_string(id), EMF_E_ARG);
        *class_id = o->getClassID();
    });
}

int emf_fusion_object_info(emf_fusion_t* h, int id, int32_t res[3], float* voxel_size, float* truncdist, float* existence) {
    REQ(h);
    return guarded([&] {
        const ObjTSDF* o = h->impl->getObject(id);
        if (!o) throw HipError("object_info: no object " + std::to_string(id), EMF_E_ARG);
        const Vec3i r = o->getVolumeRes();
        if (res) { res[0] = r[0]; res[1] = r[1]; res[2] = r[2]; }
        if (voxel_size) *voxel_size = o->getVoxelSize();
        if (truncdist) *truncdist = o->getTruncDist();
        if (existence) *existence = o->getExProb();
    });
}

int emf_fusion_set_ignore_person(emf_fusion_t* h, int on) {
    REQ(h);
    return guarded([&] { h->impl->setIgnorePerson(on != 0); });
}

int emf_fusion_last_mask_assignment(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count) {
    REQ(h);
    REQ(count);
    return guarded([&] {
        const auto& v = h->impl->lastMaskAssignment();
        *count = static_cast<int32_t>(v.size());
        for (int i = 0; ids && i < capacity && i < static_cast<int>(v.size()); ++i) ids[i] = v[i];
    });
}

int emf_fusion_last_created(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count) {
    REQ(h);
    REQ(count);
    return guarded([&] {
        const auto& v = h->impl->lastCreatedObjects();
        *count = static_cast<int32_t>(v.size());
        for (int i = 0; ids && i < capacity && i < static_cast<int>(v.size()); ++i) ids[i] = v[i];
    });
}

int emf_fusion_match_mask(emf_fusion_t* h, const emf_image_t* mask, int32_t* id, float* iou) {
    REQ(h);
    REQ(mask);
    REQ(id);
    REQ(iou);
    return guarded([&] { *id = h->impl->matchSegmentation(*mask, *iou); });
}

int emf_fusion_update_object(emf_fusion_t* h, int id, const emf_image_t* mask, float offset[3]) {
    REQ(h);
    REQ(mask);
    REQ(offset);
    return guarded([&] {
        const emf::Vec3f o = h->impl->updateObject(id, *mask);
        for (int i = 0; i < 3; ++i) offset[i] = o[i];
    });
}

int emf_fusion_extract_mesh(emf_fusion_t* h, int id, uint32_t* num_vertices, uint32_t* num_triangles) {
    REQ(h);
    REQ(num_vertices);
    REQ(num_triangles);
    return guarded([&] {
        h->mesh = h->impl->getMesh(id);
        *num_vertices = static_cast<uint32_t>(h->mesh.vertices());
        *num_triangles = static_cast<uint32_t>(h->mesh.triangles());
    });
}

int emf_fusion_copy_mesh(emf_fusion_t* h, float* vertices, float* normals, int32_t* triangles) {
    REQ(h);
    return guarded([&] {
        const emf::Mesh& m = h->mesh;
        if (vertices) std::copy(m.cloud.begin(), m.cloud.end(), vertices);
        if (normals) std::copy(m.normals.begin(), m.normals.end(), normals);
        if (triangles) std::copy(m.polygons.begin(), m.polygons.end(), triangles);
    });
}

int emf_io_write_mesh(const char* filename, uint32_t num_vertices, const float* vertices,
                      const float* normals, uint32_t num_triangles, const int32_t* triangles) {
    REQ(filename);
    if (num_vertices) {
        REQ(vertices);
        REQ(normals);
    }
    if (num_triangles) REQ(triangles);
    return guarded([&] {
        emf::Mesh m;
        m.cloud.assign(vertices, vertices + 3 * static_cast<size_t>(num_vertices));
        m.normals.assign(normals, normals + 3 * static_cast<size_t>(num_vertices));
        m.polygons.assign(triangles, triangles + 4 * static_cast<size_t>(num_triangles));
        io::writeMesh(filename, m);
    });
}

int emf_fusion_render(emf_fusion_t* h, uint8_t* rgb, uint8_t* color_map) {
    REQ(h);
    REQ(rgb);
    return guarded([&] {
        h->impl->render(rgb);
        if (color_map) std::copy(h->impl->getColorMap().begin(), h->impl->getColorMap().end(), color_map);
    });
}

int emf_fusion_set_depth_broadcast(emf_fusion_t* h, int root) {
    REQ(h);
    return guarded([&] { h->impl->setDepthBroadcastRoot(root); });
}

int emf_fusion_enable_pose_log(emf_fusion_t* h, int on) {
    REQ(h);
    return guarded([&] { h->impl->enablePoseLog(on != 0); });
}

int emf_fusion_setup_output(emf_fusion_t* h, int exp_frame_meshes, int exp_vols) {
    REQ(h);
    return guarded([&] { h->impl->setupOutput(exp_frame_meshes != 0, exp_vols != 0); });
}

int emf_fusion_write_results(emf_fusion_t* h, const char* dir, int volumes) {
    REQ(h);
    REQ(dir);
    return guarded([&] { h->impl->writeResults(dir, volumes != 0); });
}

int emf_io_write_volume(const char* filename, const float* voxels, const int32_t res[3], float voxel_size) {
    REQ(filename);
    REQ(voxels);
    REQ(res);
    return guarded([&] {
        io::writeVolume(filename, voxels, sizeof(float), Vec3i(res[0], res[1], res[2]), voxel_size);
    });
}

int emf_io_write_pose_file(const char* filename, int n, const int32_t* frames, const float* R,
                           const float* t) {
    REQ(filename);
    if (n > 0) {
        REQ(frames);
        REQ(R);
        REQ(t);
    }
    return guarded([&] {
        std::map<int, Affine3f> poses;
        for (int i = 0; i < n; ++i)
            poses[frames[i]] = Affine3f(m33(R + 9 * i), Vec3f(t[3 * i], t[3 * i + 1], t[3 * i + 2]));
        io::writePoseFile(filename, poses);
    });
}

int emf_io_png_unfilter(const uint8_t* rows, int height, int stride, int bpp, uint8_t* out) {
    REQ(rows);
    REQ(out);
    if (height < 0 || stride <= 0 || (bpp != 1 && bpp != 2)) return EMF_E_ARG;
    // PNG specification 9.2: each scan line is preceded by its filter type; Sub / Average / Paeth
    // predict from the reconstructed byte bpp positions to the left (a), above (b), above-left (c)
    const uint8_t* prev = nullptr;
    for (int y = 0; y < height; ++y) {
        const uint8_t* in = rows + static_cast<size_t>(y) * (stride + 1);
        uint8_t* cur = out + static_cast<size_t>(y) * stride;
        const int f = in[0];
        if (f > 4) return EMF_E_ARG;
        for (int x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = prev ? prev[x] : 0;
            const int c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int pred = 0;
            if (f == 1) pred = a;
            else if (f == 2) pred = b;
            else if (f == 3) pred = (a + b) >> 1;
            else if (f == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[x] = static_cast<uint8_t>(in[1 + x] + pred);
        }
        prev = cur;
    }
    return EMF_OK;
}

int emf_fusion_set_cleanup(emf_fusion_t* h, int on) {
    REQ(h);
    h->cleanUp = on != 0;
    return EMF_OK;
}

int emf_fusion_last_deleted(emf_fusion_t* h, int32_t* ids, int capacity, int32_t* count) {
    REQ(h);
    REQ(count);
    return guarded([&] {
        const auto& v = h->impl->lastDeletedObjects();
        *count = static_cast<int32_t>(v.size());
        for (int i = 0; ids && i < capacity && i < static_cast<int>(v.size()); ++i) ids[i] = v[i];
    });
}

int emf_fusion_set_preprocess(emf_fusion_t* h, int on) {
    REQ(h);
    h->preprocess = on != 0;
    return EMF_OK;
}

int emf_fusion_get_pose(emf_fusion_t* h, int id, float R[9], float t[3]) {
    REQ(h);
    REQ(R);
    REQ(t);
    return guarded([&] {
        Affine3f p;
        if (id == 0) {
            p = h->impl->getCameraPose();
        } else {
            const ObjTSDF* o = h->impl->getObject(id);
            if (!o) throw HipError("emf_fusion_get_pose: no such object on this rank", EMF_E_ARG);
            p = o->getPose();
        }
        for (int k = 0; k < 9; ++k) R[k] = p.rotation().val[k];
        for (int k = 0; k < 3; ++k) t[k] = p.translation()[k];
    });
}

int emf_fusion_track_result(emf_fusion_t* h, int id, int32_t* iterations, int32_t* accepted,
                            int32_t* converged, float* error) {
    REQ(h);
    return guarded([&] {
        const TrackResult* r = h->impl->getTrackResult(id);
        if (!r) throw HipError("emf_fusion_track_result: model was never tracked", EMF_E_ARG);
        if (iterations) *iterations = r->iterations;
        if (accepted) *accepted = r->accepted;
        if (converged) *converged = r->converged ? 1 : 0;
        if (error) *error = r->error;
    });
}

int emf_fusion_stage_estep(emf_fusion_t* h) {
    REQ(h);
    return guarded([&] { h->impl->computeAssociationWeights(); });
}
int emf_fusion_stage_raycast(emf_fusion_t* h) {
    REQ(h);
    return guarded([&] { h->impl->raycast(); });
}
int emf_fusion_stage_integrate(emf_fusion_t* h) {
    REQ(h);
    return guarded([&] { h->impl->integrateDepth(); });
}

int emf_fusion_synchronize(emf_fusion_t* h) {
    REQ(h);
    return guarded([&] { h->impl->synchronize(); });
}

int emf_fusion_enable_timings(emf_fusion_t* h, int on) {
    REQ(h);
    h->impl->enableTimings(on != 0);
    return EMF_OK;
}

int emf_fusion_last_timings(emf_fusion_t* h, emf_frame_timings_t* out) {
    REQ(h);
    REQ(out);
    const FrameTimings& t = h->impl->lastTimings();
    *out = emf_frame_timings_t{t.points, t.estep, t.raycast, t.composite, t.integrate, t.masks,
                               t.total};
    return EMF_OK;
}

int emf_fusion_enable_raycast_stats(emf_fusion_t* h, int on) {
    REQ(h);
    return guarded([&] { h->impl->enableRaycastStats(on != 0); });
}

int emf_fusion_raycast_stats(emf_fusion_t* h, uint64_t counters[4]) {
    REQ(h);
    REQ(counters);
    return guarded([&] {
        const auto c = h->impl->raycastStats();
        for (int i = 0; i < 4; ++i) counters[i] = c[i];
    });
}

int emf_fusion_kernel_timers_enable(emf_fusion_t* h, uint64_t max_launches) {
    REQ(h);
    return guarded([&] {
        h->impl->synchronize();
        h->impl->kernelTimers().enable(static_cast<size_t>(max_launches));
    });
}

int emf_fusion_kernel_timers_clear(emf_fusion_t* h) {
    REQ(h);
    h->impl->kernelTimers().clear();
    return EMF_OK;
}

int emf_fusion_kernel_timers_select(emf_fusion_t* h, uint32_t kind_mask) {
    REQ(h);
    return guarded([&] { h->impl->kernelTimers().select(kind_mask); });
}

int emf_fusion_kernel_timers_stride(emf_fusion_t* h, uint32_t every) {
    REQ(h);
    return guarded([&] { h->impl->kernelTimers().setStride(every); });
}

int emf_fusion_kernel_timers_collect(emf_fusion_t* h, emf_kernel_summary_t out[EMF_K_NUM_KINDS],
                                     uint64_t* dropped) {
    REQ(h);
    REQ(out);
    static_assert(static_cast<int>(EMF_K_NUM_KINDS) == static_cast<int>(KernelTimers::kNumKinds),
                  "kernel kind enums out of sync");
    return guarded([&] {
        h->impl->synchronize();
        const auto s = h->impl->kernelTimers().collect();
        for (int k = 0; k < EMF_K_NUM_KINDS; ++k)
            out[k] = emf_kernel_summary_t{s[k].launches, s[k].total_ms, s[k].units};
        if (dropped) *dropped = h->impl->kernelTimers().droppedLaunches();
    });
}

int emf_fusion_get_image(emf_fusion_t* h, int which, int obj_id, emf_image_t* view) {
    REQ(h);
    REQ(view);
    EMFusion& f = *h->impl;
    auto missing = [&]() {
        std::snprintf(g_err, sizeof(g_err), "get_image: object %d is not held by this rank", obj_id);
        return EMF_E_ARG;
    };
    switch (which) {
        case EMF_IMG_POINTS: *view = f.getPoints().view(); return EMF_OK;
        case EMF_IMG_BG_ASSOC: *view = f.getBgAssociation().view(); return EMF_OK;
        case EMF_IMG_OBJ_ASSOC: {
            const auto* im = f.getObjAssociation(obj_id);
            if (!im) return missing();
            *view = im->view();
            return EMF_OK;
        }
        case EMF_IMG_ASSOC_NORM: *view = f.getAssociationNorm().view(); return EMF_OK;
        case EMF_IMG_RAYLENGTHS: *view = f.getRaylengths().view(); return EMF_OK;
        case EMF_IMG_VERTICES: *view = f.getVertices().view(); return EMF_OK;
        case EMF_IMG_NORMALS: *view = f.getNormals().view(); return EMF_OK;
        case EMF_IMG_SEGMENTATION: *view = f.getModelSegmentation().view(); return EMF_OK;
        case EMF_IMG_BG_RAYLENGTHS: *view = f.getBgRaylengths().view(); return EMF_OK;
        case EMF_IMG_OBJ_RAYLENGTHS: {
            const auto* im = f.getObjRaylengths(obj_id);
            if (!im) return missing();
            *view = im->view();
            return EMF_OK;
        }
        default:
            std::snprintf(g_err, sizeof(g_err), "get_image: unknown selector %d", which);
            return EMF_E_ARG;
    }
}

int emf_fusion_get_volume(emf_fusion_t* h, int which, int obj_id, void** dev_ptr, int32_t res[3]) {
    REQ(h);
    REQ(dev_ptr);
    REQ(res);
    EMFusion& f = *h->impl;
    TSDF* vol = obj_id == 0 ? static_cast<TSDF*>(&f.getBackground()) : f.findObject(obj_id);
    if (!vol) {
        std::snprintf(g_err, sizeof(g_err), "get_volume: object %d is not held by this rank", obj_id);
        return EMF_E_ARG;
    }
    const Vec3i r = vol->getVolumeRes();
    res[0] = r[0];
    res[1] = r[1];
    res[2] = r[2];
    ObjTSDF* obj = obj_id == 0 ? nullptr : static_cast<ObjTSDF*>(vol);
    switch (which) {
        case EMF_VOL_TSDF: *dev_ptr = const_cast<float*>(vol->tsdfPtr()); return EMF_OK;
        case EMF_VOL_WEIGHTS: *dev_ptr = const_cast<float*>(vol->weightsPtr()); return EMF_OK;
        case EMF_VOL_BRICKS:
            *dev_ptr = vol->brickFlagsPtr();
            res[0] = (r[0] + 3) / 4;
            res[1] = (r[1] + 3) / 4;
            res[2] = (r[2] + 3) / 4;
            return EMF_OK;
        case EMF_VOL_FGPROBS:
            if (obj) {
                *dev_ptr = const_cast<float*>(obj->fgProbsPtr());
                return EMF_OK;
            }
            break;
        case EMF_VOL_FGMASK:
            if (obj) {
                *dev_ptr = const_cast<uint8_t*>(obj->fgVolMaskPtr());
                return EMF_OK;
            }
            break;
        default: break;
    }
    std::snprintf(g_err, sizeof(g_err), "get_volume: selector %d not available for id %d", which,
                  obj_id);
    return EMF_E_ARG;
}

int emf_fusion_visible_objects(emf_fusion_t* h, int32_t* ids, int cap, int* n) {
    REQ(h);
    REQ(n);
    int c = 0;
    for (int id : h->impl->visibleObjects()) {
        if (c < cap && ids) ids[c] = id;
        ++c;
    }
    *n = c < cap ? c : cap;
    return EMF_OK;
}

int emf_fusion_object_ids(emf_fusion_t* h, int32_t* ids, int cap, int* n) {
    REQ(h);
    REQ(n);
    int c = 0;
    for (int id : h->impl->objectIds()) {
        if (c < cap && ids) ids[c] = id;
        ++c;
    }
    *n = c < cap ? c : cap;
    return EMF_OK;
}

int emf_fusion_upload_host_time(emf_fusion_t* h, double* seconds, uint64_t* frames) {
    REQ(h);
    REQ(seconds);
    REQ(frames);
    return guarded([&] {
        const auto t = h->impl->uploadHostTime();
        *seconds = t.first;
        *frames = t.second;
    });
}
int emf_fusion_batched_chunks(emf_fusion_t* h) { return h ? h->impl->batchedChunks() : EMF_E_NULL; }
int emf_fusion_background_overlap(emf_fusion_t* h) { return h ? (h->impl->overlapsBackground() ? 1 : 0) : EMF_E_NULL; }

int emf_fusion_frame_index(emf_fusion_t* h) { return h ? h->impl->frameIndex() : EMF_E_NULL; }

int emf_fusion_owns_object(emf_fusion_t* h, int obj_id) {
    return h && h->impl->ownsObject(obj_id) ? 1 : 0;
}

int emf_comm_unique_id(void* out128) {
    REQ(out128);
    return guarded([&] { rcclGetUniqueId(out128); });
}

int emf_comm_create(const void* unique_id128, int rank, int world, emf_comm_t** out) {
    REQ(unique_id128);
    REQ(out);
    return guarded([&] {
        auto c = std::make_unique<emf_comm>();
        c->impl = makeRcclCommunicator(unique_id128, rank, world);
        *out = c.release();
    });
}

void emf_comm_destroy(emf_comm_t* c) { delete c; }

int emf_comm_create_host_staged(const emf_comm_callbacks_t* cb, emf_comm_t** out) {
    REQ(cb);
    REQ(out);
    return guarded([&] {
        HostStagedCallbacks h;
        h.rank = cb->rank;
        h.world = cb->world;
        h.allReduceSumF32 = cb->all_reduce_sum_f32;
        h.allReduceMinU64 = cb->all_reduce_min_u64;
        h.broadcast = cb->broadcast;
        h.user = cb->user;
        auto c = std::make_unique<emf_comm>();
        c->impl = makeHostStagedCommunicator(h);
        *out = c.release();
    });
}

int emf_comm_create_peer_local_group(int world, size_t slot_bytes, emf_comm_t** out) {
    REQ(out);
    return guarded([&] {
        auto group = makePeerCommunicatorsLocal(world, slot_bytes);
        for (int r = 0; r < world; ++r) {
            auto c = std::make_unique<emf_comm>();
            c->impl = group[r];
            out[r] = c.release();
        }
    });
}

int emf_comm_create_peer(int rank, int world, size_t slot_bytes, emf_allgather_fn all_gather, void* user,
                         emf_comm_t** out) {
    REQ(all_gather);
    REQ(out);
    return guarded([&] {
        PeerBootstrap b;
        b.rank = rank;
        b.world = world;
        b.allGather = all_gather;
        b.user = user;
        auto c = std::make_unique<emf_comm>();
        c->impl = makePeerCommunicator(b, slot_bytes);
        *out = c.release();
    });
}

int emf_comm_all_reduce_sum_f32(emf_comm_t* c, float* dev, size_t count, void* stream) {
    REQ(c);
    return guarded([&] {
        Stream s(static_cast<hipStream_t>(stream));
        c->impl->allReduceSumF32(dev, count, s);
    });
}
int emf_comm_all_reduce_min_u64(emf_comm_t* c, uint64_t* dev, size_t count, void* stream) {
    REQ(c);
    return guarded([&] {
        Stream s(static_cast<hipStream_t>(stream));
        c->impl->allReduceMinU64(dev, count, s);
    });
}
int emf_comm_broadcast(emf_comm_t* c, void* dev, size_t bytes, int root, void* stream) {
    REQ(c);
    return guarded([&] {
        Stream s(static_cast<hipStream_t>(stream));
        c->impl->broadcast(dev, bytes, root, s);
    });
}
int emf_comm_gather_row_bands(emf_comm_t* c, void* dev, size_t bytes_per_row, int band_rows, int total_rows,
                              void* stream) {
    REQ(c);
    return guarded([&] {
        Stream s(static_cast<hipStream_t>(stream));
        c->impl->gatherRowBands(dev, bytes_per_row, band_rows, total_rows, s);
    });
}

int emf_comm_create_delayed(emf_comm_t* inner, int microseconds, emf_comm_t** out) {
    REQ(inner);
    REQ(out);
    return guarded([&] {
        auto c = std::make_unique<emf_comm>();
        c->impl = makeDelayedCommunicator(inner->impl, microseconds);
        *out = c.release();
    });
}

int emf_comm_describe(emf_comm_t* c, char* json, size_t cap) {
    REQ(c);
    REQ(json);
    return guarded([&] {
        const std::string s = c->impl->describe();
        if (s.size() + 1 > cap) throw HipError("emf_comm_describe: buffer too small", EMF_E_ARG);
        std::memcpy(json, s.c_str(), s.size() + 1);
    });
}

int emf_comm_exchanges(emf_comm_t* c, uint64_t* out) {
    REQ(c);
    REQ(out);
    return guarded([&] { *out = c->impl->exchangesIssued(); });
}

int emf_comm_create_local_group(int world, emf_comm_t** out) {
    REQ(out);
    return guarded([&] {
        auto group = makeLocalCommunicators(world);
        for (int r = 0; r < world; ++r) {
            auto c = std::make_unique<emf_comm>();
            c->impl = group[r];
            out[r] = c.release();
        }
    });
}

int emf_synth_create(int width, int height, const float K[9], int num_spheres, uint64_t seed,
                     float noise_sigma, float dropout, emf_synth_t** out) {
    REQ(K);
    REQ(out);
    return guarded([&] {
        auto s = std::make_unique<emf_synth>();
        s->impl = std::make_unique<SyntheticScene>(Size(width, height), m33(K), num_spheres, seed,
                                                   noise_sigma, dropout);
        *out = s.release();
    });
}

void emf_synth_destroy(emf_synth_t* s) { delete s; }

int emf_synth_render(emf_synth_t* s, int frame, float* depth, uint8_t* ids) {
    REQ(s);
    REQ(depth);
    return guarded([&] { s->impl->render(frame, depth, ids); });
}

int emf_synth_camera_pose(emf_synth_t* s, int frame, float R[9], float t[3]) {
    REQ(s);
    REQ(R);
    REQ(t);
    const Affine3f p = s->impl->cameraPose(frame);
    std::memcpy(R, p.rotation().val, 9 * sizeof(float));
    std::memcpy(t, p.translation().val, 3 * sizeof(float));
    return EMF_OK;
}

int emf_synth_sphere(emf_synth_t* s, int k, int frame, float center[3], float* radius,
                     float* volume_size) {
    REQ(s);
    if (k < 0 || k >= s->impl->numSpheres()) {
        std::snprintf(g_err, sizeof(g_err), "emf_synth_sphere: index %d out of range", k);
        return EMF_E_ARG;
    }
    const Vec3f c = s->impl->sphereCenter(k, frame);
    if (center) std::memcpy(center, c.val, 3 * sizeof(float));
    if (radius) *radius = s->impl->sphere(k).radius;
    if (volume_size) *volume_size = s->impl->objectVolumeSize(k);
    return EMF_OK;
}

}  // extern "C"
