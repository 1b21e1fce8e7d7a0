// EMFusionCapture.cpp -- emf::EMFusion: results and debug captures (reference src/core/EMFusion.cpp:131-160, 243-327, 991-1236).
#include "EMFusion.hpp"
#include "EMFusionDetail.hpp"
#include "Output.hpp"

#include <sys/stat.h>

#include <cerrno>
#include <cmath>
#include <stdexcept>

namespace emf {

using namespace detail;

void EMFusion::writeResults(const std::string& dir, bool volumes) {
    synchronize();
    // boost::filesystem::create_directories(p) (EMFusion.cpp:254-255): every missing component of the path
    for (size_t k = 1; k <= dir.size(); ++k)
        if (k == dir.size() || dir[k] == '/') {
            const std::string part = dir.substr(0, k);
            if (!part.empty() && mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
                throw std::runtime_error("EMFusion::writeResults: cannot create " + part);
        }
    io::writePoseFile(dir + "/poses-cam.txt", poses);
    for (const auto& op : obj_poses)
        io::writePoseFile(dir + "/poses-" + std::to_string(op.first) + ".txt", op.second);
    for (const auto& op : addPoseOffsets(obj_poses, obj_pose_offsets))  // EMFusion.cpp:1000-1006
        io::writePoseFile(dir + "/poses-" + std::to_string(op.first) + "-corrected.txt", op.second);
    // writeMeshes (EMFusion.cpp:1147-1156) runs whether or not volumes are exported: the background,
    // the live objects, and the objects that were deleted while the log was on (their last mesh,
    // EMFusion.cpp:966)
    io::writeMesh(dir + "/mesh_bg.ply", background.getMesh());
    for (auto& obj : objects)
        if (!(ignorePerson && isPerson(obj))) meshes[obj.getID()] = obj.getMesh();
    for (const auto& m : meshes) io::writeMesh(dir + "/mesh_" + std::to_string(m.first) + ".ply", m.second);
    // writeRenderings / writeAssocs / writeHuberWeights / writeTrackWeights / writeFgProbs (EMFusion.cpp:1009-1145):
    // directories are created whether or not the log holds anything, like the reference's
    io::writeImageLog(dir + "/output", renderings);
    io::writeImageLog(dir + "/assoc_weights/bg/preTrack", bg_assocWeight_preTrack);
    io::writeImageLog(dir + "/assoc_weights/bg/postTrack", bg_assocWeight_postTrack);
    for (const auto& o : obj_assocWeights_preTrack)
        io::writeImageLog(dir + "/assoc_weights/" + std::to_string(o.first) + "/preTrack", o.second);
    for (const auto& o : obj_assocWeights_postTrack)
        io::writeImageLog(dir + "/assoc_weights/" + std::to_string(o.first) + "/postTrack", o.second);
    io::writeImageLog(dir + "/huber_weights/bg", bg_huberWeights);
    for (const auto& o : obj_huberWeights) io::writeImageLog(dir + "/huber_weights/" + std::to_string(o.first), o.second);
    io::writeImageLog(dir + "/track_weights/bg", bg_trackWeights);
    for (const auto& o : obj_trackWeights) io::writeImageLog(dir + "/track_weights/" + std::to_string(o.first), o.second);
    io::createDirectories(dir + "/fg_probs");
    for (const auto& o : obj_fgProbs) io::writeImageLog(dir + "/fg_probs/" + std::to_string(o.first), o.second);
    if (!(volumes || expVols)) return;  // `if ( expVols ) writeTSDFs ( p )` (EMFusion.cpp:290-291)
    const std::string t = dir + "/tsdfs";
    if (mkdir(t.c_str(), 0777) != 0 && errno != EEXIST)
        throw std::runtime_error("EMFusion::writeResults: cannot create " + t);
    auto dump = [&](const std::string& name, const std::vector<float>& v, const Vec3i& res, float vox) {
        io::writeVolume(t + "/" + name + ".bin", v.data(), sizeof(float), res, vox);
    };
    dump("bg_tsdf", background.getTSDF(), background.getVolumeRes(), background.getVoxelSize());
    for (auto& obj : objects) {
        if (ignorePerson && isPerson(obj)) continue;  // the same `continue` skips them (EMFusion.cpp:274-277)
        savedVolumes[obj.getID()] = saveVolumes(obj);
    }
    for (const auto& sv : savedVolumes) {  // writeTSDFs (EMFusion.cpp:1195-1216): live and deleted objects
        const std::string id = std::to_string(sv.first);
        dump("tsdf_" + id, sv.second.tsdf, sv.second.res, sv.second.voxelSize);
        dump("weights_" + id, sv.second.weights, sv.second.res, sv.second.voxelSize);
        dump("fgProbs_" + id, sv.second.fgProbs, sv.second.res, sv.second.voxelSize);
    }
}

EMFusion::SavedVolumes EMFusion::saveVolumes(ObjTSDF& obj) {  // EMFusion.cpp:279-285, 967-973
    SavedVolumes sv;
    sv.tsdf = obj.getTSDF();
    sv.weights = obj.getWeightsVol();
    sv.fgProbs = obj.getFgProbVol();
    sv.res = obj.getVolumeRes();
    sv.voxelSize = obj.getVoxelSize();
    return sv;
}

// Reference EMFusion::addPoseOffsets (EMFusion.cpp:1220-1236): undo the accumulated centre shifts so
// that the trajectory refers to the volume centre the object was created with.
std::map<int, std::map<int, Affine3f>> EMFusion::addPoseOffsets(
    const std::map<int, std::map<int, Affine3f>>& all,
    const std::map<int, std::map<int, Vec3f>>& offsets) {
    std::map<int, std::map<int, Affine3f>> cleaned;
    for (const auto& op : all) {
        Vec3f cum = Vec3f::all(0.f);
        const auto off = offsets.find(op.first);
        for (const auto& fp : op.second) {
            if (off != offsets.end()) {
                const auto o = off->second.find(fp.first);
                if (o != off->second.end()) cum = cum - o->second;
            }
            cleaned[op.first][fp.first] = fp.second.translate(fp.second.rotation() * cum);
        }
    }
    return cleaned;
}

void EMFusion::render(uint8_t* rgb) {
    if (sharded)
        throw HipError("EMFusion::render: not available on the sharded path (vertices / normals of "
                       "remote objects and background bands stay on their ranks)", EMF_E_ARG);
    const size_t bytes = static_cast<size_t>(params.frameSize.area()) * 3;
    if (frameCount < 1) {
        std::fill(rgb, rgb + bytes, uint8_t{0});
        return;
    }
    if (frameCount == 1) raycast();  // frame 0 ran without one (EMFusion.cpp:135-137)
    if (image.empty()) image = DeviceImage<uint8_t, 3>(params.frameSize);
    const emf_image_t vv = vertices.view(), nv = normals.view(), sv = modelSegmentation.view(),
                      iv = image.view();
    if (ignorePerson) {  // EMFusion.cpp:139-150: in place, like the reference
        const emf_image_t bv = bg_vertices.view(), bn = bg_normals.view();
        for (const auto& obj : objects)
            if (isPerson(obj))
                emfCheck(emf_hip_hideLabel(&sv, obj.getID(), &vv, &nv, &bv, &bn, main.abi()), "hideLabel");
    }
    const float light[3] = {0.f, 0.f, 0.f};  // cv::Affine3f::Identity()
    emfCheck(emf_hip_renderPhong(&vv, &nv, &sv, colorMap.data(), light, &iv, main.abi()), "renderPhong");
    hipCheck(hipMemcpyAsync(rgb, image.ptr(), bytes, hipMemcpyDeviceToHost, main.get()), "render D2H");
    main.waitForCompletion();
    if (saveOutput)  // `rendered.copyTo ( renderings[frameCount-1] )`, EMFusion.cpp:158-160
        renderings[frameCount - 1] = io::encodePng(rgb, params.frameSize.width, params.frameSize.height, 3);
}

// ---- per-frame debug images (reference saveOutput mode) ---------------------------------------------------

std::vector<uint8_t> EMFusion::pngOf(const float* dev, size_t pitchBytes) {
    const int w = params.frameSize.width, h = params.frameSize.height;
    std::vector<float> host(static_cast<size_t>(w) * h);
    hipCheck(hipMemcpy2DAsync(host.data(), static_cast<size_t>(w) * sizeof(float), dev, pitchBytes,
                              static_cast<size_t>(w) * sizeof(float), static_cast<size_t>(h), hipMemcpyDeviceToHost,
                              main.get()),
             "hipMemcpy2DAsync(debug image)");
    main.waitForCompletion();
    const std::vector<uint8_t> u8 = io::toU8Times255(host.data(), w, h, static_cast<size_t>(w));
    return io::encodePng(u8.data(), w, h, 1);
}

void EMFusion::storeAssocs(ImageLog& bg, std::map<int, ImageLog>& objs) {
    if (sharded) return;  // (remote objects' maps are not on this rank; the reference is single-GPU)
    const emf_image_t b = bg_associationWeights.view();
    bg[frameCount] = pngOf(static_cast<const float*>(b.data), b.pitch);
    for (const auto& obj : objects) {
        const emf_image_t a = objImages.at(obj.getID()).associationWeights.view();
        objs[obj.getID()][frameCount] = pngOf(static_cast<const float*>(a.data), a.pitch);
    }
}

void EMFusion::storeTrackWeights(int first, int count) {
    if (count <= 0 || trackStates.empty()) return;
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t px = static_cast<size_t>(w) * h, per = emf_hip_trackScratchBytes(w, h);
    if (logScratch.bytes() < 2 * px * sizeof(float) * count) logScratch = DeviceBuffer(2 * px * sizeof(float) * count);
    emf_track_params_t tp;
    tp.huberThresh = params.tsdfParams.huberThresh;
    tp.maxWeight = params.tsdfParams.maxTSDFWeight;
    tp.tau = params.tsdfParams.tau;
    tp.eps1 = params.tsdfParams.eps1;
    tp.eps2 = params.tsdfParams.eps2;
    tp.nuInit = params.tsdfParams.nu_init;
    const emf_image_t pv = points.view();
    float* huber = logScratch.as<float>();
    float* track = huber + px * count;
    // the stage's states are final and the models' association maps are still the ones it tracked with
    forChunks(first, first + count, [&](int from, int cnt) {  // (a stage of more than EMF_MAX_BATCH models ran chunk by chunk too)
        emfCheck(emf_hip_trackWeightImages(currentTable() + from, trackStates.as<emf_track_state_t>() + from, cnt, &pv, &tp,
                                           static_cast<const char*>(trackScratch.data()) + per * from, per,
                                           huber + px * (from - first), track + px * (from - first), main.abi()),
                 "trackWeightImages");
    });
    auto it = objects.begin();
    for (int m = 0; m < count; ++m) {
        const std::vector<uint8_t> hp = pngOf(huber + px * m, static_cast<size_t>(w) * sizeof(float));
        const std::vector<uint8_t> tpng = pngOf(track + px * m, static_cast<size_t>(w) * sizeof(float));
        if (first + m == 0) {
            bg_huberWeights[frameCount] = hp;
            bg_trackWeights[frameCount] = tpng;
        } else {
            const int id = (it++)->getID();
            obj_huberWeights[id][frameCount] = hp;
            obj_trackWeights[id][frameCount] = tpng;
        }
    }
}

void EMFusion::storeFgProbs() {
    if (sharded || objects.empty()) return;
    const int w = params.frameSize.width, h = params.frameSize.height;
    const size_t px = static_cast<size_t>(w) * h;
    if (logScratch.bytes() < px * sizeof(float)) logScratch = DeviceBuffer(px * sizeof(float));
    const emf_image_t pv = points.view();
    const emf_image_t out{logScratch.data(), static_cast<size_t>(w) * sizeof(float), w, h};
    for (auto& obj : objects) {
        // cuda::TSDF::getVolumeVals ( fgProbs, points, rel_pose_CO ... fgProbVals ), ObjTSDF.cpp:189-191
        const Affine3f co = obj.getPose().inv() * pose;
        const Vec3i res = obj.getVolumeRes();
        const int32_t r[3] = {res[0], res[1], res[2]};
        emfCheck(emf_hip_getVolumeVals(obj.fgProbsPtr(), 1, &pv, co.rotation().val, co.translation().val, r,
                                       obj.getVoxelSize(), &out, main.abi()),
                 "getVolumeVals(fgProbs)");
        obj_fgProbs[obj.getID()][frameCount] = pngOf(logScratch.as<float>(), out.pitch);
    }
}

Mesh EMFusion::getMesh(int id) {
    synchronize();
    if (id == 0) return background.getMesh();
    for (auto& o : objects)
        if (o.getID() == id) return o.getMesh();
    throw HipError("EMFusion::getMesh: no object " + std::to_string(id) + " on this rank", EMF_E_ARG);
}

}  // namespace emf
