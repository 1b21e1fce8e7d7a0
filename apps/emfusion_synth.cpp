// emfusion_synth.cpp -- the reference's main loop (apps/EM-Fusion.cpp:139-156: read a frame,
// emf.processFrame(frame)) on the MI355X-native classes, fed by the deterministic synthetic RGB-D
// stream instead of a dataset reader.  Prints frames/s and per-stage GPU milliseconds.
//
//   emfusion_synth [--frames N] [--objects K] [--bg-res R] [--obj-res R] [--width W --height H]
//                  [--materialize-gradients] [--autonomous] [--out DIR]
//   emfusion_synth --sequence DIR/ [--masks DIR] [--mask-frames N] [--visibility-thresh N] [--frames N]
//                  [--bg-res R] [--bg-voxel M] [--obj-res R] [--volumes] --out DIR
//   emfusion_synth --dir BASE/ [--colordir colour] [--depthdir depth] [--intrinsics fx fy cx cy] ... --out DIR
// --configfile FILE (-c): with --sequence / --dir, take every parameter from one of the reference's configuration files
// (config/default.cfg, tum.cfg ...; core/Config.hpp) instead of the sizing options above.
// --dir: the same loop on a Co-Fusion style dataset (ColorNNNN.png + DepthNNNN.exr), the reference's ImageReader
// (apps/EM-Fusion.cpp:118-126; core/Readers.hpp ImageReader + readExr).
// --sequence: the reference's loop itself (apps/EM-Fusion.cpp:100-156) on a TUM RGB-D sequence: TUMRGBDReader
// (core/Readers.hpp) -> emf.usePreprocMasks(masks) -> processFrame(frame) with camera and object tracking from the
// second frame on -> writeResults.  What apps/run_tum.py does from Python, without Python.
// --out DIR: keep the pose log and write the reference's result files at the end (writeResults:
// poses-*.txt, mesh_*.ply, tsdfs/*.bin; EMFusion.cpp:258-292) into the existing directory DIR.
// --autonomous: nothing but depth and instance masks go in, as in the reference's own loop -- objects
// are spawned from the masks of frame 0 (initNewObjVolume), camera and object poses are tracked
// (performTracking), later masks are matched to the models (matchSegmentation); the ground-truth
// poses of the stream are only used to report the tracking error at the end.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "Config.hpp"
#include "EMFusion.hpp"
#include "Readers.hpp"
#include "SyntheticScene.hpp"

// The reference's main loop on a dataset (apps/EM-Fusion.cpp:100-156): a TUM sequence (`--sequence`, TUMRGBDReader) or a
// Co-Fusion style directory (`--dir`, ImageReader: ColorNNNN.png + DepthNNNN.exr), as apps/EM-Fusion.cpp:118-131 chooses
static int runSequence(const std::string& seq, bool cofusion, const std::string& colordir, const std::string& depthdir,
                       const float* intrinsics, const std::string& configFile, const std::string& masks,
                       const std::string& outDir, int frames, int bgRes, float bgVoxel, int objRes, int maskFrames,
                       int visibilityThresh, bool volumes) {
    std::unique_ptr<emf::TUMRGBDReader> tum;
    std::unique_ptr<emf::ImageReader> dir;
    size_t available = 0;
    if (cofusion) {
        dir.reset(new emf::ImageReader(seq, colordir, depthdir));  // "<base><colordir>", "<base><depthdir>"
        available = dir->getNumFrames();
    } else {
        tum.reset(new emf::TUMRGBDReader(seq));  // "<dir>/associations.txt"
        available = tum->getNumFrames();
    }
    if (available == 0) throw std::runtime_error("no frames in " + seq);
    const size_t n = frames > 0 ? std::min<size_t>(frames, available) : available;
    std::vector<float> depth;
    auto readDepth = [&](size_t f) {
        return cofusion ? dir->readDepth(dir->firstIndex() + static_cast<int>(f), depth) : tum->readDepth(f, depth);
    };
    const emf::Size size = readDepth(0);
    emf::Params params;  // reference defaults (config/default.cfg)
    const bool configured = !configFile.empty();
    if (configured) {
        // --configfile: everything comes from the reference's configuration file (apps/EM-Fusion.cpp:268-371) and, for
        // --dir, from <base>calibration.txt (:399-410); the sizing options of this command line are not applied
        emf::loadConfigFile(params, configFile);
        if (cofusion) emf::loadCalibrationFile(params, seq + "calibration.txt");
        if (params.frameSize.width != size.width || params.frameSize.height != size.height)
            throw std::runtime_error("the configuration says " + std::to_string(params.frameSize.width) + " x " +
                                     std::to_string(params.frameSize.height) + ", the images are " + std::to_string(size.width) +
                                     " x " + std::to_string(size.height));
    } else {
        params.frameSize = size;
        params.setDefaultIntrinsics();
    }
    if (intrinsics) {  // the reference takes them from its config file (data.h: intr)
        params.intr = emf::Matx33f::eye();
        params.intr(0, 0) = intrinsics[0];
        params.intr(1, 1) = intrinsics[1];
        params.intr(0, 2) = intrinsics[2];
        params.intr(1, 2) = intrinsics[3];
    }
    if (!configured) {
        params.globalVolumeDims = emf::Vec3i::all(bgRes);
        params.globalVoxelSize = bgVoxel;
        params.volumePose = emf::Affine3f(emf::Matx33f::eye(), emf::Vec3f(0.f, 0.f, bgRes * bgVoxel / 2.f));
        params.objVolumeDims = emf::Vec3i::all(objRes);
        const float scale = static_cast<float>(size.width) / 640.f;
        params.visibilityThresh = visibilityThresh > 0 ? visibilityThresh : static_cast<int>(std::lround(1600 * scale * scale));
        params.boundary = static_cast<int>(std::lround(20 * scale));
        params.maskRCNNFrames = maskFrames;
    }
    emf::EMFusion emf(params);
    if (!masks.empty()) emf.usePreprocMasks(masks);   // apps/EM-Fusion.cpp:115
    emf.setupOutput(false, volumes);                  // apps/EM-Fusion.cpp:112
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t f = 0; f < n; ++f) {                  // while (reader->moreFrames())
        readDepth(f);                                 // frame = reader->getNextFrame()
        for (float& d : depth)
            if (!std::isfinite(d)) d = 0.f;
        emf::FrameInputs in;                          // frame 0 defines the world frame; then everything is tracked
        in.trackCamera = in.trackObjects = f > 0;
        in.cleanUp = true;
        emf.setFrameInputs(in);
        emf::RGBD frame;
        frame.size = size;
        frame.depth = depth.data();
        emf.processFrame(frame);                      // apps/EM-Fusion.cpp:152
        if (f % 50 == 0) {
            std::vector<uint8_t> maskim;
            const int inst = emf.getLastMasks(maskim);  // apps/EM-Fusion.cpp:162
            std::printf("frame %zu/%zu: %zu visible objects, %d instances in the last mask frame\n", f, n,
                        emf.visibleObjects().size(), inst);
        }
    }
    emf.synchronize();
    emf.writeResults(outDir, volumes);                // apps/EM-Fusion.cpp:204
    std::printf("%zu frames of %s (%.1f Hz) in %.1f s incl. image decoding on the host; results in %s\n", n, seq.c_str(),
                cofusion ? 30.0 : tum->getFrameRate(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                outDir.c_str());
    return 0;
}

int main(int argc, char** argv) {
    int frames = 120, objects = 4, bgRes = 512, objRes = 128, width = 640, height = 480;
    bool materialize = false, autonomous = false;
    std::string outDir, sequence, maskDir, dataDir, colordir = "colour", depthdir = "depth", configFile;
    float intrinsics[4] = {0.f, 0.f, 0.f, 0.f};
    bool haveIntrinsics = false;
    int maskFrames = 30, visThresh = 0, framesGiven = 0;
    float bgVoxel = 0.f;
    bool volumes = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? std::atoi(argv[++i]) : 0; };
        if (a == "--frames") frames = framesGiven = next();
        else if (a == "--sequence" && i + 1 < argc) sequence = argv[++i];
        else if (a == "--masks" && i + 1 < argc) maskDir = argv[++i];
        else if (a == "--dir" && i + 1 < argc) dataDir = argv[++i];
        else if ((a == "--configfile" || a == "-c") && i + 1 < argc) configFile = argv[++i];
        else if (a == "--colordir" && i + 1 < argc) colordir = argv[++i];
        else if (a == "--depthdir" && i + 1 < argc) depthdir = argv[++i];
        else if (a == "--intrinsics" && i + 4 < argc) {
            for (int k = 0; k < 4; ++k) intrinsics[k] = static_cast<float>(std::atof(argv[++i]));
            haveIntrinsics = true;
        }
        else if (a == "--mask-frames") maskFrames = next();
        else if (a == "--visibility-thresh") visThresh = next();
        else if (a == "--bg-voxel" && i + 1 < argc) bgVoxel = static_cast<float>(std::atof(argv[++i]));
        else if (a == "--volumes") volumes = true;
        else if (a == "--objects") objects = next();
        else if (a == "--bg-res") bgRes = next();
        else if (a == "--obj-res") objRes = next();
        else if (a == "--width") width = next();
        else if (a == "--height") height = next();
        else if (a == "--materialize-gradients") materialize = true;
        else if (a == "--autonomous") autonomous = true;
        else if (a == "--out" && i + 1 < argc) outDir = argv[++i];
        else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    try {
        if (!sequence.empty() || !dataDir.empty()) {
            if (outDir.empty()) throw std::runtime_error("--sequence / --dir need --out DIR");
            const bool cofusion = !dataDir.empty();
            return runSequence(cofusion ? dataDir : sequence, cofusion, colordir, depthdir, haveIntrinsics ? intrinsics : nullptr,
                               configFile, maskDir, outDir, framesGiven, bgRes, bgVoxel > 0 ? bgVoxel : 5.12f / static_cast<float>(bgRes),
                               objRes, maskFrames, visThresh, volumes);
        }
        emf::Params params;  // reference defaults (config/default.cfg)
        params.frameSize = emf::Size(width, height);
        params.setDefaultIntrinsics();
        params.globalVolumeDims = emf::Vec3i::all(bgRes);
        params.globalVoxelSize = 5.12f / static_cast<float>(bgRes);
        params.objVolumeDims = emf::Vec3i::all(objRes);
        const float scale = static_cast<float>(width) / 640.f;
        params.visibilityThresh = static_cast<int>(1600 * scale * scale);
        params.boundary = static_cast<int>(20 * scale);

        emf::SyntheticScene scene(params.frameSize, params.intr, objects);
        emf::EMFusion emf(params, materialize ? emf::TSDF::Gradients::Materialized
                                              : emf::TSDF::Gradients::OnTheFly);
        std::vector<int> ids;
        if (!autonomous)
            for (int k = 0; k < objects; ++k)
                ids.push_back(emf.addObject(scene.sphereCenter(k, 0), scene.objectVolumeSize(k)));

        const size_t P = params.frameSize.area();
        std::vector<float> depth(P);
        std::vector<uint8_t> sid(P), mask(P);
        std::vector<emf::DeviceImage<uint8_t>> maskDev;
        for (int k = 0; k < objects; ++k) maskDev.emplace_back(params.frameSize);
        emf.enableTimings(true);
        if (!outDir.empty()) emf.setupOutput(false, true);  // apps/EM-Fusion.cpp:112

        double gpuMs = 0;
        int spawned = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; ++f) {  // while (reader->moreFrames())
            scene.render(f, depth.data(), sid.data());  // frame = reader->getNextFrame()
            emf::FrameInputs in;
            in.cam_pose = scene.cameraPose(f);
            in.runMasks = f % params.maskRCNNFrames == 0;
            if (in.runMasks)
                for (int k = 0; k < objects; ++k) {
                    for (size_t i = 0; i < P; ++i) mask[i] = sid[i] == k + 1 ? 1 : 0;
                    maskDev[k].upload(mask.data(), emf.mainStream());
                    emf.mainStream().waitForCompletion();  // host buffer is reused
                }
            if (!autonomous) {
                for (int k = 0; k < objects; ++k)
                    in.obj_poses[ids[k]] = emf::Affine3f(emf::Matx33f::eye(), scene.sphereCenter(k, f));
                if (in.runMasks)
                    for (int k = 0; k < objects; ++k) in.masks[ids[k]] = maskDev[k].view();
            } else {
                in.trackCamera = in.trackObjects = f > 0;
                in.cleanUp = true;
                // a "Mask R-CNN frame": all instance masks go through initOrMatchObjs inside the
                // frame (match / spawn / existence bookkeeping), then integrateMasks, cleanUpObjs
                if (in.runMasks)
                    for (int k = 0; k < objects; ++k) in.instanceMasks.push_back(maskDev[k].view());
                in.runMasks = false;
            }
            emf.setFrameInputs(in);
            emf::RGBD frame;
            frame.size = params.frameSize;
            frame.depth = depth.data();
            emf.processFrame(frame);  // reference EMFusion.cpp:70
            gpuMs += emf.lastTimings().total;
            if (autonomous)
                for (int id : emf.lastCreatedObjects()) spawned += id >= 0;
        }
        emf.synchronize();
        if (autonomous) {
            const emf::Vec3f d = emf.getCameraPose().translation() - scene.cameraPose(frames - 1).translation();
            std::printf("autonomous: %d objects spawned from masks; camera position error after %d "
                        "tracked frames: %.1f mm",
                        spawned, frames - 1,
                        1e3 * std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
            if (const emf::TrackResult* r = emf.getTrackResult(0))
                std::printf(" (last frame: %d LM steps, %d accepted)", r->iterations, r->accepted);
            std::printf("\n");
        }
        const double wall =
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const emf::FrameTimings& t = emf.lastTimings();
        std::printf("%d frames, %d objects, bg %d^3, obj %d^3, %dx%d: %.1f frames/s of GPU time "
                    "(%.3f ms/frame), %.1f frames/s wall incl. host rendering of the stream\n",
                    frames, objects, bgRes, objRes, width, height, 1e3 * frames / gpuMs,
                    gpuMs / frames, frames / wall);
        std::printf("last frame [ms]: points %.3f  estep(x3) %.3f  raycast %.3f  composite %.3f  "
                    "integrate %.3f  masks %.3f | visible objects %zu | batched launches: %s\n",
                    t.points, t.estep, t.raycast, t.composite, t.integrate, t.masks,
                    emf.visibleObjects().size(), emf.usesBatchedLaunches() ? "yes" : "no");
        if (!outDir.empty()) {
            emf.writeResults(outDir, false);  // volumes follow setupOutput, as in the reference
            const emf::Mesh bg = emf.getMesh(0);
            std::printf("results in %s: background mesh %zu vertices, %zu triangles\n", outDir.c_str(),
                        bg.vertices(), bg.triangles());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emfusion_synth: %s\n", e.what());
        return 1;
    }
    return 0;
}
