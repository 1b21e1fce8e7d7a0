// Output.hpp -- the reference's on-disk result formats: PLY meshes
// (EMFusion::writeMesh, EMFusion.cpp:1263-1300),
// raw volume dumps (EMFusion::writeVolume, EMFusion.cpp:1302-1313) and TUM-style pose files
// (EMFusion::writePoseFile / writePoses, EMFusion.cpp:991-1007, 1238-1254).  Host code only.
#pragma once

#include <array>
#include <map>
#include <string>
#include <vector>

#include "types.hpp"

namespace emf {
namespace io {

/**
 * "<name>.bin": int32 resolution[3], size_t element size (8 bytes), float voxel size, then the
 * voxels, x fastest -- byte for byte what the reference writes, so its evaluation scripts read it.
 */
void writeVolume(const std::string& filename, const void* voxels, size_t elemSize,
                 const Vec3i& resolution, float voxelSize);
/** Inverse of writeVolume (float volumes); throws std::runtime_error on a malformed file. */
std::vector<float> readVolume(const std::string& filename, Vec3i& resolution, float& voxelSize);

/** Unit quaternion (x, y, z, w) of a rotation matrix, Eigen's Quaternion(Matrix3) algorithm. */
void rotationToQuaternion(const Matx33f& R, float q[4]);

/**
 * One line per frame: "frame tx ty tz qx qy qz qw", default ostream formatting (6 significant
 * digits), frames in ascending order -- the TUM trajectory format of the reference's pose files.
 */
void writePoseFile(const std::string& filename, const std::map<int, Affine3f>& poses);

/**
 * The label colours of the renderings (reference EMFusion::randomColors, EMFusion.cpp:614-633): 255
 * fully saturated hues, converted with cv::cvtColor(COLOR_HSV2RGB), shuffled with
 * cv::randShuffle(cv::RNG(6893)), label 0 white.  256 x RGB.  The two OpenCV routines are third-party
 * code outside the reference tree, restated from their published algorithms (parity unpinned); the
 * reference leaves hsv[0] uninitialised before the shuffle, it is black here.
 */
std::array<uint8_t, 768> randomColors();

/**
 * ASCII PLY as the reference writes it (EMFusion::writeMesh, EMFusion.cpp:1263-1300): vertices
 * "x y z nx ny nz" with %f, faces "3 i0 i1 i2".
 */
void writeMesh(const std::string& filename, const Mesh& mesh);

/**
 * cv::Mat::convertTo(CV_8U, 255) of a float image, the form in which the reference keeps its per-frame debug
 * images (storeAssocs EMFusion.cpp:307-320; TSDF::getHuberWeights / getTrackingWeights TSDF.cpp:346-354;
 * ObjTSDF::getFgProbVals ObjTSDF.cpp:237-240): saturate_cast<uchar>(v * 255) = round to nearest, ties to even,
 * clamped to 0..255 (NaN -> 0).  `pitchFloats` = floats per source row.
 */
std::vector<uint8_t> toU8Times255(const float* src, int width, int height, size_t pitchFloats);

/**
 * An 8-bit PNG (channels = 1: grayscale, 3: RGB) as cv::imwrite would produce one for the reference's
 * EMFusion::writeImage (EMFusion.cpp:1256-1261) -- same pixels; the compressed bytes are zlib's, not libpng's.
 * encodePng returns the file's bytes (the per-frame log keeps those, not the raw images).
 */
std::vector<uint8_t> encodePng(const uint8_t* pixels, int width, int height, int channels);
void writeBytes(const std::string& filename, const std::vector<uint8_t>& bytes);
/** "<dir>/%04d.png" for every entry (EMFusion::writeImage's file names); creates <dir> and its parents. */
void writeImageLog(const std::string& dir, const std::map<int, std::vector<uint8_t>>& pngByFrame);
/** mkdir -p */
void createDirectories(const std::string& dir);

}  // namespace io
}  // namespace emf
