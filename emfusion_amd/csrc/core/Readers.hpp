// Readers.hpp -- dataset input for the reference's main loop (apps/EM-Fusion.cpp:100-156) in C++:
// the TUM RGB-D sequence reader (reference src/utils/TUMRGBDReader.cpp, include/EMFusion/utils/TUMRGBDReader.h)
// and the preprocessed Mask R-CNN results that EMFusion::usePreprocMasks points at (reference
// src/core/MaskRCNN.cpp:250-282 loadPreprocessed, which calls back into apps/maskrcnn.in.py:258-268 to unpickle
// them; files written by its `preprocess`: pickle.dump((boxes, masks, scores), f, HIGHEST_PROTOCOL)).
// No OpenCV, Boost or Python here: a grayscale PNG decoder over zlib and a small unpickler that understands what
// those files hold (lists of numbers, numpy arrays).  Host-only code.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "types.hpp"

namespace emf {

/** Non-interlaced 8- or 16-bit grayscale PNG -> pixel values, row-major.  Throws std::runtime_error. */
void readPngGray(const std::string& path, std::vector<uint16_t>& pixels, int& width, int& height);

/** Keeps the reference's class name and the parts of its surface the main loop uses. */
class TUMRGBDReader {
public:
    /** path: the sequence directory with a trailing separator, as the reference takes it ("<dir>/associations.txt"). */
    explicit TUMRGBDReader(std::string path);
    /** lines `t1 file1 t2 file2`; which of the two files is the colour image is decided by the first line */
    static void readFileAssociations(const std::string& filename, std::vector<std::string>& rgbNames,
                                     std::vector<std::string>& depthNames, std::vector<double>* stamps = nullptr);
    size_t getNumFrames() const { return depthFileNames.size(); }
    double getFrameRate() const { return frameRate; }
    const std::string& depthFileName(size_t i) const { return depthFileNames[i]; }
    /** depth of frame i in metres (16-bit PNG / 5000, TUMRGBDReader.cpp readFrame); returns its size */
    Size readDepth(size_t i, std::vector<float>& depth) const;

private:
    std::string path;
    std::vector<std::string> rgbFileNames, depthFileNames;
    double frameRate = 0.0;
};

/**
 * One channel of a single-part scan-line OpenEXR file as float, row-major: what cv::imread(IMREAD_UNCHANGED) hands the
 * reference for the Co-Fusion depth files (reference src/utils/ImageReader.cpp:105-110).  Compression NONE, RLE, ZIPS
 * and ZIP; HALF, FLOAT and UINT pixels; PIZ / PXR24 / B44 / DWA, tiled, deep and multi-part files are rejected with a
 * message.  channel: name to read; empty = the only channel, else the first of Z, Y, R that exists.  Written from the
 * published file layout ("OpenEXR File Layout", openexr.com); the same decoder as emfusion_amd/readers.py read_exr.
 */
Size readExr(const std::string& path, std::vector<float>& pixels, const std::string& channel = std::string());

/**
 * Keeps the reference's class name: depth frames of a Co-Fusion style dataset, <base><colordir>/ColorNNNN.png and
 * <base><depthdir>/DepthNNNN.exr (reference src/utils/ImageReader.cpp; base with a trailing separator, as the reference
 * concatenates).  The two directories must hold the same number of .png / .exr files; frames start at the first index
 * for which both files exist; depth in metres, values above 100 set to 0 (ImageReader.cpp:112).
 */
class ImageReader {
public:
    ImageReader(std::string basepath, std::string colordir, std::string depthdir);
    size_t getNumFrames() const { return numFrames; }
    int firstIndex() const { return first; }
    std::string depthFileName(int index) const;
    std::string colorFileName(int index) const;
    /** depth of file index `index` (firstIndex() ... ); returns its size */
    Size readDepth(int index, std::vector<float>& depth) const;

private:
    std::string colorpath, depthpath;
    size_t numFrames = 0;
    int first = 0;
};

/** What MaskRCNN::loadPreprocessed hands to EMFusion::initOrMatchObjs. */
struct PreprocMasks {
    int width = 0, height = 0;
    std::vector<std::array<double, 4>> boxes;
    std::vector<std::vector<uint8_t>> masks;      // one W x H 0/1 image per instance
    std::vector<std::vector<double>> scores;      // the class scores per instance (81 for COCO)
};
/** Returns the number of instances; throws std::runtime_error on files it cannot interpret. */
int loadPreprocessedMasks(const std::string& filename, PreprocMasks& out);

}  // namespace emf
