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
