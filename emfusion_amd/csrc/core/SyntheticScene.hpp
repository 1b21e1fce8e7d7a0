// SyntheticScene.hpp -- deterministic synthetic RGB-D stream for the hot path.
//
// Stands in for the reference's dataset readers (src/utils/*Reader.cpp, out of scope): an analytic
// scene -- tilted wall z = 2.2 + 0.15 x - 0.1 y, a floor, and N spheres moving on Lissajous paths --
// rendered to z-depth in metres with multiplicative noise and dropout, plus the ground-truth camera
// pose, object poses and per-object 0/1 masks that tracking and Mask R-CNN would have produced.
// Everything derives from the seed through a counter-based hash, so any frame can be rendered
// independently and identically on every rank.
#pragma once

#include <cstdint>
#include <vector>

#include "types.hpp"

namespace emf {

class SyntheticScene {
public:
    struct Sphere {
        Vec3f center0;   // rest position (world)
        float radius;
        Vec3f amp;       // Lissajous amplitudes [m]
        Vec3f freq;      // radians per frame
        Vec3f phase;
    };

    SyntheticScene(Size frameSize, const Matx33f& intr, int numSpheres, uint64_t seed = 0xE3F5,
                   float noiseSigma = 0.002f, float dropout = 0.01f);

    int numSpheres() const { return static_cast<int>(spheres.size()); }
    const Sphere& sphere(int k) const { return spheres[k]; }

    /** camera -> world: 5 cm circle, small oscillating rotation (<= 0.16 deg / frame). */
    Affine3f cameraPose(int frame) const;
    /** centre of sphere k at `frame` (world); moves <= 1 cm / frame. */
    Vec3f sphereCenter(int k, int frame) const;
    /** edge length of the object volume that holds sphere k (reference volPad = 2 on ~1.6 r). */
    float objectVolumeSize(int k) const { return 3.2f * spheres[k].radius; }

    /**
     * Render frame `frame`: depth (W*H floats, metres, 0 = invalid) and, if ids != nullptr, the
     * 1-based index of the sphere visible at each pixel (0 = background).
     */
    void render(int frame, float* depth, uint8_t* ids = nullptr) const;

private:
    Size size;
    Matx33f K;
    uint64_t seed;
    float noiseSigma, dropout;
    std::vector<Sphere> spheres;
};

}  // namespace emf
