// SyntheticScene.cpp -- see SyntheticScene.hpp.  Host-only code (input generation, untimed).
#include "SyntheticScene.hpp"

#include <cmath>
#include <limits>

namespace emf {
namespace {

// splitmix64: counter-based, so pixel (x, y) of frame f has its own stream
inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double u01(uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); }

struct Rng {
    uint64_t s;
    double next() {
        s = mix(s);
        return u01(s);
    }
    double uniform(double a, double b) { return a + (b - a) * next(); }
};

Matx33f rotationAbout(double ax, double ay, double az, double angle) {
    const double n = std::sqrt(ax * ax + ay * ay + az * az);
    ax /= n; ay /= n; az /= n;
    const double c = std::cos(angle), s = std::sin(angle), t = 1 - c;
    return Matx33f(static_cast<float>(t * ax * ax + c), static_cast<float>(t * ax * ay - s * az),
                   static_cast<float>(t * ax * az + s * ay), static_cast<float>(t * ax * ay + s * az),
                   static_cast<float>(t * ay * ay + c), static_cast<float>(t * ay * az - s * ax),
                   static_cast<float>(t * ax * az - s * ay), static_cast<float>(t * ay * az + s * ax),
                   static_cast<float>(t * az * az + c));
}

}  // namespace

SyntheticScene::SyntheticScene(Size frameSize, const Matx33f& intr, int numSpheres, uint64_t _seed,
                               float _noiseSigma, float _dropout)
    : size(frameSize), K(intr), seed(_seed), noiseSigma(_noiseSigma), dropout(_dropout) {
    Rng rng{mix(seed)};
    for (int k = 0; k < numSpheres; ++k) {
        Sphere s;
        s.radius = static_cast<float>(rng.uniform(0.15, 0.30));
        // centres in a 2 x 1.5 x 0.8 m box in front of the camera, above the floor
        s.center0 = Vec3f(static_cast<float>(rng.uniform(-1.0, 1.0)),
                          static_cast<float>(rng.uniform(-0.75, 0.45)),
                          static_cast<float>(rng.uniform(1.2, 2.0)));
        // |velocity| <= amp * freq per axis; 0.06 m * 0.08 rad/frame = 4.8 mm / frame / axis
        s.amp = Vec3f(static_cast<float>(rng.uniform(0.02, 0.06)),
                      static_cast<float>(rng.uniform(0.01, 0.04)),
                      static_cast<float>(rng.uniform(0.02, 0.06)));
        s.freq = Vec3f(static_cast<float>(rng.uniform(0.03, 0.08)),
                       static_cast<float>(rng.uniform(0.03, 0.08)),
                       static_cast<float>(rng.uniform(0.03, 0.08)));
        s.phase = Vec3f(static_cast<float>(rng.uniform(0, 6.28)),
                        static_cast<float>(rng.uniform(0, 6.28)),
                        static_cast<float>(rng.uniform(0, 6.28)));
        spheres.push_back(s);
    }
}

Affine3f SyntheticScene::cameraPose(int frame) const {
    const double a = 2.0 * M_PI * frame / 90.0;
    const Vec3f t(static_cast<float>(0.05 * std::cos(a) - 0.05), static_cast<float>(0.05 * std::sin(a)),
                  0.f);
    const double ang = (3.0 * M_PI / 180.0) * std::sin(2.0 * M_PI * frame / 120.0);
    return Affine3f(rotationAbout(0.2, 1.0, 0.1, ang), t);
}

Vec3f SyntheticScene::sphereCenter(int k, int frame) const {
    const Sphere& s = spheres[k];
    return Vec3f(s.center0[0] + s.amp[0] * std::sin(s.freq[0] * frame + s.phase[0]),
                 s.center0[1] + s.amp[1] * std::sin(s.freq[1] * frame + s.phase[1]),
                 s.center0[2] + s.amp[2] * std::sin(s.freq[2] * frame + s.phase[2]));
}

void SyntheticScene::render(int frame, float* depth, uint8_t* ids) const {
    const Affine3f cam = cameraPose(frame);
    const Matx33f& R = cam.rotation();
    const Vec3f& o = cam.translation();
    std::vector<Vec3f> centers(spheres.size());
    for (size_t k = 0; k < spheres.size(); ++k) centers[k] = sphereCenter(static_cast<int>(k), frame);
    const double fx = K(0, 0), fy = K(1, 1), cx = K(0, 2), cy = K(1, 2);
    // wall: z = c + a x + b y  <=>  n . p = c with n = (-a, -b, 1)
    const double wa = 0.15, wb = -0.1, wc = 2.2, floorY = 1.0;
    for (int y = 0; y < size.height; ++y) {
        for (int x = 0; x < size.width; ++x) {
            // ray through the pixel with camera-frame z component 1: the ray parameter IS z-depth
            const double dcx = (x - cx) / fx, dcy = (y - cy) / fy;
            const double dx = R(0, 0) * dcx + R(0, 1) * dcy + R(0, 2);
            const double dy = R(1, 0) * dcx + R(1, 1) * dcy + R(1, 2);
            const double dz = R(2, 0) * dcx + R(2, 1) * dcy + R(2, 2);
            double best = std::numeric_limits<double>::infinity();
            int id = 0;
            const double denom = -wa * dx - wb * dy + dz;
            if (std::fabs(denom) > 1e-12) {
                const double s = (wc - (-wa * o[0] - wb * o[1] + o[2])) / denom;
                if (s > 0 && s < best) best = s;
            }
            if (std::fabs(dy) > 1e-12) {
                const double s = (floorY - o[1]) / dy;
                if (s > 0 && s < best) best = s;
            }
            const double A = dx * dx + dy * dy + dz * dz;
            for (size_t k = 0; k < spheres.size(); ++k) {
                const double ocx = o[0] - centers[k][0], ocy = o[1] - centers[k][1],
                             ocz = o[2] - centers[k][2];
                const double B = 2 * (dx * ocx + dy * ocy + dz * ocz);
                const double C = ocx * ocx + ocy * ocy + ocz * ocz -
                                 static_cast<double>(spheres[k].radius) * spheres[k].radius;
                const double disc = B * B - 4 * A * C;
                if (disc < 0) continue;
                const double s = (-B - std::sqrt(disc)) / (2 * A);
                if (s > 0 && s < best) {
                    best = s;
                    id = static_cast<int>(k) + 1;
                }
            }
            double d = std::isfinite(best) ? best : 0.0;
            const uint64_t h = mix(seed ^ (static_cast<uint64_t>(frame) << 40) ^
                                   (static_cast<uint64_t>(y) << 20) ^ static_cast<uint64_t>(x));
            if (noiseSigma > 0) {  // Box-Muller on two hashed uniforms
                const double u1 = u01(mix(h ^ 0x1111)) + 1e-12, u2 = u01(mix(h ^ 0x2222));
                d *= 1.0 + noiseSigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
            }
            if (dropout > 0 && u01(mix(h ^ 0x3333)) < dropout) d = 0.0;
            const size_t i = static_cast<size_t>(y) * size.width + x;
            depth[i] = static_cast<float>(d);
            if (ids) ids[i] = static_cast<uint8_t>(id);
        }
    }
}

}  // namespace emf
