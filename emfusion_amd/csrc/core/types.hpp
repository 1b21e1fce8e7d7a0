// types.hpp -- small POD math/image types that stand in for the OpenCV types in the reference's
// public signatures (cv::Vec3i, cv::Vec3f, cv::Matx33f, cv::Affine3f, cv::Size, cv::cuda::GpuMat,
// cv::cuda::Stream).  No OpenCV / Eigen / Sophus exists on the target box, so the kept class API
// (TSDF / ObjTSDF / EMFusion) is re-typed on these; memory layouts match what the reference's
// kernel wrappers reinterpret (row-major 3x3, 3 floats), see reference TSDF.cu:417-422.
#pragma once

#include <hip/hip_runtime_api.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "emf_hip.h"

namespace emf {

struct Vec3i {
    int32_t val[3] = {0, 0, 0};
    Vec3i() = default;
    Vec3i(int a, int b, int c) : val{a, b, c} {}
    static Vec3i all(int v) { return Vec3i(v, v, v); }
    int operator[](int i) const { return val[i]; }
    int& operator[](int i) { return val[i]; }
};

struct Vec3f {
    float val[3] = {0.f, 0.f, 0.f};
    Vec3f() = default;
    Vec3f(float a, float b, float c) : val{a, b, c} {}
    static Vec3f all(float v) { return Vec3f(v, v, v); }
    float operator[](int i) const { return val[i]; }
    float& operator[](int i) { return val[i]; }
};
inline Vec3f operator+(const Vec3f& a, const Vec3f& b) {
    return Vec3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]);
}
inline Vec3f operator-(const Vec3f& a, const Vec3f& b) {
    return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]);
}
inline Vec3f operator-(const Vec3f& a) { return Vec3f(-a[0], -a[1], -a[2]); }
inline Vec3f operator*(const Vec3f& a, float f) { return Vec3f(a[0] * f, a[1] * f, a[2] * f); }
// cv::Vec / float multiplies by the reciprocal (opencv2/core/matx.hpp, Matx_ScaleOp with 1.f / alpha):
// the same rounding is needed where the reference's host code divides a vector
inline Vec3f operator/(const Vec3f& a, float f) {
    const float r = 1.f / f;
    return Vec3f(a[0] * r, a[1] * r, a[2] * r);
}

// Row-major 3x3, same memory as cv::Matx33f.
struct Matx33f {
    float val[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Matx33f() = default;
    Matx33f(float a, float b, float c, float d, float e, float f, float g, float h, float i)
        : val{a, b, c, d, e, f, g, h, i} {}
    explicit Matx33f(const float* p) : val{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]} {}
    static Matx33f eye() { return Matx33f(); }
    float operator()(int r, int c) const { return val[3 * r + c]; }
    float& operator()(int r, int c) { return val[3 * r + c]; }
    Matx33f t() const {
        return Matx33f(val[0], val[3], val[6], val[1], val[4], val[7], val[2], val[5], val[8]);
    }
};
inline Matx33f operator*(const Matx33f& a, const Matx33f& b) {
    Matx33f r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
inline Vec3f operator*(const Matx33f& m, const Vec3f& v) {
    return Vec3f(m(0, 0) * v[0] + m(0, 1) * v[1] + m(0, 2) * v[2],
                 m(1, 0) * v[0] + m(1, 1) * v[1] + m(1, 2) * v[2],
                 m(2, 0) * v[0] + m(2, 1) * v[1] + m(2, 2) * v[2]);
}

// Rigid transform x' = R x + t in single precision, like cv::Affine3f restricted to what the
// path uses (rotation(), translation(), inv(), composition, translate()).
class Affine3f {
public:
    Affine3f() = default;
    Affine3f(const Matx33f& R, const Vec3f& t) : R_(R), t_(t) {}
    static Affine3f Identity() { return Affine3f(); }
    const Matx33f& rotation() const { return R_; }
    const Vec3f& translation() const { return t_; }
    Affine3f translate(const Vec3f& d) const { return Affine3f(R_, t_ + d); }
    // inverse of a rigid transform: (R^T, -R^T t)
    Affine3f inv() const {
        const Matx33f Rt = R_.t();
        return Affine3f(Rt, -(Rt * t_));
    }
    // (a * b)(x) = a(b(x))
    friend Affine3f operator*(const Affine3f& a, const Affine3f& b) {
        return Affine3f(a.R_ * b.R_, a.R_ * b.t_ + a.t_);
    }

private:
    Matx33f R_;
    Vec3f t_;
};

// What cv::viz::Mesh carries in the reference (TSDF.cpp:365-371): vertex cloud, per-vertex normals,
// polygons as (3, i0, i1, i2) quadruples.
struct Mesh {
    std::vector<float> cloud;       // 3 per vertex, volume frame
    std::vector<float> normals;     // 3 per vertex (interpolated gradients, not normalised)
    std::vector<int32_t> polygons;  // 4 per triangle
    size_t vertices() const { return cloud.size() / 3; }
    size_t triangles() const { return polygons.size() / 4; }
};

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    size_t area() const { return static_cast<size_t>(width) * height; }
};

class HipError : public std::runtime_error {
public:
    HipError(const std::string& what, int code) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

void hipCheck(hipError_t e, const char* what);
// throws HipError carrying emf_hip_last_error_string() when an emf_hip_* call fails
void emfCheck(int rc, const char* what);

// Stand-in for cv::cuda::Stream: owns (or borrows) a hipStream_t.
class Stream {
public:
    Stream();                       // creates a non-blocking stream
    explicit Stream(int priority);  // > 0: highest, < 0: lowest, 0: middle of the device's priority range
    explicit Stream(hipStream_t s); // borrows (never destroyed); nullptr = the null stream
    ~Stream();
    Stream(Stream&& o) noexcept;
    Stream& operator=(Stream&& o) noexcept;
    Stream(const Stream&) = delete;
    Stream& operator=(const Stream&) = delete;
    static Stream& Null();
    hipStream_t get() const { return s_; }
    emf_stream_t abi() const { return reinterpret_cast<emf_stream_t>(s_); }
    void waitForCompletion() const;
    // record this stream's event at its current tail (one event per stream, re-recorded)
    void record();
    // make this stream wait for `other`'s last record() (device-side, no host sync)
    void waitOn(const Stream& other);
    // = other.record() + waitOn(other)
    void waitFor(Stream& other);

private:
    hipStream_t s_ = nullptr;
    bool owned_ = false;
    hipEvent_t ev_ = nullptr;  // lazily created by record()
};

/** Environment switches whose A/B is on record as lost ("switchboard"): read only by builds with
 *  -DEMF_DEBUG_SWITCHES (make EXTRA_HOST=-DEMF_DEBUG_SWITCHES); the product build ignores them. */
const char* demotedSwitchSet(const char* name);  // types.cpp: warns once per variable that is set, returns nullptr
inline const char* debugEnv(const char* name) {
#ifdef EMF_DEBUG_SWITCHES
    return std::getenv(name);
#else
    return demotedSwitchSet(name);  // a script that still sets it A/Bs two identical configurations: say so, once
#endif
}

/** Queue priority of one of the frame's streams: `dflt` unless the environment variable `env` says "high" / "+1",
 *  "normal" / "0" or "low" / "-1" (A/B measurements of the stream-to-queue mapping). */
inline int streamPriority(const char* env, int dflt) {
    const char* v = debugEnv(env);
    if (!v || !v[0]) return dflt;
    if (v[0] == 'h' || v[0] == '+' || v[0] == '1') return 1;
    if (v[0] == 'l' || v[0] == '-') return -1;
    return 0;
}

// Continuous device buffer (what cv::cuda::createContinuous gives): RAII over hipMalloc + a pool.
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes);
    ~DeviceBuffer();
    DeviceBuffer(DeviceBuffer&& o) noexcept;
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* data() const { return p_; }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p_);
    }
    size_t bytes() const { return n_; }
    bool empty() const { return p_ == nullptr; }
    // Released buffers wait in a process-wide pool (types.cpp, capped by EMF_POOL_MIB) instead of going
    // through hipFree, which synchronises the device.  trimPool() really frees them (and does wait);
    // emf_fusion_trim_pool() is its C entry, and a failing hipMalloc trims and retries once.
    static size_t pooledBytes();
    static void trimPool();
    void setZero(const Stream& s) const;
    void fill32(uint32_t pattern, const Stream& s) const;  // bytes() must be a multiple of 4
    void download(void* host, const Stream& s) const;       // synchronises `s`
    void upload(const void* host, const Stream& s) const;   // async on `s`

private:
    void* p_ = nullptr;
    size_t n_ = 0;
    uint64_t home_ = 0;  // the host thread that allocated it (types.cpp: only that thread fences and re-uses it)
};

// Continuous W x H image with C interleaved channels of T (the GpuMat of the reference).
template <typename T, int C = 1>
class DeviceImage {
public:
    DeviceImage() = default;
    explicit DeviceImage(Size sz) : size_(sz), buf_(sz.area() * C * sizeof(T)) {}
    Size size() const { return size_; }
    bool empty() const { return buf_.empty(); }
    T* ptr() const { return buf_.template as<T>(); }
    size_t bytes() const { return buf_.bytes(); }
    emf_image_t view() const {
        return emf_image_t{buf_.data(), static_cast<size_t>(size_.width) * C * sizeof(T),
                           size_.width, size_.height};
    }
    void setZero(const Stream& s) const { buf_.setZero(s); }
    void setTo(T v, const Stream& s) const {
        static_assert(sizeof(T) == 4 || sizeof(T) == 1, "fill supports 1- and 4-byte elements");
        if constexpr (sizeof(T) == 4) {
            uint32_t bits;
            __builtin_memcpy(&bits, &v, 4);
            buf_.fill32(bits, s);
        } else {
            hipCheck(hipMemsetAsync(buf_.data(), static_cast<int>(v), buf_.bytes(), s.get()),
                     "DeviceImage::setTo");
        }
    }
    std::vector<T> download(const Stream& s) const {
        std::vector<T> h(size_.area() * C);
        buf_.download(h.data(), s);
        return h;
    }
    void upload(const T* host, const Stream& s) const { buf_.upload(host, s); }

private:
    Size size_;
    DeviceBuffer buf_;
};

// Non-owning view of an image that lives in caller-managed device memory (e.g. a depth frame
// already resident in HBM).
inline emf_image_t imageView(const void* dev, Size sz, size_t elemBytes, size_t pitch = 0) {
    return emf_image_t{const_cast<void*>(dev),
                       pitch ? pitch : static_cast<size_t>(sz.width) * elemBytes, sz.width,
                       sz.height};
}

}  // namespace emf
