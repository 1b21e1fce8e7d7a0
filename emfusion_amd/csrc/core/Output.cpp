#include "Output.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>

#include <sys/stat.h>
#include <cerrno>
#include <zlib.h>

namespace emf {
namespace io {

void writeVolume(const std::string& filename, const void* voxels, size_t elemSize,
                 const Vec3i& resolution, float voxelSize) {
    std::ofstream ofile(filename, std::ios::binary);
    const int32_t res[3] = {resolution[0], resolution[1], resolution[2]};
    const uint64_t es = elemSize;  // size_t of the reference's 64-bit build
    const size_t total = static_cast<size_t>(res[0]) * res[1] * res[2];
    ofile.write(reinterpret_cast<const char*>(res), sizeof(res));
    ofile.write(reinterpret_cast<const char*>(&es), sizeof(es));
    ofile.write(reinterpret_cast<const char*>(&voxelSize), sizeof(float));
    ofile.write(static_cast<const char*>(voxels), static_cast<std::streamsize>(total * elemSize));
    ofile.close();
    if (!ofile.good()) throw std::runtime_error("emf::io::writeVolume: error writing " + filename);
}

std::vector<float> readVolume(const std::string& filename, Vec3i& resolution, float& voxelSize) {
    std::ifstream ifile(filename, std::ios::binary);
    int32_t res[3] = {0, 0, 0};
    uint64_t es = 0;
    ifile.read(reinterpret_cast<char*>(res), sizeof(res));
    ifile.read(reinterpret_cast<char*>(&es), sizeof(es));
    ifile.read(reinterpret_cast<char*>(&voxelSize), sizeof(float));
    if (!ifile.good() || es != sizeof(float) || res[0] <= 0 || res[1] <= 0 || res[2] <= 0)
        throw std::runtime_error("emf::io::readVolume: " + filename + " is not a float volume dump");
    resolution = Vec3i(res[0], res[1], res[2]);
    std::vector<float> v(static_cast<size_t>(res[0]) * res[1] * res[2]);
    ifile.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(float)));
    if (!ifile.good()) throw std::runtime_error("emf::io::readVolume: " + filename + " is truncated");
    return v;
}

void rotationToQuaternion(const Matx33f& m, float q[4]) {
    // Eigen::Quaternion from a rotation matrix (Shepperd): q = (x, y, z, w)
    float t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.f) {
        t = std::sqrt(t + 1.f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m(2, 1) - m(1, 2)) * t;
        q[1] = (m(0, 2) - m(2, 0)) * t;
        q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
}

void writePoseFile(const std::string& filename, const std::map<int, Affine3f>& poses) {
    std::ofstream f(filename);
    for (const auto& p : poses) {
        float q[4];
        rotationToQuaternion(p.second.rotation(), q);
        const Vec3f& t = p.second.translation();
        f << p.first << " " << t[0] << " " << t[1] << " " << t[2] << " " << q[0] << " " << q[1] << " "
          << q[2] << " " << q[3] << std::endl;
    }
    f.close();
    if (!f.good()) throw std::runtime_error("emf::io::writePoseFile: error writing " + filename);
}

std::array<uint8_t, 768> randomColors() {
    float rgb[256][3];
    for (int i = 0; i < 256; ++i) {
        // hsv = (i / 256 * 360, 1, 1); OpenCV's HSV2RGB_f with hrange 360: sector + fraction of h * 6 / 360
        float h = i == 0 ? 0.f : (static_cast<float>(i) / 256.f) * 360.f;
        const float s = i == 0 ? 0.f : 1.f, v = i == 0 ? 0.f : 1.f;
        float r = v, g = v, b = v;
        if (s != 0.f) {
            static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
            h *= 6.f / 360.f;
            while (h < 0.f) h += 6.f;
            while (h >= 6.f) h -= 6.f;
            int sector = static_cast<int>(std::floor(h));
            h -= static_cast<float>(sector);
            if (static_cast<unsigned>(sector) >= 6u) {
                sector = 0;
                h = 0.f;
            }
            const float tab[4] = {v, v * (1.f - s), v * (1.f - s * h), v * (1.f - s * (1.f - h))};
            b = tab[sector_data[sector][0]];
            g = tab[sector_data[sector][1]];
            r = tab[sector_data[sector][2]];
        }
        rgb[i][0] = r;
        rgb[i][1] = g;
        rgb[i][2] = b;
    }
    std::array<uint8_t, 768> out{};
    for (int i = 0; i < 256; ++i)
        for (int c = 0; c < 3; ++c) {  // convertTo(CV_8U, 255): saturate_cast<uchar>(cvRound(x * 255))
            const long q = std::lrintf(rgb[i][c] * 255.f);
            out[3 * i + c] = static_cast<uint8_t>(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    // cv::randShuffle(rgb, 1, &rng) with cv::RNG rng(6893): multiply-with-carry generator,
    // for i in [0, n): swap(a[rng % n], a[i])
    uint64_t state = 6893;
    for (unsigned i = 0; i < 256; ++i) {
        state = static_cast<uint64_t>(static_cast<uint32_t>(state)) * 4164903690ull + static_cast<uint32_t>(state >> 32);
        const unsigned j = static_cast<uint32_t>(state) % 256u;
        for (int c = 0; c < 3; ++c) std::swap(out[3 * j + c], out[3 * i + c]);
    }
    out[0] = out[1] = out[2] = 255;
    return out;
}

void writeMesh(const std::string& filename, const Mesh& mesh) {
    FILE* file = std::fopen(filename.c_str(), "w");
    if (!file) throw std::runtime_error("Could not write ply file: " + filename);
    const int nv = static_cast<int>(mesh.vertices()), nf = static_cast<int>(mesh.triangles());
    std::fprintf(file,
                 "ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                 "element face %d\nproperty list uchar int vertex_index\nend_header\n",
                 nv, nf);
    for (int i = 0; i < nv; ++i) {
        const float* v = &mesh.cloud[3 * static_cast<size_t>(i)];
        const float* n = &mesh.normals[3 * static_cast<size_t>(i)];
        std::fprintf(file, "%f %f %f %f %f %f\n", v[0], v[1], v[2], n[0], n[1], n[2]);
    }
    for (int i = 0; i < nf; ++i) {
        const int32_t* t = &mesh.polygons[4 * static_cast<size_t>(i)];
        std::fprintf(file, "%d %d %d %d\n", t[0], t[1], t[2], t[3]);
    }
    if (std::fclose(file) != 0) throw std::runtime_error("emf::io::writeMesh: error writing " + filename);
}

std::vector<uint8_t> toU8Times255(const float* src, int width, int height, size_t pitchFloats) {
    std::vector<uint8_t> out(static_cast<size_t>(width) * height);
    for (int y = 0; y < height; ++y) {
        const float* row = src + static_cast<size_t>(y) * pitchFloats;
        uint8_t* o = out.data() + static_cast<size_t>(y) * width;
        for (int x = 0; x < width; ++x) {
            const float v = row[x] * 255.f;
            // cvRound = lrint in the default rounding mode (ties to even); NaN and values beyond int saturate
            int i = 0;
            if (v >= 255.f) i = 255;
            else if (v > 0.f) i = static_cast<int>(std::nearbyint(v));
            o[x] = static_cast<uint8_t>(i < 0 ? 0 : (i > 255 ? 255 : i));
        }
    }
    return out;
}

std::vector<uint8_t> encodePng(const uint8_t* pixels, int width, int height, int channels) {
    if (width < 1 || height < 1 || (channels != 1 && channels != 3))
        throw std::runtime_error("emf::io::encodePng: bad image shape");
    const size_t stride = static_cast<size_t>(width) * channels;
    std::vector<uint8_t> raw((stride + 1) * height);
    for (int y = 0; y < height; ++y) {  // filter type 0 (None) in front of every scan line
        raw[(stride + 1) * y] = 0;
        std::copy(pixels + stride * y, pixels + stride * (y + 1), raw.begin() + (stride + 1) * y + 1);
    }
    uLongf zlen = compressBound(static_cast<uLong>(raw.size()));
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK)
        throw std::runtime_error("emf::io::encodePng: zlib failed");
    z.resize(zlen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto be32 = [&](uint32_t v) {
        for (int k = 3; k >= 0; --k) out.push_back(static_cast<uint8_t>(v >> (8 * k)));
    };
    auto chunk = [&](const char kind[4], const uint8_t* body, size_t n) {
        be32(static_cast<uint32_t>(n));
        const size_t at = out.size();
        out.insert(out.end(), kind, kind + 4);
        out.insert(out.end(), body, body + n);
        be32(static_cast<uint32_t>(crc32(0L, out.data() + at, static_cast<uInt>(4 + n))));
    };
    uint8_t ihdr[13];
    for (int k = 0; k < 4; ++k) {
        ihdr[k] = static_cast<uint8_t>(static_cast<uint32_t>(width) >> (8 * (3 - k)));
        ihdr[4 + k] = static_cast<uint8_t>(static_cast<uint32_t>(height) >> (8 * (3 - k)));
    }
    ihdr[8] = 8;                       // bit depth
    ihdr[9] = channels == 1 ? 0 : 2;   // colour type: grayscale / truecolour
    ihdr[10] = ihdr[11] = ihdr[12] = 0;  // deflate, adaptive filtering, no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    return out;
}

void writeBytes(const std::string& filename, const std::vector<uint8_t>& bytes) {
    std::ofstream f(filename, std::ios::binary);
    f.write(reinterpret_cast<const char*>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
    f.close();
    if (!f.good()) throw std::runtime_error("emf::io::writeBytes: error writing " + filename);
}

void createDirectories(const std::string& dir) {
    for (size_t k = 1; k <= dir.size(); ++k)
        if (k == dir.size() || dir[k] == '/') {
            const std::string part = dir.substr(0, k);
            if (!part.empty() && mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
                throw std::runtime_error("emf::io::createDirectories: cannot create " + part);
        }
}

void writeImageLog(const std::string& dir, const std::map<int, std::vector<uint8_t>>& pngByFrame) {
    createDirectories(dir);
    for (const auto& e : pngByFrame) {
        char name[32];
        std::snprintf(name, sizeof(name), "/%04d.png", e.first);  // setfill('0') << setw(4) << id
        writeBytes(dir + name, e.second);
    }
}

}  // namespace io
}  // namespace emf
