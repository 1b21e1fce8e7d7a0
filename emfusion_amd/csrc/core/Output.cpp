#include "Output.hpp"

#include <cmath>
#include <cstdint>
#include <fstream>
#include <stdexcept>

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

}  // namespace io
}  // namespace emf
