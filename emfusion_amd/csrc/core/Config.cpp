// Config.cpp -- see Config.hpp.  The file syntax is boost::program_options' config-file syntax as its documentation
// describes it (sections, name = value, '#' comments); the option list and the value parsers are those of the
// reference's apps/EM-Fusion.cpp:40-104, 272-363.
#include "Config.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace emf {
namespace {

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return s.substr(a, b - a);
}

// getValues<T> (EM-Fusion.cpp:40-57): split at any of ", ", every piece one number
template <class T>
std::vector<T> numbers(const std::string& value) {
    std::vector<T> out;
    size_t i = 0;
    while (i <= value.size()) {
        const size_t e = std::min(value.find_first_of(", ", i), value.size());
        const std::string piece = value.substr(i, e - i);
        if (!piece.empty()) {
            std::istringstream ss(piece);
            T v;
            if (!(ss >> v) || !(ss >> std::ws).eof()) throw std::runtime_error("invalid option value '" + value + "'");
            out.push_back(v);
        }
        i = e + 1;
    }
    return out;
}
template <class T>
T number(const std::string& value) {
    std::istringstream ss(value);
    T v;
    if (!(ss >> v) || !(ss >> std::ws).eof()) throw std::runtime_error("invalid option value '" + value + "'");
    return v;
}
bool boolean(std::string v) {  // boost's validator for bool
    std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
    if (v.empty() || v == "on" || v == "yes" || v == "1" || v == "true") return true;
    if (v == "off" || v == "no" || v == "0" || v == "false") return false;
    throw std::runtime_error("invalid bool value '" + v + "'");
}

using Setter = std::function<void(Params&, const std::string&)>;
template <class T>
Setter scalar(T Params::*m) { return [m](Params& p, const std::string& v) { p.*m = number<T>(v); }; }
template <class T>
Setter tsdf(T TSDFParams::*m) { return [m](Params& p, const std::string& v) { p.tsdfParams.*m = number<T>(v); }; }
Setter vec3(Vec3i Params::*m) {
    return [m](Params& p, const std::string& v) {
        const std::vector<int> n = numbers<int>(v);
        if (n.size() != 3) throw std::runtime_error("invalid option value '" + v + "' (three integers expected)");
        p.*m = Vec3i(n[0], n[1], n[2]);
    };
}
Setter intrinsic(int r, int c) { return [r, c](Params& p, const std::string& v) { p.intr(r, c) = number<float>(v); }; }

const std::map<std::string, Setter>& options() {  // EM-Fusion.cpp:272-363
    static const std::map<std::string, Setter> o = {
        {"Params.frameSize", [](Params& p, const std::string& v) {
             const std::vector<int> n = numbers<int>(v);
             if (n.size() != 2) throw std::runtime_error("invalid option value '" + v + "' (two integers expected)");
             p.frameSize = Size(n[0], n[1]);
         }},
        {"Params.intr.fx", intrinsic(0, 0)}, {"Params.intr.fy", intrinsic(1, 1)},
        {"Params.intr.cx", intrinsic(0, 2)}, {"Params.intr.cy", intrinsic(1, 2)},
        {"Params.bilateral_sigma_depth", scalar(&Params::bilateral_sigma_depth)},
        {"Params.bilateral_sigma_spatial", scalar(&Params::bilateral_sigma_spatial)},
        {"Params.bilateral_kernel_size", scalar(&Params::bilateral_kernel_size)},
        {"Params.globalVolumeDims", vec3(&Params::globalVolumeDims)},
        {"Params.globalVoxelSize", scalar(&Params::globalVoxelSize)},
        {"Params.globalRelTruncDist", scalar(&Params::globalRelTruncDist)},
        {"Params.objVolumeDims", vec3(&Params::objVolumeDims)},
        {"Params.objRelTruncDist", scalar(&Params::objRelTruncDist)},
        {"Params.volumePose", [](Params& p, const std::string& v) {  // cv::Affine3f().translate(t), EM-Fusion.cpp:87-101
             const std::vector<float> n = numbers<float>(v);
             if (n.size() != 3) throw std::runtime_error("invalid option value '" + v + "' (three numbers expected)");
             p.volumePose = Affine3f(Matx33f::eye(), Vec3f(n[0], n[1], n[2]));
         }},
        {"Params.volPad", scalar(&Params::volPad)},
        {"Params.maxTrackingIter", scalar(&Params::maxTrackingIter)},
        {"Params.maskRCNNFrames", scalar(&Params::maskRCNNFrames)},
        {"Params.existenceThresh", scalar(&Params::existenceThresh)},
        {"Params.volIOUThresh", scalar(&Params::volIOUThresh)},
        {"Params.matchIOUThresh", scalar(&Params::matchIOUThresh)},
        {"Params.distanceThresh", scalar(&Params::distanceThresh)},
        {"Params.visibilityThresh", scalar(&Params::visibilityThresh)},
        {"Params.assocThresh", scalar(&Params::assocThresh)},
        {"Params.boundary", scalar(&Params::boundary)},
        {"Params.tsdfParams.tau", tsdf(&TSDFParams::tau)},
        {"Params.tsdfParams.eps1", tsdf(&TSDFParams::eps1)},
        {"Params.tsdfParams.eps2", tsdf(&TSDFParams::eps2)},
        {"Params.tsdfParams.nu_init", tsdf(&TSDFParams::nu_init)},
        {"Params.tsdfParams.huberThresh", tsdf(&TSDFParams::huberThresh)},
        {"Params.tsdfParams.maxTSDFWeight", tsdf(&TSDFParams::maxTSDFWeight)},
        {"Params.tsdfParams.assocSigma", tsdf(&TSDFParams::assocSigma)},
        {"Params.tsdfParams.alpha", tsdf(&TSDFParams::alpha)},
        {"Params.tsdfParams.uniPrior", tsdf(&TSDFParams::uniPrior)},
        {"Params.ignore_person", [](Params& p, const std::string& v) { p.ignore_person = boolean(v); }},
        {"Params.MaskRCNNParams.FILTER_CLASSES", [](Params& p, const std::string& v) { p.FILTER_CLASSES.push_back(v); }},
        {"Params.MaskRCNNParams.STATIC_OBJECTS", [](Params& p, const std::string& v) { p.STATIC_OBJECTS.push_back(v); }},
    };
    return o;
}

}  // namespace

void loadConfigFile(Params& p, const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("can not read options configuration file '" + path + "'");
    std::string line, section;
    std::set<std::string> seen;
    bool clearedFilter = false, clearedStatic = false;
    for (int no = 1; std::getline(f, line); ++no) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        line = trim(line);
        if (line.empty()) continue;
        auto fail = [&](const std::string& what) -> void {
            throw std::runtime_error(path + ":" + std::to_string(no) + ": " + what);
        };
        if (line.front() == '[' && line.back() == ']') {
            section = trim(line.substr(1, line.size() - 2));
            if (!section.empty() && section.back() != '.') section += '.';
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) fail("unrecognized line '" + line + "'");
        const std::string key = section + trim(line.substr(0, eq)), value = trim(line.substr(eq + 1));
        const auto it = options().find(key);
        if (it == options().end()) fail("unrecognised option '" + key + "'");
        const bool list = key == "Params.MaskRCNNParams.FILTER_CLASSES" || key == "Params.MaskRCNNParams.STATIC_OBJECTS";
        if (!list && !seen.insert(key).second) fail("option '" + key + "' cannot be specified more than once");
        // a list given in the file replaces what `p` held (a vector option starts empty in the reference)
        if (key == "Params.MaskRCNNParams.FILTER_CLASSES" && !clearedFilter) { p.FILTER_CLASSES.clear(); clearedFilter = true; }
        if (key == "Params.MaskRCNNParams.STATIC_OBJECTS" && !clearedStatic) { p.STATIC_OBJECTS.clear(); clearedStatic = true; }
        try {
            it->second(p, value);
        } catch (const std::runtime_error& e) {
            fail(std::string("option '") + key + "': " + e.what());
        }
    }
}

bool loadCalibrationFile(Params& p, const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    // calibstr >> fx >> fy >> cx >> cy >> width >> height (EM-Fusion.cpp:401-408): what parses is taken
    float v[4];
    for (int k = 0; k < 4; ++k)
        if (!(f >> v[k])) return true;
    p.intr(0, 0) = v[0];
    p.intr(1, 1) = v[1];
    p.intr(0, 2) = v[2];
    p.intr(1, 2) = v[3];
    int w = 0, h = 0;
    if (f >> w >> h) p.frameSize = Size(w, h);
    return true;
}

std::string dumpConfig(const Params& p) {
    std::ostringstream o;
    o.precision(9);
    o << "Params.frameSize = " << p.frameSize.width << " " << p.frameSize.height << "\n"
      << "Params.intr.fx = " << p.intr(0, 0) << "\nParams.intr.fy = " << p.intr(1, 1) << "\nParams.intr.cx = " << p.intr(0, 2)
      << "\nParams.intr.cy = " << p.intr(1, 2) << "\n"
      << "Params.bilateral_sigma_depth = " << p.bilateral_sigma_depth << "\nParams.bilateral_sigma_spatial = "
      << p.bilateral_sigma_spatial << "\nParams.bilateral_kernel_size = " << p.bilateral_kernel_size << "\n"
      << "Params.globalVolumeDims = " << p.globalVolumeDims[0] << " " << p.globalVolumeDims[1] << " " << p.globalVolumeDims[2] << "\n"
      << "Params.globalVoxelSize = " << p.globalVoxelSize << "\nParams.globalRelTruncDist = " << p.globalRelTruncDist << "\n"
      << "Params.objVolumeDims = " << p.objVolumeDims[0] << " " << p.objVolumeDims[1] << " " << p.objVolumeDims[2] << "\n"
      << "Params.objRelTruncDist = " << p.objRelTruncDist << "\n"
      << "Params.volumePose = " << p.volumePose.translation()[0] << " " << p.volumePose.translation()[1] << " "
      << p.volumePose.translation()[2] << "\n"
      << "Params.volPad = " << p.volPad << "\nParams.maxTrackingIter = " << p.maxTrackingIter << "\nParams.maskRCNNFrames = "
      << p.maskRCNNFrames << "\nParams.existenceThresh = " << p.existenceThresh << "\nParams.volIOUThresh = " << p.volIOUThresh
      << "\nParams.matchIOUThresh = " << p.matchIOUThresh << "\nParams.distanceThresh = " << p.distanceThresh
      << "\nParams.visibilityThresh = " << p.visibilityThresh << "\nParams.assocThresh = " << p.assocThresh
      << "\nParams.boundary = " << p.boundary << "\n"
      << "Params.tsdfParams.tau = " << p.tsdfParams.tau << "\nParams.tsdfParams.eps1 = " << p.tsdfParams.eps1
      << "\nParams.tsdfParams.eps2 = " << p.tsdfParams.eps2 << "\nParams.tsdfParams.nu_init = " << p.tsdfParams.nu_init
      << "\nParams.tsdfParams.huberThresh = " << p.tsdfParams.huberThresh << "\nParams.tsdfParams.maxTSDFWeight = "
      << p.tsdfParams.maxTSDFWeight << "\nParams.tsdfParams.assocSigma = " << p.tsdfParams.assocSigma
      << "\nParams.tsdfParams.alpha = " << p.tsdfParams.alpha << "\nParams.tsdfParams.uniPrior = " << p.tsdfParams.uniPrior << "\n"
      << "Params.ignore_person = " << (p.ignore_person ? "yes" : "no") << "\n";
    for (const std::string& s : p.FILTER_CLASSES) o << "Params.MaskRCNNParams.FILTER_CLASSES = " << s << "\n";
    for (const std::string& s : p.STATIC_OBJECTS) o << "Params.MaskRCNNParams.STATIC_OBJECTS = " << s << "\n";
    return o.str();
}

}  // namespace emf
