// Readers.cpp -- see Readers.hpp.  The PNG layout follows the PNG specification (chunks, zlib stream, scan-line
// filters, section 9.2); the pickle layer follows the opcode list of CPython's pickletools (protocols 0-5) as far
// as the files of the reference's preprocessing need it.
#include "Readers.hpp"

#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace emf {
namespace {

std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

}  // namespace

void readPngGray(const std::string& path, std::vector<uint16_t>& pixels, int& width, int& height) {
    const std::string raw = slurp(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (raw.size() < 8 || std::memcmp(raw.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG file");
    const auto* p = reinterpret_cast<const unsigned char*>(raw.data());
    size_t pos = 8;
    std::string idat;
    int depth = 0, color = -1, interlace = 0;
    width = height = 0;
    while (pos + 12 <= raw.size()) {
        const uint32_t n = be32(p + pos);
        const char* kind = raw.data() + pos + 4;
        if (pos + 12 + n > raw.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        if (!std::memcmp(kind, "IHDR", 4) && n >= 13) {
            width = static_cast<int>(be32(p + pos + 8));
            height = static_cast<int>(be32(p + pos + 12));
            depth = p[pos + 16];
            color = p[pos + 17];
            interlace = p[pos + 20];
        } else if (!std::memcmp(kind, "IDAT", 4)) {
            idat.append(raw.data() + pos + 8, n);
        } else if (!std::memcmp(kind, "IEND", 4)) {
            break;
        }
        pos += 12 + n;
    }
    if (color != 0 || (depth != 8 && depth != 16) || interlace != 0 || width <= 0 || height <= 0)
        throw std::runtime_error(path + ": only non-interlaced 8/16-bit grayscale PNGs are supported");
    const int bpp = depth / 8;
    const size_t stride = static_cast<size_t>(width) * bpp;
    std::vector<unsigned char> rows((stride + 1) * height);
    uLongf got = rows.size();
    if (uncompress(rows.data(), &got, reinterpret_cast<const Bytef*>(idat.data()), idat.size()) != Z_OK || got != rows.size())
        throw std::runtime_error(path + ": cannot inflate the image data");
    // scan-line filters (PNG 9.2): Sub / Average / Paeth predict from the reconstructed bytes to the left (a),
    // above (b) and above-left (c)
    std::vector<unsigned char> img(stride * height);
    const unsigned char* prev = nullptr;
    for (int y = 0; y < height; ++y) {
        const unsigned char* in = rows.data() + static_cast<size_t>(y) * (stride + 1);
        unsigned char* cur = img.data() + static_cast<size_t>(y) * stride;
        const int f = in[0];
        if (f > 4) throw std::runtime_error(path + ": bad PNG filter type");
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= static_cast<size_t>(bpp) ? cur[x - bpp] : 0;
            const int b = prev ? prev[x] : 0;
            const int c = (prev && x >= static_cast<size_t>(bpp)) ? prev[x - bpp] : 0;
            int pred = 0;
            if (f == 1) pred = a;
            else if (f == 2) pred = b;
            else if (f == 3) pred = (a + b) >> 1;
            else if (f == 4) {
                const int q = a + b - c, pa = std::abs(q - a), pb = std::abs(q - b), pc = std::abs(q - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[x] = static_cast<unsigned char>(in[1 + x] + pred);
        }
        prev = cur;
    }
    pixels.resize(static_cast<size_t>(width) * height);
    for (size_t i = 0; i < pixels.size(); ++i)
        pixels[i] = bpp == 1 ? img[i] : static_cast<uint16_t>((img[2 * i] << 8) | img[2 * i + 1]);  // big endian
}

// ---- TUM RGB-D sequences ---------------------------------------------------------------------------------------

TUMRGBDReader::TUMRGBDReader(std::string path_) : path(std::move(path_)) {
    std::vector<double> stamps;
    readFileAssociations(path + "associations.txt", rgbFileNames, depthFileNames, &stamps);
    const size_t n = depthFileNames.size();
    frameRate = n > 1 && stamps.back() > stamps.front() ? static_cast<double>(n) / (stamps.back() - stamps.front()) : 0.0;
}

void TUMRGBDReader::readFileAssociations(const std::string& filename, std::vector<std::string>& rgbNames,
                                         std::vector<std::string>& depthNames, std::vector<double>* stamps) {
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("Could not open association file!");  // (the reference's message)
    std::string line;
    int rgbFirst = -1;
    while (std::getline(f, line)) {
        std::vector<std::string> parts;  // split at single blanks / tabs, like boost::split(is_any_of("\t "))
        std::string cur;
        for (char ch : line) {
            if (ch == ' ' || ch == '\t') {
                parts.push_back(cur);
                cur.clear();
            } else if (ch != '\r') {
                cur.push_back(ch);
            }
        }
        parts.push_back(cur);
        if (parts.size() != 4 || (!line.empty() && line[0] == '#')) continue;
        if (rgbFirst < 0) rgbFirst = parts[1].rfind("rgb/", 0) == 0 ? 1 : 0;
        rgbNames.push_back(rgbFirst ? parts[1] : parts[3]);
        depthNames.push_back(rgbFirst ? parts[3] : parts[1]);
        if (stamps) stamps->push_back(std::atof(parts[0].c_str()));
    }
}

Size TUMRGBDReader::readDepth(size_t i, std::vector<float>& depth) const {
    std::vector<uint16_t> px;
    int w = 0, h = 0;
    readPngGray(path + depthFileNames.at(i), px, w, h);
    depth.resize(px.size());
    const float s = 1.f / 5000.f;  // TUM depth scale (TUMRGBDReader.cpp: convertTo(CV_32FC1, 1 / 5000.))
    for (size_t k = 0; k < px.size(); ++k) depth[k] = static_cast<float>(px[k]) * s;
    return Size(w, h);
}

// ---- OpenEXR scan-line files (the depth images of the Co-Fusion datasets) --------------------------------------------
namespace {

float halfToFloat(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal half: normalise
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            bits = sign | static_cast<uint32_t>(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;  // inf / NaN
    else bits = sign | (exp + 127 - 15) << 23 | man << 13;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

uint32_t le32(const unsigned char* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
uint64_t le64(const unsigned char* p) { return uint64_t(le32(p)) | uint64_t(le32(p + 4)) << 32; }

// -n literal bytes / the next byte n + 1 times (OpenEXR's run-length code)
std::string exrUnRle(const unsigned char* d, size_t n, size_t want, const std::string& path) {
    std::string out;
    out.reserve(want);
    size_t i = 0;
    while (i < n) {
        const int c = static_cast<signed char>(d[i++]);
        if (c < 0) {
            const size_t k = static_cast<size_t>(-c);
            if (i + k > n) throw std::runtime_error(path + ": truncated RLE block");
            out.append(reinterpret_cast<const char*>(d + i), k);
            i += k;
        } else {
            if (i >= n) throw std::runtime_error(path + ": truncated RLE block");
            out.append(static_cast<size_t>(c) + 1, static_cast<char>(d[i++]));
        }
    }
    if (out.size() != want) throw std::runtime_error(path + ": EXR RLE block decodes to the wrong size");
    return out;
}

// undo the byte predictor and the even / odd split OpenEXR applies before RLE / ZIP
std::string exrUnpredict(const std::string& t) {
    std::string d(t);
    for (size_t k = 1; k < d.size(); ++k)
        d[k] = static_cast<char>(static_cast<unsigned char>(d[k - 1]) + static_cast<unsigned char>(d[k]) - 128);
    std::string out(d.size(), '\0');
    const size_t half = (d.size() + 1) / 2;
    for (size_t k = 0; k < d.size(); ++k) out[k] = (k & 1) ? d[half + k / 2] : d[k / 2];
    return out;
}

}  // namespace

Size readExr(const std::string& path, std::vector<float>& pixels, const std::string& channelIn) {
    const std::string raw = slurp(path);
    const unsigned char* r = reinterpret_cast<const unsigned char*>(raw.data());
    if (raw.size() < 8 || le32(r) != 0x01312f76u) throw std::runtime_error(path + " is not an OpenEXR file");
    const uint32_t version = le32(r + 4);
    if ((version & 0xffu) != 2 || (version & 0x1a00u))  // tiled, deep, multi-part
        throw std::runtime_error(path + ": only single-part scan-line EXR files are supported");
    size_t pos = 8;
    std::map<std::string, std::string> attrs;
    auto cstr = [&](size_t& p) {
        const size_t e = raw.find('\0', p);
        if (e == std::string::npos) throw std::runtime_error(path + ": truncated EXR header");
        std::string s = raw.substr(p, e - p);
        p = e + 1;
        return s;
    };
    while (pos < raw.size() && raw[pos] != '\0') {
        const std::string name = cstr(pos);
        cstr(pos);  // type name
        if (pos > raw.size() || raw.size() - pos < 4) throw std::runtime_error(path + ": truncated EXR header");
        const uint32_t size = le32(r + pos);
        pos += 4;
        if (size > raw.size() - pos) throw std::runtime_error(path + ": truncated EXR header");
        attrs[name] = raw.substr(pos, size);
        pos += size;
    }
    ++pos;
    if (!attrs.count("channels") || !attrs.count("compression") || !attrs.count("dataWindow") || attrs["dataWindow"].size() != 16)
        throw std::runtime_error(path + ": EXR header lacks channels / compression / dataWindow");
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    {
        const std::string& c = attrs["channels"];
        size_t i = 0;
        while (i < c.size() && c[i] != '\0') {
            const size_t e = c.find('\0', i);
            if (e == std::string::npos || e + 17 > c.size()) throw std::runtime_error(path + ": bad EXR channel list");
            const unsigned char* q = reinterpret_cast<const unsigned char*>(c.data()) + e + 1;
            const int type = static_cast<int>(le32(q));
            if (le32(q + 8) != 1 || le32(q + 12) != 1) throw std::runtime_error(path + ": sub-sampled EXR channels are not supported");
            if (type < 0 || type > 2) throw std::runtime_error(path + ": unknown EXR pixel type");
            chans.push_back({c.substr(i, e - i), type});
            i = e + 17;
        }
    }
    const int comp = static_cast<unsigned char>(attrs["compression"][0]);
    if (comp > 3)
        throw std::runtime_error(path + ": EXR compression " + std::to_string(comp) + " is not supported (NONE, RLE, ZIPS, ZIP are)");
    const unsigned char* dw = reinterpret_cast<const unsigned char*>(attrs["dataWindow"].data());
    const int xmin = static_cast<int>(le32(dw)), ymin = static_cast<int>(le32(dw + 4)), xmax = static_cast<int>(le32(dw + 8)),
              ymax = static_cast<int>(le32(dw + 12));
    const long long wl = static_cast<long long>(xmax) - xmin + 1, hl = static_cast<long long>(ymax) - ymin + 1;
    if (wl < 1 || hl < 1 || wl > 65536 || hl > 65536) throw std::runtime_error(path + ": empty or implausible EXR data window");
    const int w = static_cast<int>(wl), h = static_cast<int>(hl);
    std::string channel = channelIn;
    auto has = [&](const std::string& n) {
        for (const Chan& c : chans)
            if (c.name == n) return true;
        return false;
    };
    if (channel.empty()) {
        if (chans.size() == 1) channel = chans[0].name;
        else
            for (const char* n : {"Z", "Y", "R"})
                if (has(n)) { channel = n; break; }
    }
    if (channel.empty() || !has(channel)) throw std::runtime_error(path + ": no channel '" + channel + "' in the EXR file");
    static const size_t pixelBytes[3] = {4, 2, 4};  // UINT, HALF, FLOAT
    size_t lineBytes = 0, offsetInLine = 0;
    int type = 2;
    bool found = false;
    for (const Chan& c : chans) {  // channels are stored in the order of the list (alphabetical)
        if (c.name == channel) { type = c.type; found = true; }
        if (!found) offsetInLine += pixelBytes[c.type] * w;
        lineBytes += pixelBytes[c.type] * w;
    }
    const int perBlock = comp == 3 ? 16 : 1;
    const int nblocks = (h + perBlock - 1) / perBlock;
    if (pos > raw.size() || 8 * static_cast<size_t>(nblocks) > raw.size() - pos)
        throw std::runtime_error(path + ": truncated EXR offset table");
    // the image is sized from a 16-byte header field: before allocating, hold it against what the file can carry behind
    // the offset table.  A stored block never shrinks below its lines / ratio: uncompressed (NONE) not at all, the RLE
    // at best 1 : 64 (a run of 128 bytes in 2), zlib at best ~1 : 1030 -- so w * h pixels need lineBytes * h / ratio
    // bytes of blocks, and a 16 MB file cannot ask for a 16 GiB image unless it really is all ZIP-compressed zeros.
    // (An image larger than 2^28 pixels -- 16384 x 16384; depth maps are 0.3 - 1.2 Mpixel -- is refused outright.)
    const size_t ratio = comp == 0 ? 1 : (comp == 1 ? 64 : 1032);
    const size_t avail = raw.size() - pos - 8 * static_cast<size_t>(nblocks);
    if (lineBytes == 0 || lineBytes * static_cast<size_t>(h) / ratio > avail)
        throw std::runtime_error(path + ": EXR data window is larger than the file can hold");
    if (static_cast<size_t>(w) * static_cast<size_t>(h) > (size_t(1) << 28))
        throw std::runtime_error(path + ": EXR data window above 2^28 pixels");
    pixels.assign(static_cast<size_t>(w) * h, 0.f);
    for (int b = 0; b < nblocks; ++b) {
        const uint64_t off = le64(r + pos + 8 * static_cast<size_t>(b));
        // (wrap-free: `off` is an untrusted 64-bit value)
        if (raw.size() < 8 || off > raw.size() - 8) throw std::runtime_error(path + ": EXR block offset beyond the file");
        const int y = static_cast<int>(le32(r + off));
        const uint32_t size = le32(r + off + 4);
        if (size > raw.size() - 8 - off || y < ymin || y > ymax) throw std::runtime_error(path + ": bad EXR block");
        const int lines = std::min(perBlock, ymax - y + 1);
        const size_t want = static_cast<size_t>(lines) * lineBytes;
        std::string data(raw, off + 8, size);
        if (comp != 0 && size < want) {  // a block that does not shrink is stored as it is
            if (comp == 1) data = exrUnRle(r + off + 8, size, want, path);
            else {
                std::string z(want, '\0');
                uLongf got = static_cast<uLongf>(want);
                if (uncompress(reinterpret_cast<Bytef*>(&z[0]), &got, r + off + 8, size) != Z_OK || got != want)
                    throw std::runtime_error(path + ": cannot inflate an EXR block");
                data.swap(z);
            }
            data = exrUnpredict(data);
        }
        if (data.size() != want) throw std::runtime_error(path + ": EXR block has the wrong size");
        for (int l = 0; l < lines; ++l) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(data.data()) + l * lineBytes + offsetInLine;
            float* dst = pixels.data() + static_cast<size_t>(y - ymin + l) * w;
            for (int x = 0; x < w; ++x) {
                if (type == 1) dst[x] = halfToFloat(static_cast<uint16_t>(src[2 * x] | src[2 * x + 1] << 8));
                else if (type == 2) { const uint32_t u = le32(src + 4 * x); std::memcpy(&dst[x], &u, 4); }
                else dst[x] = static_cast<float>(le32(src + 4 * x));
            }
        }
    }
    return Size(w, h);
}

// ---- ImageReader (reference src/utils/ImageReader.cpp) -----------------------------------------------------------------
namespace {
int countWithExtension(const std::string& dir, const char* ext) {
    DIR* d = opendir(dir.c_str());
    if (!d) return -1;
    int n = 0;
    const size_t el = std::strlen(ext);
    while (const dirent* e = readdir(d)) {
        const size_t l = std::strlen(e->d_name);
        if (l > el && std::strcmp(e->d_name + l - el, ext) == 0) ++n;
    }
    closedir(d);
    return n;
}
bool fileExists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
std::string indexed(const std::string& dir, const char* stem, int i, const char* ext) {
    char name[64];
    std::snprintf(name, sizeof(name), "/%s%04d%s", stem, i, ext);  // setfill('0') << setw(4)
    return dir + name;
}
}  // namespace

ImageReader::ImageReader(std::string basepath, std::string colordir, std::string depthdir)
    : colorpath(basepath + colordir), depthpath(basepath + depthdir) {
    const int rgbs = countWithExtension(colorpath, ".png"), depths = countWithExtension(depthpath, ".exr");
    if (rgbs < 0 || depths < 0) throw std::runtime_error("Could not read color or depth dir!");
    if (rgbs != depths) throw std::runtime_error("Different number of rgb and depth files!");
    numFrames = static_cast<size_t>(rgbs);
    while (!(fileExists(colorFileName(first)) && fileExists(depthFileName(first)))) {  // ImageReader.cpp:79-94
        ++first;
        if (first >= rgbs) throw std::runtime_error("Could not find starting index!");
    }
}
std::string ImageReader::depthFileName(int index) const { return indexed(depthpath, "Depth", index, ".exr"); }
std::string ImageReader::colorFileName(int index) const { return indexed(colorpath, "Color", index, ".png"); }
Size ImageReader::readDepth(int index, std::vector<float>& depth) const {
    const Size s = readExr(depthFileName(index), depth);
    for (float& d : depth)
        if (d > 100.f) d = 0.f;  // depth.setTo ( 0, depth > 100 )
    return s;
}

// ---- a small unpickler ---------------------------------------------------------------------------------------------
namespace {

struct PyVal;
using P = std::shared_ptr<PyVal>;
struct PyVal {
    enum Kind { None, Bool, Int, Float, Bytes, Tuple, List, Dict, Global, Object, Mark } kind = None;
    long long i = 0;
    double f = 0;
    std::string s;           // Bytes / text; Global: "module name"
    std::vector<P> items;    // Tuple / List; Dict: key, value, key, value ...
    P callable, args, state; // Object: callable(*args), then __setstate__(state)
};
P mk(PyVal::Kind k) {
    auto v = std::make_shared<PyVal>();
    v->kind = k;
    return v;
}

class Unpickler {
public:
    explicit Unpickler(const std::string& data, std::string name) : d(data), file(std::move(name)) {}
    P load() {
        for (;;) {
            const unsigned char op = u8();
            switch (op) {
                case 0x80: u8(); break;                       // PROTO
                case 0x95: skip(8); break;                    // FRAME
                case '.': return pop();                       // STOP
                case '(': st.push_back(mk(PyVal::Mark)); break;
                case '0': pop(); break;
                case '1': popMark(); break;
                case '2': st.push_back(top()); break;
                case 'N': st.push_back(mk(PyVal::None)); break;
                case 0x88: case 0x89: { auto v = mk(PyVal::Bool); v->i = op == 0x88; st.push_back(v); break; }
                case 'I': {
                    const std::string t = line();
                    auto v = mk(t == "00" || t == "01" ? PyVal::Bool : PyVal::Int);
                    v->i = (t == "01") ? 1 : (t == "00" ? 0 : std::atoll(t.c_str()));
                    st.push_back(v);
                    break;
                }
                case 'J': pushInt(static_cast<int32_t>(le(4))); break;
                case 'K': pushInt(u8()); break;
                case 'M': pushInt(static_cast<long long>(le(2))); break;
                case 'L': { std::string t = line(); if (!t.empty() && t.back() == 'L') t.pop_back(); pushInt(std::atoll(t.c_str())); break; }
                case 0x8a: pushInt(longBytes(u8())); break;
                case 0x8b: pushInt(longBytes(static_cast<size_t>(le(4)))); break;
                case 'F': { auto v = mk(PyVal::Float); v->f = std::atof(line().c_str()); st.push_back(v); break; }
                case 'G': {
                    need(8);
                    uint64_t b = 0;
                    for (int k = 0; k < 8; ++k) b = (b << 8) | static_cast<unsigned char>(d[pos + k]);
                    pos += 8;
                    auto v = mk(PyVal::Float);
                    std::memcpy(&v->f, &b, 8);
                    st.push_back(v);
                    break;
                }
                case 'S': { std::string t = line(); pushBytes(unquote(t)); break; }
                case 'V': pushBytes(rawUnicodeToUtf8(line())); break;
                case 'T': case 'X': case 'B': pushBytes(take(static_cast<size_t>(le(4)))); break;
                case 'U': case 'C': case 0x8c: pushBytes(take(u8())); break;
                case 0x8d: case 0x8e: case 0x96: pushBytes(take(static_cast<size_t>(le(8)))); break;
                case ')': st.push_back(mk(PyVal::Tuple)); break;
                case 't': { auto v = mk(PyVal::Tuple); v->items = popMark(); st.push_back(v); break; }
                case 0x85: case 0x86: case 0x87: {
                    const size_t n = op - 0x84;
                    if (st.size() < n) fail("stack underflow");
                    auto v = mk(PyVal::Tuple);
                    v->items.assign(st.end() - n, st.end());
                    st.resize(st.size() - n);
                    st.push_back(v);
                    break;
                }
                case ']': st.push_back(mk(PyVal::List)); break;
                case 'l': { auto v = mk(PyVal::List); v->items = popMark(); st.push_back(v); break; }
                case 'a': { P x = pop(); top()->items.push_back(x); break; }
                case 'e': { auto xs = popMark(); auto& it = top()->items; it.insert(it.end(), xs.begin(), xs.end()); break; }
                case '}': st.push_back(mk(PyVal::Dict)); break;
                case 'd': { auto v = mk(PyVal::Dict); v->items = popMark(); st.push_back(v); break; }
                case 's': { P val = pop(), key = pop(); top()->items.push_back(key); top()->items.push_back(val); break; }
                case 'u': { auto xs = popMark(); auto& it = top()->items; it.insert(it.end(), xs.begin(), xs.end()); break; }
                case 'c': { auto v = mk(PyVal::Global); const std::string m = line(); v->s = m + " " + line(); st.push_back(v); break; }
                case 0x93: { P n = pop(), m = pop(); auto v = mk(PyVal::Global); v->s = m->s + " " + n->s; st.push_back(v); break; }
                case 'R': case 0x81: { P a = pop(), c = pop(); auto v = mk(PyVal::Object); v->callable = c; v->args = a; st.push_back(v); break; }
                case 'b': { P s = pop(); top()->state = s; break; }
                case 'p': memo[std::atoll(line().c_str())] = top(); break;
                case 'q': memo[u8()] = top(); break;
                case 'r': memo[static_cast<long long>(le(4))] = top(); break;
                case 0x94: memo[static_cast<long long>(memo.size())] = top(); break;
                case 'g': st.push_back(get(std::atoll(line().c_str()))); break;
                case 'h': st.push_back(get(u8())); break;
                case 'j': st.push_back(get(static_cast<long long>(le(4)))); break;
                default: {
                    char buf[64];
                    std::snprintf(buf, sizeof(buf), "pickle opcode 0x%02x is not supported", op);
                    fail(buf);
                }
            }
        }
    }

private:
    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error(file + ": " + what); }
    void need(size_t n) const { if (pos + n > d.size()) fail("truncated pickle"); }
    unsigned char u8() { need(1); return static_cast<unsigned char>(d[pos++]); }
    void skip(size_t n) { need(n); pos += n; }
    uint64_t le(int n) {
        need(n);
        uint64_t v = 0;
        for (int k = n - 1; k >= 0; --k) v = (v << 8) | static_cast<unsigned char>(d[pos + k]);
        pos += n;
        return v;
    }
    long long longBytes(size_t n) {  // little-endian two's complement
        need(n);
        long long v = 0;
        for (size_t k = 0; k < n && k < 8; ++k) v |= static_cast<long long>(static_cast<unsigned char>(d[pos + k])) << (8 * k);
        if (n > 0 && n < 8 && (static_cast<unsigned char>(d[pos + n - 1]) & 0x80)) v |= -(1ll << (8 * n));
        pos += n;
        return v;
    }
    std::string take(size_t n) { need(n); std::string s = d.substr(pos, n); pos += n; return s; }
    std::string line() {
        const size_t e = d.find('\n', pos);
        if (e == std::string::npos) fail("truncated pickle");
        std::string s = d.substr(pos, e - pos);
        pos = e + 1;
        return s;
    }
    static void putUtf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o.push_back(static_cast<char>(cp));
        else if (cp < 0x800) { o.push_back(static_cast<char>(0xC0 | (cp >> 6))); o.push_back(static_cast<char>(0x80 | (cp & 63))); }
        else { o.push_back(static_cast<char>(0xE0 | (cp >> 12))); o.push_back(static_cast<char>(0x80 | ((cp >> 6) & 63))); o.push_back(static_cast<char>(0x80 | (cp & 63))); }
    }
    static std::string rawUnicodeToUtf8(const std::string& t) {  // protocol 0 text: latin-1 bytes + \uXXXX escapes
        std::string o;
        for (size_t k = 0; k < t.size(); ++k) {
            if (t[k] == '\\' && k + 5 < t.size() + 1 && t[k + 1] == 'u') { putUtf8(o, static_cast<unsigned>(std::stoul(t.substr(k + 2, 4), nullptr, 16))); k += 5; }
            else putUtf8(o, static_cast<unsigned char>(t[k]));
        }
        return o;
    }
    static std::string unquote(const std::string& t) {  // 'text' with \xNN, \\, \', \n escapes (protocol 0 strings)
        std::string o;
        if (t.size() < 2) return o;
        for (size_t k = 1; k + 1 < t.size(); ++k) {
            if (t[k] != '\\' || k + 2 >= t.size()) { o.push_back(t[k]); continue; }
            const char c = t[++k];
            if (c == 'x' && k + 2 < t.size()) { o.push_back(static_cast<char>(std::stoi(t.substr(k + 1, 2), nullptr, 16))); k += 2; }
            else if (c == 'n') o.push_back('\n');
            else if (c == 't') o.push_back('\t');
            else if (c == 'r') o.push_back('\r');
            else if (c >= '0' && c <= '7') { int v = 0, n = 0; while (n < 3 && k < t.size() && t[k] >= '0' && t[k] <= '7') { v = 8 * v + (t[k++] - '0'); ++n; } --k; o.push_back(static_cast<char>(v)); }
            else o.push_back(c);
        }
        return o;
    }
    void pushInt(long long v) { auto x = mk(PyVal::Int); x->i = v; st.push_back(x); }
    void pushBytes(std::string s) { auto x = mk(PyVal::Bytes); x->s = std::move(s); st.push_back(x); }
    P pop() { if (st.empty()) fail("stack underflow"); P v = st.back(); st.pop_back(); return v; }
    P& top() { if (st.empty()) fail("stack underflow"); return st.back(); }
    std::vector<P> popMark() {
        size_t k = st.size();
        while (k > 0 && st[k - 1]->kind != PyVal::Mark) --k;
        if (k == 0) fail("no MARK on the stack");
        std::vector<P> xs(st.begin() + k, st.end());
        st.resize(k - 1);
        return xs;
    }
    P get(long long key) { auto it = memo.find(key); if (it == memo.end()) fail("bad memo key"); return it->second; }

    const std::string& d;
    std::string file;
    size_t pos = 0;
    std::vector<P> st;
    std::map<long long, P> memo;
};

// bytes as Python 3 writes them below protocol 3: _codecs.encode(<text>, 'latin1') -- the text is UTF-8 here
std::string bytesOf(const P& v, const std::string& file) {
    if (v && v->kind == PyVal::Bytes) return v->s;
    if (v && v->kind == PyVal::Object && v->callable && v->callable->kind == PyVal::Global && v->callable->s == "_codecs encode" &&
        v->args && !v->args->items.empty() && v->args->items[0]->kind == PyVal::Bytes) {
        const std::string& t = v->args->items[0]->s;
        std::string o;
        for (size_t k = 0; k < t.size(); ++k) {
            const unsigned char c = static_cast<unsigned char>(t[k]);
            if (c < 0x80) o.push_back(static_cast<char>(c));
            else if ((c & 0xE0) == 0xC0 && k + 1 < t.size()) { o.push_back(static_cast<char>(((c & 31) << 6) | (static_cast<unsigned char>(t[k + 1]) & 63))); ++k; }
            else throw std::runtime_error(file + ": text that is not latin-1 where bytes were expected");
        }
        return o;
    }
    throw std::runtime_error(file + ": expected bytes");
}

// numpy.ndarray as its __reduce__ writes it: _reconstruct(ndarray, (0,), b'b') + state (version, shape, dtype,
// is_fortran, data); protocol 5 in-band: numpy.core.numeric._frombuffer(data, dtype, shape, order)
struct NdArray {
    std::vector<long long> shape;
    int itemsize = 1;
    char kind = 'u';  // 'b' bool, 'u' / 'i' integers, 'f' floats
    bool fortran = false;
    std::string data;
    double number(const char* p) const {
        if (kind == 'f') {
            if (itemsize == 8) { double d; std::memcpy(&d, p, 8); return d; }
            if (itemsize == 4) { float f; std::memcpy(&f, p, 4); return f; }
            throw std::runtime_error("numpy float16 is not supported");
        }
        if (kind == 'i') {
            if (itemsize == 8) { int64_t v; std::memcpy(&v, p, 8); return static_cast<double>(v); }
            if (itemsize == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
            if (itemsize == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
            return static_cast<signed char>(*p);
        }
        uint64_t v = 0;
        std::memcpy(&v, p, static_cast<size_t>(itemsize));  // little endian (what x86 numpy writes; '>' is rejected)
        return static_cast<double>(v);
    }
};
bool endsWith(const std::string& s, const std::string& t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }
int dtypeItemsize(const P& dt, const std::string& file, char* kind = nullptr) {
    if (!dt || dt->kind != PyVal::Object || !dt->args || dt->args->items.empty() || dt->args->items[0]->kind != PyVal::Bytes)
        throw std::runtime_error(file + ": cannot read a numpy dtype");
    const std::string& t = dt->args->items[0]->s;  // "b1", "u1", "i8", "f4", ...
    const int n = t.size() >= 2 ? std::atoi(t.c_str() + 1) : 0;
    if ((n != 1 && n != 2 && n != 4 && n != 8) || std::string("buif").find(t[0]) == std::string::npos)
        throw std::runtime_error(file + ": numpy dtype " + t + " is not supported");
    if (dt->state && dt->state->kind == PyVal::Tuple && dt->state->items.size() > 1 && dt->state->items[1]->s == ">")
        throw std::runtime_error(file + ": big-endian numpy arrays are not supported");
    if (kind) *kind = t[0];
    return n;
}
bool asArray(const P& v, NdArray& a, const std::string& file) {
    if (!v || v->kind != PyVal::Object || !v->callable || v->callable->kind != PyVal::Global) return false;
    const std::string& fn = v->callable->s;
    auto shapeOf = [&](const P& t) {
        a.shape.clear();
        if (!t || (t->kind != PyVal::Tuple && t->kind != PyVal::List)) throw std::runtime_error(file + ": array shape is not a tuple");
        for (const P& e : t->items) {
            // (a negative extent would pass the size test below as a product of two negatives; 2^31 bounds the products)
            if (!e || e->kind != PyVal::Int || e->i < 0 || e->i > 0x7fffffffLL) throw std::runtime_error(file + ": bad array shape");
            a.shape.push_back(e->i);
        }
    };
    if (endsWith(fn, "_reconstruct") && v->state && v->state->kind == PyVal::Tuple && v->state->items.size() >= 5) {
        const auto& s = v->state->items;
        shapeOf(s[1]);
        a.itemsize = dtypeItemsize(s[2], file, &a.kind);
        if (!s[3]) throw std::runtime_error(file + ": malformed numpy array");
        a.fortran = s[3]->i != 0;
        a.data = bytesOf(s[4], file);
        return true;
    }
    if (endsWith(fn, "_frombuffer") && v->args && v->args->items.size() >= 4) {
        const auto& s = v->args->items;
        a.data = bytesOf(s[0], file);
        a.itemsize = dtypeItemsize(s[1], file, &a.kind);
        shapeOf(s[2]);
        if (!s[3]) throw std::runtime_error(file + ": malformed numpy array");
        a.fortran = s[3]->s == "F";
        return true;
    }
    return false;
}
double asNumber(const P& v, const std::string& file) {
    if (v && (v->kind == PyVal::Int || v->kind == PyVal::Bool)) return static_cast<double>(v->i);
    if (v && v->kind == PyVal::Float) return v->f;
    if (v && v->kind == PyVal::Object && v->callable && v->callable->kind == PyVal::Global && endsWith(v->callable->s, "scalar") &&
        v->args && v->args->items.size() == 2) {  // numpy.core.multiarray.scalar(dtype, bytes)
        const P& dt = v->args->items[0];  // numpy.dtype('f8', ...): an object whose first argument is the type string
        if (!dt || !dt->args || dt->args->items.empty() || !dt->args->items[0] || !v->args->items[1])
            throw std::runtime_error(file + ": malformed numpy scalar");
        const std::string& t = dt->args->items[0]->s;
        const std::string& b = v->args->items[1]->s;
        if (t == "f8" && b.size() == 8) { double d; std::memcpy(&d, b.data(), 8); return d; }
        if (t == "f4" && b.size() == 4) { float f; std::memcpy(&f, b.data(), 4); return f; }
        long long i = 0;
        std::memcpy(&i, b.data(), std::min<size_t>(8, b.size()));
        return static_cast<double>(i);
    }
    throw std::runtime_error(file + ": expected a number");
}
// rows of numbers from a list of lists or a 2-D float64 / float32 / integer array
void asRows(const P& v, std::vector<std::vector<double>>& rows, const std::string& file) {
    rows.clear();
    NdArray a;
    if (asArray(v, a, file)) {
        if (a.shape.size() != 2) {
            if (a.shape.size() == 1 && a.shape[0] == 0) return;
            throw std::runtime_error(file + ": expected a 2-D array of numbers");
        }
        const long long n = a.shape[0], m = a.shape[1];
        if (static_cast<size_t>(n * m * a.itemsize) != a.data.size()) throw std::runtime_error(file + ": array size mismatch");
        rows.assign(n, std::vector<double>(m));
        for (long long r = 0; r < n; ++r)
            for (long long c = 0; c < m; ++c) {
                rows[r][c] = a.number(a.data.data() + (a.fortran ? c * n + r : r * m + c) * a.itemsize);
            }
        return;
    }
    if (!v || (v->kind != PyVal::List && v->kind != PyVal::Tuple)) throw std::runtime_error(file + ": expected a sequence of rows");
    for (const P& row : v->items) {
        std::vector<double> r;
        NdArray ra;
        if (asArray(row, ra, file)) {
            if (ra.shape.size() != 1) throw std::runtime_error(file + ": expected 1-D rows");
            for (long long c = 0; c < ra.shape[0]; ++c) r.push_back(ra.number(ra.data.data() + c * ra.itemsize));
        } else {
            if (!row || (row->kind != PyVal::List && row->kind != PyVal::Tuple)) throw std::runtime_error(file + ": expected rows of numbers");
            for (const P& e : row->items) r.push_back(asNumber(e, file));
        }
        rows.push_back(std::move(r));
    }
}
void pushMask(const NdArray& a, size_t offset, long long h, long long w, bool fortran, PreprocMasks& out, const std::string& file) {
    if (out.masks.empty()) {
        out.width = static_cast<int>(w);
        out.height = static_cast<int>(h);
    } else if (out.width != w || out.height != h) {
        throw std::runtime_error(file + ": instance masks of different sizes");
    }
    std::vector<uint8_t> m(static_cast<size_t>(w * h));
    for (long long y = 0; y < h; ++y)
        for (long long x = 0; x < w; ++x) {
            const char* p = a.data.data() + offset + (fortran ? x * h + y : y * w + x) * a.itemsize;
            bool nz = false;
            for (int b = 0; b < a.itemsize; ++b) nz = nz || p[b] != 0;
            m[static_cast<size_t>(y * w + x)] = nz ? 1 : 0;
        }
    out.masks.push_back(std::move(m));
}

}  // namespace

int loadPreprocessedMasks(const std::string& filename, PreprocMasks& out) {
    out = PreprocMasks();
    const std::string raw = slurp(filename);
    const P top = Unpickler(raw, filename).load();
    if (!top || (top->kind != PyVal::Tuple && top->kind != PyVal::List) || top->items.size() != 3)
        throw std::runtime_error("Maskrcnn function did not return a tuple or a tuple of the wrong size!");  // (the reference's message)
    // masks: a list of (H, W) arrays -- generate_result's segmentation[:, :, m] -- or one (N, H, W) array
    const P& pm = top->items[1];
    NdArray a;
    if (asArray(pm, a, filename)) {
        if (a.shape.size() == 3) {
            const long long n = a.shape[0], h = a.shape[1], w = a.shape[2];
            if (a.fortran) throw std::runtime_error(filename + ": Fortran-ordered (N, H, W) mask array is not supported");
            if (static_cast<size_t>(n * h * w * a.itemsize) != a.data.size()) throw std::runtime_error(filename + ": array size mismatch");
            for (long long k = 0; k < n; ++k) pushMask(a, static_cast<size_t>(k * h * w * a.itemsize), h, w, false, out, filename);
        } else if (!(a.shape.size() == 1 && a.shape[0] == 0)) {
            throw std::runtime_error(filename + ": instance masks must be 2-D");
        }
    } else if (pm && (pm->kind == PyVal::List || pm->kind == PyVal::Tuple)) {
        for (const P& m : pm->items) {
            NdArray one;
            if (!asArray(m, one, filename) || one.shape.size() != 2) throw std::runtime_error(filename + ": instance masks must be 2-D arrays");
            if (static_cast<size_t>(one.shape[0] * one.shape[1] * one.itemsize) != one.data.size())
                throw std::runtime_error(filename + ": array size mismatch");
            pushMask(one, 0, one.shape[0], one.shape[1], one.fortran, out, filename);
        }
    } else {
        throw std::runtime_error(filename + ": cannot read the instance masks");
    }
    std::vector<std::vector<double>> boxes;
    asRows(top->items[0], boxes, filename);
    asRows(top->items[2], out.scores, filename);
    for (const auto& b : boxes) {
        if (b.size() != 4) throw std::runtime_error(filename + ": a bounding box needs 4 numbers");
        out.boxes.push_back({b[0], b[1], b[2], b[3]});
    }
    if (out.boxes.size() != out.masks.size() || (out.scores.size() != out.masks.size() && !out.scores.empty()))
        throw std::runtime_error(filename + ": boxes, masks and scores differ in number");
    return static_cast<int>(out.masks.size());
}

}  // namespace emf
