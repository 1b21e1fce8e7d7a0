// pixel_ops.hip -- per-pixel kernels: back-projection, trilinear volume lookups, the fused E-step
// (association likelihood + normalisation) and raycast compositing.
#include "device_core.hpp"

namespace emf_hip {
namespace {

constexpr int kTileX = 64;  // one wave spans 64 consecutive pixels of a row: coalesced image I/O
constexpr int kTileY = 4;

__device__ __forceinline__ bool pixel_of(int w, int h, int& x, int& y) {
    x = blockIdx.x * kTileX + threadIdx.x;
    y = blockIdx.y * kTileY + threadIdx.y;
    return x < w && y < h;
}

inline dim3 pixel_grid(int w, int h) { return dim3(ceil_div(w, kTileX), ceil_div(h, kTileY)); }
inline dim3 pixel_block() { return dim3(kTileX, kTileY); }

// ---- a1: computePoints (reference EMFusion.cu:29-47) ---------------------------------------------

__global__ __launch_bounds__(256) void k_compute_points(Img<const float> depth, Img<float> points,
                                                        int w, int h, float fx, float fy, float cx,
                                                        float cy) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    const float d = depth.row(y)[x];
    float* p = points.row(y) + 3 * x;
    p[0] = (static_cast<float>(x) - cx) * d / fx;
    p[1] = (static_cast<float>(y) - cy) * d / fy;
    p[2] = d;
}

// ---- a2: getVolumeVals (reference TSDF.cu:662-688) -----------------------------------------------

struct LookupArgs {
    const float* vol;
    Img<const float> points;
    Img<float> vals;
    int w, h;
    M33 R;  // camera -> volume
    V3 t;
    I3 n;
    float voxelSize;
};

template <int CH>
__global__ __launch_bounds__(256) void k_volume_vals(const LookupArgs a) {
    int x, y;
    if (!pixel_of(a.w, a.h, x, y)) return;
    float out[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) out[c] = 0.f;  // fused vals.setTo(0), TSDF.cu:705
    const float* pp = a.points.row(y) + 3 * x;
    const V3 pc = v3(pp[0], pp[1], pp[2]);
    if (pc.z > 0) {
        const V3 v = to_voxel(mul(a.R, pc) + a.t, a.voxelSize, half_extent(a.n));
        if (!outside(v, 1.f, a.n)) {
            const Cell c = cell_of(v, a.n);
            const size_t sy = static_cast<size_t>(a.n.x), sz = sy * a.n.y;
            const float* p = a.vol + c.base * CH;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                out[ch] = blend8(p[ch], p[CH + ch], p[sy * CH + ch], p[(sy + 1) * CH + ch],
                                 p[sz * CH + ch], p[(sz + 1) * CH + ch], p[(sz + sy) * CH + ch],
                                 p[(sz + sy + 1) * CH + ch], c.fx, c.fy, c.fz);
        }
    }
    float* o = a.vals.row(y) + CH * x;
#pragma unroll
    for (int c = 0; c < CH; ++c) o[c] = out[c];
}

// ---- a3-a5: association likelihood (reference TSDF.cpp:125-156, ObjTSDF.cpp:181-201) --------------

struct AssocArgs {
    AssocModel m;  // device_core.hpp
    Img<const float> points;
    Img<float> out;
    int w, h;
};

__global__ __launch_bounds__(256) void k_assoc(const AssocArgs a) {
    int x, y;
    if (!pixel_of(a.w, a.h, x, y)) return;
    const float* pp = a.points.row(y) + 3 * x;
    a.out.row(y)[x] = assoc_weight(a.m, v3(pp[0], pp[1], pp[2]));
}

// ---- a6: normalisation (reference EMFusion.cpp:653-665) ------------------------------------------

constexpr int kMapsPerLaunch = 16;

struct MapTable {
    Img<float> m[kMapsPerLaunch];
    int count;
};

// mode 0: norm = m0 + m1 + ...                (first chunk)
// mode 1: norm = norm + m0 + m1 + ...         (later chunks)
// extra (optional) is added after the last map of the last chunk.
__global__ __launch_bounds__(256) void k_assoc_sum(const MapTable t, Img<float> norm,
                                                   Img<const float> extra, int w, int h,
                                                   int accumulate) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    float s;
    int k = 0;
    if (accumulate) {
        s = norm.row(y)[x];
    } else {
        s = t.m[0].row(y)[x];
        k = 1;
    }
    for (; k < t.count; ++k) s = s + t.m[k].row(y)[x];
    if (extra.data) s = s + extra.row(y)[x];
    norm.row(y)[x] = s;
}

__global__ __launch_bounds__(256) void k_assoc_divide(const MapTable t, Img<const float> norm,
                                                      int w, int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    const float s = norm.row(y)[x];
    for (int k = 0; k < t.count; ++k) {
        float* p = t.m[k].row(y) + x;
        *p = (s != 0.f) ? *p / s : 0.f;  // cv::cuda::divide: x / 0 := 0 (Q7)
    }
}

// single-launch form for <= 16 maps: sum the first nsum in order, optional extra, divide all,
// optional norm output
__global__ __launch_bounds__(256) void k_assoc_normalize(const MapTable t, int nsum,
                                                         Img<const float> extra, Img<float> norm,
                                                         int w, int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    float v[kMapsPerLaunch];
    float s = 0.f;
    bool started = false;
#pragma unroll
    for (int k = 0; k < kMapsPerLaunch; ++k) {
        if (k < t.count) {
            v[k] = t.m[k].row(y)[x];
            if (k < nsum) {
                s = started ? s + v[k] : v[k];
                started = true;
            }
        }
    }
    if (extra.data) {
        const float e = extra.row(y)[x];
        s = started ? s + e : e;
    }
    if (norm.data) norm.row(y)[x] = s;
#pragma unroll
    for (int k = 0; k < kMapsPerLaunch; ++k)
        if (k < t.count) t.m[k].row(y)[x] = (s != 0.f) ? v[k] / s : 0.f;
}

// norm = extra (used when no local map enters the sum)
__global__ __launch_bounds__(256) void k_copy_map(Img<const float> src, Img<float> dst, int w,
                                                  int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    dst.row(y)[x] = src.row(y)[x];
}

// ---- a12: raycast compositing (reference EMFusion.cpp:760-794) -----------------------------------

constexpr int kObjsPerLaunch = 16;

struct CompositeTable {
    Img<const float> ray[kObjsPerLaunch];
    Img<const float> vert[kObjsPerLaunch];
    Img<const float> nrm[kObjsPerLaunch];
    Img<const uint8_t> seg[kObjsPerLaunch];
    uint8_t id[kObjsPerLaunch];  // saturated object id written into the segmentation
    int count;
};

// seg value -> object index (-1: no object has that id)
struct SlotTable {
    int16_t slot[256];
};

struct CompositeArgs {
    Img<const float> bgRay, bgVert, bgNorm;
    Img<const uint8_t> bgMask;
    Img<float> ray, vert, nrm, diff;
    Img<uint8_t> seg, noObj;
    int w, h;
    int first, last;  // chunk position
    int32_t* zero;    // first chunk: visCounts to clear for k_vis_counts, which runs after the last chunk
    int nzero;        // (<= 255: one workgroup's worth; saves the memset launch between the two)
    // last chunk, fused form (emf_hip_compositeVisibility): the visibility counts of k_vis_counts are taken here,
    // from the segmentation values this launch writes; `counts` must be zero when the launch starts
    int32_t* counts;
    int boundary;
    SlotTable slots;
};

__global__ __launch_bounds__(256) void k_composite(const CompositeTable t, const CompositeArgs a) {
    if (a.zero && blockIdx.x == 0 && blockIdx.y == 0) {
        const int i = threadIdx.y * blockDim.x + threadIdx.x;
        if (i < a.nzero) a.zero[i] = 0;
    }
    int x, y;
    const bool in = pixel_of(a.w, a.h, x, y);
    if (!in && !a.counts) return;  // (the fused form has barriers below)
    uint8_t s = 0;
    if (in) {
    float r = 0.f;
    V3 vv = v3(0.f, 0.f, 0.f), nn = v3(0.f, 0.f, 0.f);
    if (!a.first) {  // resume from the composite the previous chunk left
        r = a.ray.row(y)[x];
        const float* pv = a.vert.row(y) + 3 * x;
        const float* pn = a.nrm.row(y) + 3 * x;
        vv = v3(pv[0], pv[1], pv[2]);
        nn = v3(pn[0], pn[1], pn[2]);
        s = a.seg.row(y)[x];
    }
    for (int k = 0; k < t.count; ++k) {  // list order, strict '<' (Q15)
        const float rk = t.ray[k].row(y)[x];
        const bool take = t.seg[k].row(y)[x] != 0 && (r <= 0 || rk < r);
        if (take) {
            r = rk;
            const float* pv = t.vert[k].row(y) + 3 * x;
            const float* pn = t.nrm[k].row(y) + 3 * x;
            vv = v3(pv[0], pv[1], pv[2]);
            nn = v3(pn[0], pn[1], pn[2]);
            s = t.id[k];
        }
    }
    if (a.last) {
        float d = a.diff.row(y)[x];
        if (a.bgMask.row(y)[x]) {  // masked subtract, stale elsewhere (Q12)
            d = r - a.bgRay.row(y)[x];
            a.diff.row(y)[x] = d;
        }
        if (d > 0.05f) s = 0;  // background wins when it is > 5 cm in front
        const uint8_t no = s == 0 ? 255 : 0;
        if (no) {  // vertices / normals fall back to the background, the raylength does not
            const float* pv = a.bgVert.row(y) + 3 * x;
            const float* pn = a.bgNorm.row(y) + 3 * x;
            vv = v3(pv[0], pv[1], pv[2]);
            nn = v3(pn[0], pn[1], pn[2]);
        }
        a.noObj.row(y)[x] = no;
    }
    a.ray.row(y)[x] = r;
    float* ov = a.vert.row(y) + 3 * x;
    float* on = a.nrm.row(y) + 3 * x;
    ov[0] = vv.x;
    ov[1] = vv.y;
    ov[2] = vv.z;
    on[0] = nn.x;
    on[1] = nn.y;
    on[2] = nn.z;
    a.seg.row(y)[x] = s;
    }
    if (a.counts) {  // (uniform) k_vis_counts on the values just written
        __shared__ int lh[256];
        const int tid = threadIdx.y * kTileX + threadIdx.x;
        lh[tid] = 0;
        __syncthreads();
        if (in && s && x >= a.boundary && x < a.w - a.boundary && y >= a.boundary && y < a.h - a.boundary)
            atomicAdd(&lh[s], 1);
        __syncthreads();
        const int k = a.slots.slot[tid];
        if (k >= 0 && lh[tid]) atomicAdd(&a.counts[k], lh[tid]);
    }
}

// the gate of emf_hip_visibilityFlags, leaving the counts cleared for the next composite
__global__ void k_vis_flags_clear(int32_t* __restrict__ counts, int nmodels, int thresh,
                                  int32_t* __restrict__ visible, int32_t* __restrict__ mirror) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nmodels) return;
    const int32_t c = s == 0 ? 0 : counts[s - 1];
    visible[s] = s == 0 ? 1 : (c > thresh ? 1 : 0);
    if (s > 0) {
        if (mirror) mirror[s - 1] = c;  // host-visible copy: no copy kernel, no copy engine
        counts[s - 1] = 0;
    }
}

// visibility: per-object pixel counts inside the inset rectangle (EMFusion.cpp:778-791).
// Each workgroup builds a 256-bin LDS histogram of the segmentation, then adds the non-empty bins
// to the owning object's counter (slot = seg value -> object index, -1 if no object has that id).

__global__ __launch_bounds__(256) void k_vis_counts(Img<const uint8_t> seg, int w, int h,
                                                    int boundary, const SlotTable slots,
                                                    int* __restrict__ counts) {
    __shared__ int lh[256];
    const int tid = threadIdx.y * kTileX + threadIdx.x;
    lh[tid] = 0;
    __syncthreads();
    int x, y;
    if (pixel_of(w, h, x, y) && x >= boundary && x < w - boundary && y >= boundary &&
        y < h - boundary) {
        const uint8_t s = seg.row(y)[x];
        if (s) atomicAdd(&lh[s], 1);
    }
    __syncthreads();
    const int k = slots.slot[tid];
    if (k >= 0 && lh[tid]) atomicAdd(&counts[k], lh[tid]);
}

// ---- integrateMasks occlusion (reference EMFusion.cpp:897-900) -----------------------------------

__global__ __launch_bounds__(256) void k_occluded(Img<const uint8_t> objSeg, Img<const uint8_t> seg,
                                                  int id, Img<uint8_t> occ, int w, int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    const int own = (static_cast<int>(seg.row(y)[x]) == id) ? 255 : 0;
    const int v = static_cast<int>(objSeg.row(y)[x]) - own;
    occ.row(y)[x] = static_cast<uint8_t>(v < 0 ? 0 : v);
}


// ---- f-2: depth pre-processing (EMFusion::preprocessDepth, EMFusion.cpp:294-305) ----------------
// cv::cuda::bilateralFilter (OpenCV, third-party; restated, parity unpinned) + the reference's two
// patches (NaN -> 0, raw == 0 -> 0) in one pass.  A 32 x 8 pixel workgroup stages its tile plus the
// ksz / 2 halo in LDS (borders reflected, BORDER_REFLECT_101), so each depth value is read from
// memory once instead of up to 37 times.
constexpr int kBfX = 32, kBfY = 8, kBfMaxR = 7;  // kernel sizes up to 15

struct BilateralArgs {
    Img<const float> raw;
    Img<float> out;
    int w, h, ksz;
    float ss, sc;  // -0.5 / sigma_spatial^2, -0.5 / sigma_depth^2
};

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(kBfX* kBfY) void k_preprocess_depth(const BilateralArgs a) {
    __shared__ float tile[kBfY + 2 * kBfMaxR][kBfX + 2 * kBfMaxR + 1];
    const int r = a.ksz / 2;
    const int x0 = blockIdx.x * kBfX, y0 = blockIdx.y * kBfY;
    const int tw = kBfX + 2 * r, th = kBfY + 2 * r;
    for (int i = threadIdx.y * kBfX + threadIdx.x; i < tw * th; i += kBfX * kBfY) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[ty][tx] = a.raw.row(reflect101(y0 + ty - r, a.h))[reflect101(x0 + tx - r, a.w)];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= a.w || y >= a.h) return;
    const float center = tile[threadIdx.y + r][threadIdx.x + r];
    const float r2 = static_cast<float>(r * r);
    float sum1 = 0.f, sum2 = 0.f;
    for (int dy = 0; dy < a.ksz; ++dy)
        for (int dx = 0; dx < a.ksz; ++dx) {
            const float space2 = static_cast<float>((dx - r) * (dx - r) + (dy - r) * (dy - r));
            if (space2 > r2) continue;
            const float v = tile[threadIdx.y + dy][threadIdx.x + dx];
            const float d = fabsf(v - center);
            const float wgt = expf(space2 * a.ss + (d * d) * a.sc);
            sum1 = sum1 + wgt * v;
            sum2 = sum2 + wgt;
        }
    float o = sum1 / sum2;
    if (o != o) o = 0.f;         // compare(depth, depth, NE) -> setTo(0)
    if (center == 0.f) o = 0.f;  // compare(depth_raw, 0, EQ) -> setTo(0)
    a.out.row(y)[x] = o;
}

// ---- ignore_person in EMFusion::render (EMFusion.cpp:139-150): compare / setTo / 2 x masked copyTo ----
__global__ __launch_bounds__(256) void k_hide_label(Img<uint8_t> seg, int id, Img<float> vert, Img<float> nrm,
                                                    Img<const float> bgVert, Img<const float> bgNrm, int w, int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    if (seg.row(y)[x] != id) return;
    seg.row(y)[x] = 0;
    for (int c = 0; c < 3; ++c) {
        vert.row(y)[3 * x + c] = bgVert.row(y)[3 * x + c];
        nrm.row(y)[3 * x + c] = bgNrm.row(y)[3 * x + c];
    }
}

// ---- f-4: kernel_renderPhong + renderGPU's colour lookup (EMFusion.cu:100-186) ---------------------
// One launch: the label -> colour lookup (cv::cuda::LookUpTable on a 3-channel copy of the
// segmentation) happens in registers from a table passed by value, and background pixels are
// written as 0 here instead of by a preceding image.setTo(0).
struct PhongArgs {
    Img<const float> points, normals;
    Img<const uint8_t> seg;
    Img<uint8_t> image;  // u8 x 3
    int w, h;
    V3 light;
    uint8_t colors[256 * 3];
};

__device__ __forceinline__ float fastpow(float base, int exp) {  // EMFusion.cu:100-113
    float result = 1.f;
    while (exp) {
        if (exp & 1) result *= base;
        base *= base;
        exp >>= 1;
    }
    return result;
}

// static_cast<uchar>(float) is undefined for values outside [0, 256) in the reference; values that
// can occur are < 1 * 255 (the three coefficients sum to 1) and slightly negative ones where the
// normal faces away from the light: those and NaN become 0 here
__device__ __forceinline__ uint8_t to_u8(float v) {
    return v >= 0.f ? static_cast<uint8_t>(v < 255.f ? static_cast<int>(v) : 255) : uint8_t{0};
}

__global__ __launch_bounds__(256) void k_render_phong(const PhongArgs a) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.h) return;
    const float* pp = a.points.row(y) + 3 * x;
    const float* np = a.normals.row(y) + 3 * x;
    const V3 p = v3(pp[0], pp[1], pp[2]), n = v3(np[0], np[1], np[2]);
    uint8_t* out = a.image.row(y) + 3 * x;
    if (p.x == 0.f && p.y == 0.f && p.z == 0.f) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    const uint8_t* c = a.colors + 3 * a.seg.row(y)[x];
    const float ka = 0.3f, kd = 0.5f, ks = 0.2f;
    const V3 Rd = v3(static_cast<float>(c[0]) / 255.f, static_cast<float>(c[1]) / 255.f,
                     static_cast<float>(c[2]) / 255.f);
    V3 l = v3(a.light.x - p.x, a.light.y - p.y, a.light.z - p.z);
    l = l / norm(l);
    const V3 v = v3(-p.x, -p.y, -p.z) / norm(p);
    const V3 two = n * (2.f * dot(l, n));
    V3 r = v3(two.x - l.x, two.y - l.y, two.z - l.z);
    r = r / norm(r);
    const float diff = dot(n, l), spec = fastpow(dot(r, v), 20);
    // ka * Ra + kd * Rd * dot(n, l) + ks * Rs * pow, Ra = Rs = (1, 1, 1), summed left to right
    const V3 I = v3(ka * 1.f + (kd * Rd.x) * diff + (ks * 1.f) * spec, ka * 1.f + (kd * Rd.y) * diff + (ks * 1.f) * spec,
                    ka * 1.f + (kd * Rd.z) * diff + (ks * 1.f) * spec);
    out[0] = to_u8(I.x * 255.f);
    out[1] = to_u8(I.y * 255.f);
    out[2] = to_u8(I.z * 255.f);
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

namespace {

int fill_map_table(MapTable& t, const emf_image_t* maps, int count, const emf_image_t* ref,
                   const char* what) {
    t.count = count;
    for (int k = 0; k < count; ++k) {
        EMF_TRY(check_image(&maps[k], 4, what));
        EMF_TRY(check_same_size(&maps[k], ref, what, "maps[0]"));
        t.m[k] = img<float>(&maps[k]);
    }
    return EMF_OK;
}

}  // namespace

extern "C" {

int emf_hip_computePoints(const emf_image_t* depth, const emf_image_t* points, const float K[9],
                          emf_stream_t stream) {
    EMF_TRY(check_image(depth, 4, "computePoints: depth"));
    EMF_TRY(check_image(points, 12, "computePoints: points"));
    EMF_TRY(check_same_size(depth, points, "depth", "points"));
    EMF_REQUIRE_PTR(K);
    hipLaunchKernelGGL(k_compute_points, pixel_grid(depth->width, depth->height), pixel_block(), 0,
                       as_stream(stream), img<const float>(depth), img<float>(points),
                       depth->width, depth->height, K[0], K[4], K[2], K[5]);
    return launch_status("computePoints");
}

int emf_hip_getVolumeVals(const float* vol, int channels, const emf_image_t* points,
                          const float R_CO[9], const float t_CO[3], const int32_t res[3],
                          float voxelSize, const emf_image_t* vals, emf_stream_t stream) {
    EMF_REQUIRE_PTR(vol);
    if (channels < 1 || channels > 3)
        return fail(EMF_E_ARG, "getVolumeVals: channels = %d, expected 1..3", channels);
    EMF_TRY(check_image(points, 12, "getVolumeVals: points"));
    EMF_TRY(check_image(vals, 4 * static_cast<size_t>(channels), "getVolumeVals: vals"));
    EMF_TRY(check_same_size(points, vals, "points", "vals"));
    EMF_REQUIRE_PTR(R_CO);
    EMF_REQUIRE_PTR(t_CO);
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f)) return fail(EMF_E_ARG, "getVolumeVals: voxelSize must be > 0");
    LookupArgs a;
    a.vol = vol;
    a.points = img<const float>(points);
    a.vals = img<float>(vals);
    a.w = points->width;
    a.h = points->height;
    a.R = m33_from(R_CO);
    a.t = v3_from(t_CO);
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    const dim3 g = pixel_grid(a.w, a.h), b = pixel_block();
    if (channels == 1) hipLaunchKernelGGL(k_volume_vals<1>, g, b, 0, as_stream(stream), a);
    if (channels == 2) hipLaunchKernelGGL(k_volume_vals<2>, g, b, 0, as_stream(stream), a);
    if (channels == 3) hipLaunchKernelGGL(k_volume_vals<3>, g, b, 0, as_stream(stream), a);
    return launch_status("getVolumeVals");
}

int emf_hip_computeAssociation(const float* tsdf, const float* fgProbs, const emf_image_t* points,
                               const float R_CO[9], const float t_CO[3], const int32_t res[3],
                               float voxelSize, float truncdist, float assocSigma, float alpha,
                               float uniPrior, const emf_image_t* out, emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_TRY(check_image(points, 12, "computeAssociation: points"));
    EMF_TRY(check_image(out, 4, "computeAssociation: out"));
    EMF_TRY(check_same_size(points, out, "points", "out"));
    EMF_REQUIRE_PTR(R_CO);
    EMF_REQUIRE_PTR(t_CO);
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f) || !(assocSigma > 0.f))
        return fail(EMF_E_ARG, "computeAssociation: voxelSize %g / assocSigma %g must be > 0",
                    voxelSize, assocSigma);
    AssocArgs a;
    a.m.tsdf = tsdf;
    a.m.fgProbs = fgProbs;
    a.points = img<const float>(points);
    a.out = img<float>(out);
    a.w = points->width;
    a.h = points->height;
    a.m.R = m33_from(R_CO);
    a.m.t = v3_from(t_CO);
    a.m.n = i3_from(res);
    a.m.voxelSize = voxelSize;
    a.m.c1 = -truncdist / assocSigma;
    a.m.c2 = 1.f / (2.f * assocSigma);
    a.m.alpha = alpha;
    a.m.c3 = (1 - alpha) * uniPrior;
    hipLaunchKernelGGL(k_assoc, pixel_grid(a.w, a.h), pixel_block(), 0, as_stream(stream), a);
    return launch_status("computeAssociation");
}

int emf_hip_normalizeAssociation(const emf_image_t* maps_host, int nmaps, int nsum,
                                 const emf_image_t* extraSum, const emf_image_t* norm,
                                 emf_stream_t stream) {
    EMF_REQUIRE_PTR(maps_host);
    if (nmaps < 1 || nmaps > EMF_MAX_MODELS)
        return fail(EMF_E_LIMIT, "normalizeAssociation: nmaps = %d, expected 1..%d", nmaps,
                    EMF_MAX_MODELS);
    if (nsum < 0 || nsum > nmaps)
        return fail(EMF_E_ARG, "normalizeAssociation: nsum = %d, expected 0..nmaps (%d)", nsum,
                    nmaps);
    if (nsum == 0 && !extraSum)
        return fail(EMF_E_ARG, "normalizeAssociation: nsum == 0 needs extraSum");
    EMF_TRY(check_image(&maps_host[0], 4, "normalizeAssociation: maps[0]"));
    const int w = maps_host[0].width, h = maps_host[0].height;
    Img<const float> extra{nullptr, 0};
    if (extraSum) {
        EMF_TRY(check_image(extraSum, 4, "normalizeAssociation: extraSum"));
        EMF_TRY(check_same_size(extraSum, &maps_host[0], "extraSum", "maps[0]"));
        extra = img<const float>(extraSum);
    }
    Img<float> nrm{nullptr, 0};
    if (norm) {
        EMF_TRY(check_image(norm, 4, "normalizeAssociation: norm"));
        EMF_TRY(check_same_size(norm, &maps_host[0], "norm", "maps[0]"));
        nrm = img<float>(norm);
    }
    const dim3 g = pixel_grid(w, h), b = pixel_block();
    if (nmaps <= kMapsPerLaunch) {
        MapTable t;
        EMF_TRY(fill_map_table(t, maps_host, nmaps, &maps_host[0], "normalizeAssociation: maps"));
        hipLaunchKernelGGL(k_assoc_normalize, g, b, 0, as_stream(stream), t, nsum, extra, nrm, w,
                           h);
        return launch_status("normalizeAssociation");
    }
    // more maps than one launch carries: the running sum lives in `norm`, which is then required
    if (!norm)
        return fail(EMF_E_ARG, "normalizeAssociation: norm is required when nmaps > %d",
                    kMapsPerLaunch);
    if (nsum == 0) {
        hipLaunchKernelGGL(k_copy_map, g, b, 0, as_stream(stream), extra, nrm, w, h);
    }
    for (int k0 = 0; k0 < nsum; k0 += kMapsPerLaunch) {
        const int cnt = nsum - k0 < kMapsPerLaunch ? nsum - k0 : kMapsPerLaunch;
        MapTable t;
        EMF_TRY(fill_map_table(t, maps_host + k0, cnt, &maps_host[0], "normalizeAssociation: maps"));
        const bool lastChunk = k0 + cnt == nsum;
        hipLaunchKernelGGL(k_assoc_sum, g, b, 0, as_stream(stream), t, nrm,
                           lastChunk ? extra : Img<const float>{nullptr, 0}, w, h, k0 != 0);
    }
    for (int k0 = 0; k0 < nmaps; k0 += kMapsPerLaunch) {
        const int cnt = nmaps - k0 < kMapsPerLaunch ? nmaps - k0 : kMapsPerLaunch;
        MapTable t;
        EMF_TRY(fill_map_table(t, maps_host + k0, cnt, &maps_host[0], "normalizeAssociation: maps"));
        hipLaunchKernelGGL(k_assoc_divide, g, b, 0, as_stream(stream), t,
                           Img<const float>{nrm.data, nrm.pitch}, w, h);
    }
    return launch_status("normalizeAssociation");
}

int emf_hip_sumAssociation(const emf_image_t* maps_host, int nmaps, const emf_image_t* sum,
                           emf_stream_t stream) {
    EMF_REQUIRE_PTR(maps_host);
    if (nmaps < 1 || nmaps > EMF_MAX_MODELS)
        return fail(EMF_E_LIMIT, "sumAssociation: nmaps = %d, expected 1..%d", nmaps,
                    EMF_MAX_MODELS);
    EMF_TRY(check_image(sum, 4, "sumAssociation: sum"));
    EMF_TRY(check_image(&maps_host[0], 4, "sumAssociation: maps[0]"));
    EMF_TRY(check_same_size(sum, &maps_host[0], "sum", "maps[0]"));
    const int w = sum->width, h = sum->height;
    for (int k0 = 0; k0 < nmaps; k0 += kMapsPerLaunch) {
        const int cnt = nmaps - k0 < kMapsPerLaunch ? nmaps - k0 : kMapsPerLaunch;
        MapTable t;
        EMF_TRY(fill_map_table(t, maps_host + k0, cnt, &maps_host[0], "sumAssociation: maps"));
        hipLaunchKernelGGL(k_assoc_sum, pixel_grid(w, h), pixel_block(), 0, as_stream(stream), t,
                           img<float>(sum), Img<const float>{nullptr, 0}, w, h, k0 != 0);
    }
    return launch_status("sumAssociation");
}

namespace {
int composite_impl(int nobj, const int32_t* ids_host, const emf_image_t* objRay_host,
                   const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                   const emf_image_t* objSeg_host, const emf_image_t* bgRay,
                   const emf_image_t* bgVert, const emf_image_t* bgNorm,
                   const emf_image_t* bgMask, const emf_image_t* ray,
                   const emf_image_t* vert, const emf_image_t* norm,
                   const emf_image_t* seg, const emf_image_t* diff,
                   const emf_image_t* noObj, int boundary, int32_t* visCounts, bool fused,
                   emf_stream_t stream) {
    if (nobj < 0 || nobj > EMF_MAX_MODELS - 1)
        return fail(EMF_E_LIMIT, "compositeRaycast: nobj = %d, expected 0..%d", nobj,
                    EMF_MAX_MODELS - 1);
    if (nobj > 0) {
        EMF_REQUIRE_PTR(ids_host);
        EMF_REQUIRE_PTR(objRay_host);
        EMF_REQUIRE_PTR(objVert_host);
        EMF_REQUIRE_PTR(objNorm_host);
        EMF_REQUIRE_PTR(objSeg_host);
        EMF_REQUIRE_PTR(visCounts);
    }
    EMF_TRY(check_image(bgRay, 4, "compositeRaycast: bgRay"));
    EMF_TRY(check_image(bgVert, 12, "compositeRaycast: bgVert"));
    EMF_TRY(check_image(bgNorm, 12, "compositeRaycast: bgNorm"));
    EMF_TRY(check_image(bgMask, 1, "compositeRaycast: bgMask"));
    EMF_TRY(check_image(ray, 4, "compositeRaycast: ray"));
    EMF_TRY(check_image(vert, 12, "compositeRaycast: vert"));
    EMF_TRY(check_image(norm, 12, "compositeRaycast: norm"));
    EMF_TRY(check_image(seg, 1, "compositeRaycast: seg"));
    EMF_TRY(check_image(diff, 4, "compositeRaycast: diff"));
    EMF_TRY(check_image(noObj, 1, "compositeRaycast: noObj"));
    const emf_image_t* all[] = {bgVert, bgNorm, bgMask, ray, vert, norm, seg, diff, noObj};
    for (const emf_image_t* im : all) EMF_TRY(check_same_size(im, bgRay, "image", "bgRay"));
    if (boundary < 0) return fail(EMF_E_ARG, "compositeRaycast: boundary = %d < 0", boundary);
    const int w = bgRay->width, h = bgRay->height;

    CompositeArgs a;
    a.bgRay = img<const float>(bgRay);
    a.bgVert = img<const float>(bgVert);
    a.bgNorm = img<const float>(bgNorm);
    a.bgMask = img<const uint8_t>(bgMask);
    a.ray = img<float>(ray);
    a.vert = img<float>(vert);
    a.nrm = img<float>(norm);
    a.diff = img<float>(diff);
    a.seg = img<uint8_t>(seg);
    a.noObj = img<uint8_t>(noObj);
    a.w = w;
    a.h = h;
    a.counts = nullptr;
    a.boundary = boundary;
    for (int v = 0; v < 256; ++v) a.slots.slot[v] = -1;
    for (int k = 0; k < nobj; ++k)  // compare(seg, id): ids outside 1..255 never match
        if (ids_host[k] >= 1 && ids_host[k] <= 255 && a.slots.slot[ids_host[k]] < 0)
            a.slots.slot[ids_host[k]] = static_cast<int16_t>(k);
    const dim3 g = pixel_grid(w, h), b = pixel_block();
    int k0 = 0;
    do {
        const int cnt = nobj - k0 < kObjsPerLaunch ? nobj - k0 : kObjsPerLaunch;
        CompositeTable t;
        t.count = cnt;
        for (int k = 0; k < cnt; ++k) {
            const int j = k0 + k;
            EMF_TRY(check_image(&objRay_host[j], 4, "compositeRaycast: objRay"));
            EMF_TRY(check_image(&objVert_host[j], 12, "compositeRaycast: objVert"));
            EMF_TRY(check_image(&objNorm_host[j], 12, "compositeRaycast: objNorm"));
            EMF_TRY(check_image(&objSeg_host[j], 1, "compositeRaycast: objSeg"));
            EMF_TRY(check_same_size(&objRay_host[j], bgRay, "objRay", "bgRay"));
            EMF_TRY(check_same_size(&objVert_host[j], bgRay, "objVert", "bgRay"));
            EMF_TRY(check_same_size(&objNorm_host[j], bgRay, "objNorm", "bgRay"));
            EMF_TRY(check_same_size(&objSeg_host[j], bgRay, "objSeg", "bgRay"));
            t.ray[k] = img<const float>(&objRay_host[j]);
            t.vert[k] = img<const float>(&objVert_host[j]);
            t.nrm[k] = img<const float>(&objNorm_host[j]);
            t.seg[k] = img<const uint8_t>(&objSeg_host[j]);
            const int id = ids_host[j];
            t.id[k] = static_cast<uint8_t>(id < 0 ? 0 : (id > 255 ? 255 : id));
        }
        a.first = k0 == 0;
        a.last = k0 + cnt >= nobj;
        a.zero = a.first && !fused ? visCounts : nullptr;
        a.nzero = nobj;
        a.counts = fused && a.last && nobj > 0 ? visCounts : nullptr;
        hipLaunchKernelGGL(k_composite, g, b, 0, as_stream(stream), t, a);
        k0 += cnt;
    } while (k0 < nobj);
    EMF_TRY(launch_status("compositeRaycast"));
    if (nobj > 0 && !fused) {
        hipLaunchKernelGGL(k_vis_counts, g, b, 0, as_stream(stream), img<const uint8_t>(seg), w, h,
                           boundary, a.slots, visCounts);
        return launch_status("compositeRaycast: visibility");
    }
    return EMF_OK;
}
}  // namespace

int emf_hip_compositeRaycast(int nobj, const int32_t* ids_host, const emf_image_t* objRay_host,
                             const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                             const emf_image_t* objSeg_host, const emf_image_t* bgRay,
                             const emf_image_t* bgVert, const emf_image_t* bgNorm,
                             const emf_image_t* bgMask, const emf_image_t* ray,
                             const emf_image_t* vert, const emf_image_t* norm,
                             const emf_image_t* seg, const emf_image_t* diff,
                             const emf_image_t* noObj, int boundary, int32_t* visCounts,
                             emf_stream_t stream) {
    return composite_impl(nobj, ids_host, objRay_host, objVert_host, objNorm_host, objSeg_host, bgRay, bgVert,
                          bgNorm, bgMask, ray, vert, norm, seg, diff, noObj, boundary, visCounts, false, stream);
}

int emf_hip_compositeVisibility(int nobj, const int32_t* ids_host, const emf_image_t* objRay_host,
                                const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                                const emf_image_t* objSeg_host, const emf_image_t* bgRay,
                                const emf_image_t* bgVert, const emf_image_t* bgNorm,
                                const emf_image_t* bgMask, const emf_image_t* ray,
                                const emf_image_t* vert, const emf_image_t* norm,
                                const emf_image_t* seg, const emf_image_t* diff,
                                const emf_image_t* noObj, int boundary, int32_t* visCounts,
                                int visibilityThresh, int32_t* visible_dev, int32_t* countsMirror,
                                emf_stream_t stream) {
    EMF_REQUIRE_PTR(visible_dev);
    EMF_TRY(composite_impl(nobj, ids_host, objRay_host, objVert_host, objNorm_host, objSeg_host, bgRay, bgVert,
                           bgNorm, bgMask, ray, vert, norm, seg, diff, noObj, boundary, visCounts, true, stream));
    hipLaunchKernelGGL(k_vis_flags_clear, dim3(ceil_div(nobj + 1, 64)), dim3(64), 0, as_stream(stream), visCounts,
                       nobj + 1, visibilityThresh, visible_dev, countsMirror);
    return launch_status("compositeVisibility");
}

int emf_hip_occludedMask(const emf_image_t* objSeg, const emf_image_t* seg, int id,
                         const emf_image_t* occluded, emf_stream_t stream) {
    EMF_TRY(check_image(objSeg, 1, "occludedMask: objSeg"));
    EMF_TRY(check_image(seg, 1, "occludedMask: seg"));
    EMF_TRY(check_image(occluded, 1, "occludedMask: occluded"));
    EMF_TRY(check_same_size(objSeg, seg, "objSeg", "seg"));
    EMF_TRY(check_same_size(objSeg, occluded, "objSeg", "occluded"));
    hipLaunchKernelGGL(k_occluded, pixel_grid(seg->width, seg->height), pixel_block(), 0,
                       as_stream(stream), img<const uint8_t>(objSeg), img<const uint8_t>(seg), id,
                       img<uint8_t>(occluded), seg->width, seg->height);
    return launch_status("occludedMask");
}

int emf_hip_preprocessDepth(const emf_image_t* depthRaw, const emf_image_t* depth, int kernelSize,
                            float sigmaDepth, float sigmaSpatial, emf_stream_t stream) {
    EMF_TRY(check_image(depthRaw, 4, "preprocessDepth: depthRaw"));
    EMF_TRY(check_image(depth, 4, "preprocessDepth: depth"));
    EMF_TRY(check_same_size(depthRaw, depth, "depthRaw", "depth"));
    if (depthRaw->data == depth->data) return fail(EMF_E_ARG, "preprocessDepth: in place is not supported");
    if (kernelSize < 1 || kernelSize > 2 * kBfMaxR + 1 || kernelSize % 2 == 0)
        return fail(EMF_E_ARG, "preprocessDepth: kernel size %d, expected odd and <= %d", kernelSize,
                    2 * kBfMaxR + 1);
    if (!(sigmaDepth > 0.f) || !(sigmaSpatial > 0.f))
        return fail(EMF_E_ARG, "preprocessDepth: sigmas must be > 0");
    BilateralArgs a;
    a.raw = img<const float>(depthRaw);
    a.out = img<float>(depth);
    a.w = depthRaw->width;
    a.h = depthRaw->height;
    a.ksz = kernelSize;
    a.ss = -0.5f / (sigmaSpatial * sigmaSpatial);
    a.sc = -0.5f / (sigmaDepth * sigmaDepth);
    hipLaunchKernelGGL(k_preprocess_depth,
                       dim3(static_cast<unsigned>(ceil_div(a.w, kBfX)), static_cast<unsigned>(ceil_div(a.h, kBfY))),
                       dim3(kBfX, kBfY), 0, as_stream(stream), a);
    return launch_status("preprocessDepth");
}

int emf_hip_hideLabel(const emf_image_t* segmentation, int id, const emf_image_t* vertices,
                      const emf_image_t* normals, const emf_image_t* bgVertices, const emf_image_t* bgNormals,
                      emf_stream_t stream) {
    EMF_TRY(check_image(segmentation, 1, "hideLabel: segmentation"));
    EMF_TRY(check_image(vertices, 12, "hideLabel: vertices"));
    EMF_TRY(check_image(normals, 12, "hideLabel: normals"));
    EMF_TRY(check_image(bgVertices, 12, "hideLabel: bgVertices"));
    EMF_TRY(check_image(bgNormals, 12, "hideLabel: bgNormals"));
    EMF_TRY(check_same_size(segmentation, vertices, "segmentation", "vertices"));
    EMF_TRY(check_same_size(segmentation, normals, "segmentation", "normals"));
    EMF_TRY(check_same_size(segmentation, bgVertices, "segmentation", "bgVertices"));
    EMF_TRY(check_same_size(segmentation, bgNormals, "segmentation", "bgNormals"));
    if (id < 1 || id > 255) return fail(EMF_E_ARG, "hideLabel: id %d is not a label", id);
    hipLaunchKernelGGL(k_hide_label, pixel_grid(segmentation->width, segmentation->height), pixel_block(), 0,
                       as_stream(stream), img<uint8_t>(segmentation), id, img<float>(vertices), img<float>(normals),
                       img<const float>(bgVertices), img<const float>(bgNormals), segmentation->width,
                       segmentation->height);
    return launch_status("hideLabel");
}

int emf_hip_renderPhong(const emf_image_t* vertices, const emf_image_t* normals,
                        const emf_image_t* segmentation, const uint8_t colorMap[768],
                        const float lightPos[3], const emf_image_t* image, emf_stream_t stream) {
    EMF_TRY(check_image(vertices, 12, "renderPhong: vertices"));
    EMF_TRY(check_image(normals, 12, "renderPhong: normals"));
    EMF_TRY(check_image(segmentation, 1, "renderPhong: segmentation"));
    EMF_TRY(check_image(image, 3, "renderPhong: image"));
    EMF_TRY(check_same_size(vertices, normals, "vertices", "normals"));
    EMF_TRY(check_same_size(vertices, segmentation, "vertices", "segmentation"));
    EMF_TRY(check_same_size(vertices, image, "vertices", "image"));
    EMF_REQUIRE_PTR(colorMap);
    EMF_REQUIRE_PTR(lightPos);
    PhongArgs a;
    a.points = img<const float>(vertices);
    a.normals = img<const float>(normals);
    a.seg = img<const uint8_t>(segmentation);
    a.image = img<uint8_t>(image);
    a.w = vertices->width;
    a.h = vertices->height;
    a.light = v3_from(lightPos);
    for (int i = 0; i < 768; ++i) a.colors[i] = colorMap[i];
    hipLaunchKernelGGL(k_render_phong,
                       dim3(static_cast<unsigned>(ceil_div(a.w, 32)), static_cast<unsigned>(ceil_div(a.h, 8))),
                       dim3(256), 0, as_stream(stream), a);
    return launch_status("renderPhong");
}

}  // extern "C"
