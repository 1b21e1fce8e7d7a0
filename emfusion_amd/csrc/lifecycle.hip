// lifecycle.hip -- device primitives of object creation / matching from instance masks
// (the part that needs no mesh): EMFusion::initNewObjVolume,
// computeValidPoints, filterPoints + transformPoints + computePercentiles, matchSegmentation
// (reference EMFusion.cpp:329-540, 797-825; EMFusion.cu:63-98).
//
// The reference compacts the masked points with thrust::copy_if (per channel), transforms them
// with cv::cuda::transformPoints, splits the channels, runs three full device-wide thrust::sort
// passes and downloads two columns -- to read two order statistics per axis.  An order statistic
// does not need the order: here the k-th smallest value of each axis is found by a radix SELECT
// over the float keys (4 passes of 8 bits, LDS histograms, a few hundred bytes of scratch), fused
// with the mask test, the validity test and the rigid transform, with no compaction and no
// sorted copy.  The selected values are elements of the data, so the result is bit-identical to
// sorting.  matchSegmentation's per-object compare / and / or / countNonZero chain (two host
// round trips per object) is one kernel that counts intersection and union for all objects.
#include "mesh_core.hpp"

namespace emf_hip {
namespace {

constexpr int kLcBlock = 256;

// order-preserving map float -> uint32 (negative floats reversed, positive offset)
__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct StatsArgs {
    Img<const float> points;   // camera-frame points, f32x3
    Img<const uint8_t> mask;   // instance mask, non-zero = inside
    int w, h;
    M33 R;                     // target frame <- camera
    V3 t;
    emf_point_stats_t* out;    // result (device)
    unsigned* scratch;         // [0] count, [1..6] selected prefixes, [7..12] residual ranks,
                               // [16 .. 16 + 6 * 256) histograms
    int pass;                  // 0..3: which byte of the key, from the top
};

constexpr int kHistBase = 16;

// transformed point of pixel i if it is inside the mask and valid (computeValidPoints: any
// coordinate non-zero, EMFusion.cpp:397-406)
__device__ __forceinline__ bool masked_point(const StatsArgs& a, size_t i, V3& p) {
    const int y = static_cast<int>(i / a.w), x = static_cast<int>(i - static_cast<size_t>(y) * a.w);
    if (a.mask.row(y)[x] == 0) return false;
    const float* q = a.points.row(y) + 3 * x;
    const V3 pc = v3(q[0], q[1], q[2]);
    if (pc.x == 0.f && pc.y == 0.f && pc.z == 0.f) return false;
    p = mul(a.R, pc) + a.t;  // cv::cuda::transformPoints (EMFusion.cpp:408-415)
    return true;
}

// pass 0 also counts the points; every pass histograms one key byte of the points whose higher
// bytes equal the selected prefix, separately for the 6 (axis, rank) selections
__global__ __launch_bounds__(kLcBlock) void k_stats_hist(const StatsArgs a) {
    __shared__ unsigned hist[6][256];
    for (int i = threadIdx.x; i < 6 * 256; i += kLcBlock) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const size_t n = static_cast<size_t>(a.w) * a.h;
    const int shift = 24 - 8 * a.pass;
    unsigned local = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kLcBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kLcBlock) {
        V3 p;
        if (!masked_point(a, i, p)) continue;
        ++local;
        const unsigned key[3] = {order_key(p.x), order_key(p.y), order_key(p.z)};
#pragma unroll
        for (int s = 0; s < 6; ++s) {  // s = 2 * axis + (0: p10, 1: p90)
            const unsigned k = key[s >> 1];
            const bool match = a.pass == 0 || (k >> (shift + 8)) == a.scratch[1 + s];
            if (match) atomicAdd(&hist[s][(k >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * 256; i += kLcBlock) {
        const unsigned v = (&hist[0][0])[i];
        if (v) atomicAdd(&a.scratch[kHistBase + i], v);
    }
    if (a.pass == 0 && local) atomicAdd(&a.scratch[0], local);
}

// ---- second source of the statistics: the vertex cloud of the object's iso-surface -----------------
// EMFusion::updateObj (EMFusion.cpp:827-863) takes the percentiles over obj.getMesh().cloud plus the
// new points.  The reference's marching cubes (TSDF.cu:855-1152) emits, per cube whose 8 voxels all
// pass the mask (weights > 0, and fgVolMask for objects: ObjTSDF.cpp:251-252), one vertex per edge
// whose end points differ in sign (popcount of edgeTable[class]), interpolated by vertexInterp --
// the triangle table only decides connectivity.  So the cloud needs no mesh: each cube streams its
// edge vertices into the same histograms.

struct MeshStatsArgs {
    MeshSource src;
    unsigned* scratch;
    int pass;
};

__global__ __launch_bounds__(kLcBlock) void k_stats_hist_mesh(const MeshStatsArgs a) {
    __shared__ unsigned hist[6][256];
    for (int i = threadIdx.x; i < 6 * 256; i += kLcBlock) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const I3 n = a.src.n;
    const size_t cubes = static_cast<size_t>(n.x - 1) * (n.y - 1) * (n.z - 1);
    const int shift = 24 - 8 * a.pass;
    const size_t sy = static_cast<size_t>(n.x), sz = sy * n.y;
    const V3 half = half_extent(n);
    unsigned local = 0;
    for (size_t c = static_cast<size_t>(blockIdx.x) * kLcBlock + threadIdx.x; c < cubes;
         c += static_cast<size_t>(gridDim.x) * kLcBlock) {
        const int x = static_cast<int>(c % (n.x - 1));
        const size_t r = c / (n.x - 1);
        const int y = static_cast<int>(r % (n.y - 1)), z = static_cast<int>(r / (n.y - 1));
        const size_t base = static_cast<size_t>(z) * sz + static_cast<size_t>(y) * sy + x;
        float val[8];
        V3 ps[8];
        bool valid = true;
        unsigned cls = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int dx, dy, dz;
            cube_corner(i, dx, dy, dz);
            const size_t idx = base + dx + dy * sy + dz * sz;
            valid = valid && a.src.weights[idx] > 0.f && (!a.src.fg || a.src.fg[idx] != 0);
            val[i] = a.src.tsdf[idx];
            cls |= (val[i] < 0.f ? 1u : 0u) << i;
            // ( x + 1 - ( volSize.x - 1 ) / 2.f ) * voxelSize  (TSDF.cu:945-968)
            ps[i] = v3((static_cast<float>(x + dx) - half.x) * a.src.voxelSize,
                       (static_cast<float>(y + dy) - half.y) * a.src.voxelSize,
                       (static_cast<float>(z + dz) - half.z) * a.src.voxelSize);
        }
        if (!valid || cls == 0u || cls == 255u) continue;
        constexpr int e0[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
        constexpr int e1[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            if ((((cls >> e0[e]) ^ (cls >> e1[e])) & 1u) == 0u) continue;  // = bit e of edgeTable[cls]
            const V3 p = vertex_interp(ps[e0[e]], ps[e1[e]], val[e0[e]], val[e1[e]]);
            ++local;
            const unsigned key[3] = {order_key(p.x), order_key(p.y), order_key(p.z)};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const unsigned k = key[s >> 1];
                const bool match = a.pass == 0 || (k >> (shift + 8)) == a.scratch[1 + s];
                if (match) atomicAdd(&hist[s][(k >> shift) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * 256; i += kLcBlock) {
        const unsigned v = (&hist[0][0])[i];
        if (v) atomicAdd(&a.scratch[kHistBase + i], v);
    }
    if (a.pass == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
        if ((threadIdx.x & 63) == 0 && local) atomicAdd(&a.scratch[0], local);
    }
}

// ---- kernel_copyValues (TSDF.cu:768-819) as a gather over the destination --------------------------
// dst(x', y', z') = src(x' + off) where that lies inside the source, else 0: the same result as the
// reference's setTo(0) + scatter, in one pass and without a separate clear.
struct CopyArgs {
    const float* src;
    float* dst;
    int channels;
    I3 off, srcRes, dstRes;
};

__global__ __launch_bounds__(kLcBlock) void k_copy_values(const CopyArgs a) {
    const size_t total = static_cast<size_t>(a.dstRes.x) * a.dstRes.y * a.dstRes.z;
    const size_t i = static_cast<size_t>(blockIdx.x) * kLcBlock + threadIdx.x;
    if (i >= total) return;
    const int x = static_cast<int>(i % a.dstRes.x);
    const size_t r = i / a.dstRes.x;
    const int y = static_cast<int>(r % a.dstRes.y), z = static_cast<int>(r / a.dstRes.y);
    const int sx = x + a.off.x, sy = y + a.off.y, sz = z + a.off.z;  // x_new = x - offset
    const bool in = sx >= 0 && sx < a.srcRes.x && sy >= 0 && sy < a.srcRes.y && sz >= 0 && sz < a.srcRes.z;
    const size_t si = (static_cast<size_t>(sz) * a.srcRes.y + sy) * a.srcRes.x + sx;
    for (int c = 0; c < a.channels; ++c) a.dst[i * a.channels + c] = in ? a.src[si * a.channels + c] : 0.f;
}

// one wave: pick, for each of the 6 selections, the bucket that holds the wanted rank; extend the
// prefix, reduce the rank, clear the histograms for the next pass; after the last pass write the
// result
__global__ __launch_bounds__(64) void k_stats_pick(const StatsArgs a) {
    const int s = threadIdx.x;
    const unsigned count = a.scratch[0];
    if (s < 6 && count > 0) {
        unsigned rank;
        if (a.pass == 0) {
            // sorted.col(static_cast<int>(points.cols * .1f)) / (.9f)  (EMFusion.cu:90-91)
            const float frac = (s & 1) ? .9f : .1f;
            rank = static_cast<unsigned>(static_cast<int>(static_cast<float>(count) * frac));
            if (rank >= count) rank = count - 1;
        } else {
            rank = a.scratch[7 + s];
        }
        const unsigned* h = a.scratch + kHistBase + 256 * s;
        unsigned acc = 0, b = 0;
        for (; b < 255u; ++b) {
            if (acc + h[b] > rank) break;
            acc += h[b];
        }
        const unsigned prefix = a.pass == 0 ? b : ((a.scratch[1 + s] << 8) | b);
        a.scratch[1 + s] = prefix;
        a.scratch[7 + s] = rank - acc;
        if (a.pass == 3) {
            const float v = key_value(prefix);
            if (s & 1) a.out->p90[s >> 1] = v;
            else a.out->p10[s >> 1] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * 256; i += 64) a.scratch[kHistBase + i] = 0u;
    if (threadIdx.x == 0 && a.pass == 3) {
        a.out->count = count;
        if (count == 0)
            for (int k = 0; k < 3; ++k) a.out->p10[k] = a.out->p90[k] = 0.f;
    }
}

__global__ void k_stats_clear(unsigned* scratch) {
    for (int i = threadIdx.x; i < kHistBase + 6 * 256; i += blockDim.x) scratch[i] = 0u;
}

// ---- mask / model-segmentation overlap (matchSegmentation, EMFusion.cpp:797-825) -------------------

struct IouArgs {
    Img<const uint8_t> seg;       // new instance mask (non-zero = inside)
    Img<const uint8_t> modelSeg;  // composite model segmentation (object ids, 0 = background)
    int w, h;
    unsigned* counts;             // [0] pixels of the mask; [1 + id] intersection with model id;
                                  // [257 + id] pixels of model id  (id 1..255)
};

__global__ __launch_bounds__(kLcBlock) void k_mask_overlap(const IouArgs a) {
    __shared__ unsigned inter[256], area[256], maskPx;
    for (int i = threadIdx.x; i < 256; i += kLcBlock) inter[i] = area[i] = 0u;
    if (threadIdx.x == 0) maskPx = 0u;
    __syncthreads();
    const size_t n = static_cast<size_t>(a.w) * a.h;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kLcBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kLcBlock) {
        const int y = static_cast<int>(i / a.w), x = static_cast<int>(i - static_cast<size_t>(y) * a.w);
        const bool in = a.seg.row(y)[x] != 0;
        const unsigned id = a.modelSeg.row(y)[x];
        if (in) atomicAdd(&maskPx, 1u);
        if (id) {
            atomicAdd(&area[id], 1u);
            if (in) atomicAdd(&inter[id], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kLcBlock) {
        if (inter[i]) atomicAdd(&a.counts[1 + i], inter[i]);
        if (area[i]) atomicAdd(&a.counts[257 + i], area[i]);
    }
    if (threadIdx.x == 0 && maskPx) atomicAdd(&a.counts[0], maskPx);
}

// ---- carving an object's footprint out of an unmatched mask (initObjsFromUnmatched) ----------------

struct CarveArgs {
    Img<uint8_t> seg;             // unmatched instance mask, modified in place
    Img<const uint8_t> modelSeg;  // composite model segmentation
    Img<const uint8_t> match;     // the mask matched to the object, or data == nullptr
    int id, w, h;
    unsigned* counts;             // [0] pixels before, [1] pixels after
};

// seg &= !((modelSeg == id) | match)   (EMFusion.cpp:462-478), counting |seg| before and after
__global__ __launch_bounds__(kLcBlock) void k_carve_mask(const CarveArgs a) {
    const size_t n = static_cast<size_t>(a.w) * a.h;
    unsigned pre = 0, post = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kLcBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kLcBlock) {
        const int y = static_cast<int>(i / a.w), x = static_cast<int>(i - static_cast<size_t>(y) * a.w);
        uint8_t& s = a.seg.row(y)[x];
        if (!s) continue;
        ++pre;
        const bool taken = a.modelSeg.row(y)[x] == a.id || (a.match.data && a.match.row(y)[x] != 0);
        if (taken) s = 0;
        else ++post;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pre += __shfl_xor(pre, o);
        post += __shfl_xor(post, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (pre) atomicAdd(&a.counts[0], pre);
        if (post) atomicAdd(&a.counts[1], post);
    }
}

// ---- association mass under an object's mask (cleanUpObjs, EMFusion.cpp:936-949) ----------------

struct MassArgs {
    Img<const uint8_t> objSeg;  // the object's own raycast mask (0 / 1)
    Img<const uint8_t> match;   // matched instance mask or data == nullptr
    Img<const float> assoc;     // the object's association weights
    int w, h;
    emf_mask_mass_t* out;
};

// Two launches, fixed summation order (deterministic).  cleanUpObjs runs in EVERY frame of the reference's entry point,
// once per visible object, with the host waiting for the answer: the one-workgroup form of rounds 3-5 (1024 lanes x 300
// pixels each, an integer division per pixel) took 199 us of a 2.1 ms tracked frame (round 6, kernel trace of
// `bench.py --track`).  Now kMassBlocks workgroups take bands of whole image rows (no division; a lane's pixels in row
// order, a wave's lanes by the xor tree, the block's waves in index order) and leave a partial each behind out[0];
// k_mask_mass_finish adds the partials in block order.
constexpr int kMassBlocks = 240;
__global__ __launch_bounds__(256) void k_mask_mass(const MassArgs a) {
    __shared__ double sums[4];
    __shared__ unsigned counts[4];
    const int rows = (a.h + kMassBlocks - 1) / kMassBlocks;
    const int y0 = blockIdx.x * rows, y1 = min(y0 + rows, a.h);
    double s = 0.0;
    unsigned c = 0;
    for (int y = y0; y < y1; ++y) {
        const uint8_t* seg = a.objSeg.row(y);
        const uint8_t* mt = a.match.data ? a.match.row(y) : nullptr;
        const float* as = a.assoc.row(y);
        for (int x = threadIdx.x; x < a.w; x += 256)
            if (seg[x] != 0 || (mt && mt[x] != 0)) {
                s += static_cast<double>(as[x]);
                ++c;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sums[threadIdx.x >> 6] = s;
        counts[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0;
        unsigned tc = 0;
        for (int i = 0; i < 4; ++i) {
            ts += sums[i];
            tc += counts[i];
        }
        a.out[1 + blockIdx.x].count = tc;
        a.out[1 + blockIdx.x].sum = ts;
    }
}
__global__ __launch_bounds__(64) void k_mask_mass_finish(emf_mask_mass_t* out) {
    double s = 0.0;
    unsigned c = 0;
    for (int b = threadIdx.x; b < kMassBlocks; b += 64) {  // (a lane's partials in block order, then the xor tree)
        s += out[1 + b].sum;
        c += out[1 + b].count;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    if (threadIdx.x == 0) {
        out[0].count = c;
        out[0].sum = s;
    }
}

__global__ void k_clear_u32(unsigned* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

size_t emf_hip_pointStatsScratchBytes(void) { return (kHistBase + 6 * 256) * sizeof(unsigned); }

namespace {
int point_stats(const emf_image_t* points, const emf_image_t* mask, const float R[9], const float t[3],
                const MeshSource* mesh, void* scratch_dev, emf_point_stats_t* stats_dev,
                emf_stream_t stream);
}

int emf_hip_maskedPointStats(const emf_image_t* points, const emf_image_t* mask, const float R[9],
                             const float t[3], void* scratch_dev, emf_point_stats_t* stats_dev,
                             emf_stream_t stream) {
    return point_stats(points, mask, R, t, nullptr, scratch_dev, stats_dev, stream);
}

int emf_hip_objectExtentStats(const emf_image_t* points, const emf_image_t* mask, const float R[9],
                              const float t[3], const float* tsdf, const float* weights,
                              const uint8_t* fgVolMask, const int32_t res[3], float voxelSize,
                              void* scratch_dev, emf_point_stats_t* stats_dev, emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_TRY(check_res(res));
    if (res[0] < 2 || res[1] < 2 || res[2] < 2) return fail(EMF_E_SHAPE, "objectExtentStats: volume too small");
    MeshSource m{tsdf, weights, fgVolMask, i3_from(res), voxelSize};
    return point_stats(points, mask, R, t, &m, scratch_dev, stats_dev, stream);
}

int emf_hip_copyValues(const float* src, float* dst, int channels, const int32_t offset[3],
                       const int32_t srcRes[3], const int32_t dstRes[3], emf_stream_t stream) {
    EMF_REQUIRE_PTR(src);
    EMF_REQUIRE_PTR(dst);
    EMF_REQUIRE_PTR(offset);
    EMF_TRY(check_res(srcRes));
    EMF_TRY(check_res(dstRes));
    if (channels < 1 || channels > 3) return fail(EMF_E_ARG, "copyValues: %d channels", channels);
    CopyArgs a{src, dst, channels, i3_from(offset), i3_from(srcRes), i3_from(dstRes)};
    const size_t total = static_cast<size_t>(dstRes[0]) * dstRes[1] * dstRes[2];
    hipLaunchKernelGGL(k_copy_values, dim3(static_cast<unsigned>(ceil_div(total, kLcBlock))), dim3(kLcBlock),
                       0, as_stream(stream), a);
    return launch_status("copyValues");
}

namespace {
int point_stats(const emf_image_t* points, const emf_image_t* mask, const float R[9], const float t[3],
                const MeshSource* mesh, void* scratch_dev, emf_point_stats_t* stats_dev,
                emf_stream_t stream) {
    EMF_TRY(check_image(points, 12, "maskedPointStats: points"));
    EMF_TRY(check_image(mask, 1, "maskedPointStats: mask"));
    EMF_TRY(check_same_size(points, mask, "points", "mask"));
    EMF_REQUIRE_PTR(R);
    EMF_REQUIRE_PTR(t);
    EMF_REQUIRE_PTR(scratch_dev);
    EMF_REQUIRE_PTR(stats_dev);
    StatsArgs a;
    a.points = img<const float>(points);
    a.mask = img<const uint8_t>(mask);
    a.w = points->width;
    a.h = points->height;
    a.R = m33_from(R);
    a.t = v3_from(t);
    a.out = stats_dev;
    a.scratch = static_cast<unsigned*>(scratch_dev);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_stats_clear, dim3(1), dim3(256), 0, s, a.scratch);
    const size_t n = static_cast<size_t>(a.w) * a.h;
    const unsigned blocks = static_cast<unsigned>(ceil_div(n, kLcBlock * 4));
    MeshStatsArgs ma{};
    unsigned meshBlocks = 0;
    if (mesh) {
        ma.src = *mesh;
        ma.scratch = a.scratch;
        const size_t cubes = static_cast<size_t>(mesh->n.x - 1) * (mesh->n.y - 1) * (mesh->n.z - 1);
        meshBlocks = static_cast<unsigned>(ceil_div(cubes, kLcBlock * 4));
    }
    for (int pass = 0; pass < 4; ++pass) {
        a.pass = pass;
        hipLaunchKernelGGL(k_stats_hist, dim3(blocks), dim3(kLcBlock), 0, s, a);
        if (mesh) {
            ma.pass = pass;
            hipLaunchKernelGGL(k_stats_hist_mesh, dim3(meshBlocks), dim3(kLcBlock), 0, s, ma);
        }
        hipLaunchKernelGGL(k_stats_pick, dim3(1), dim3(64), 0, s, a);
    }
    return launch_status("maskedPointStats");
}
}  // namespace

int emf_hip_carveMask(const emf_image_t* seg, const emf_image_t* modelSeg, int id,
                      const emf_image_t* matchMask, uint32_t* counts_dev, emf_stream_t stream) {
    EMF_TRY(check_image(seg, 1, "carveMask: seg"));
    EMF_TRY(check_image(modelSeg, 1, "carveMask: modelSeg"));
    EMF_TRY(check_same_size(seg, modelSeg, "seg", "modelSeg"));
    if (matchMask) {
        EMF_TRY(check_image(matchMask, 1, "carveMask: matchMask"));
        EMF_TRY(check_same_size(seg, matchMask, "seg", "matchMask"));
    }
    EMF_REQUIRE_PTR(counts_dev);
    if (id < 1 || id > 255) return fail(EMF_E_ARG, "carveMask: object id %d outside 1..255", id);
    CarveArgs a;
    a.seg = img<uint8_t>(seg);
    a.modelSeg = img<const uint8_t>(modelSeg);
    a.match = matchMask ? img<const uint8_t>(matchMask) : Img<const uint8_t>{nullptr, 0};
    a.id = id;
    a.w = seg->width;
    a.h = seg->height;
    a.counts = counts_dev;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_clear_u32, dim3(1), dim3(64), 0, s, counts_dev, 2);
    const size_t n = static_cast<size_t>(a.w) * a.h;
    hipLaunchKernelGGL(k_carve_mask, dim3(static_cast<unsigned>(ceil_div(n, kLcBlock * 4))),
                       dim3(kLcBlock), 0, s, a);
    return launch_status("carveMask");
}

int emf_hip_maskAssociationMass(const emf_image_t* objSeg, const emf_image_t* matchMask,
                                const emf_image_t* assoc, emf_mask_mass_t* out_dev,
                                emf_stream_t stream) {
    EMF_TRY(check_image(objSeg, 1, "maskAssociationMass: objSeg"));
    EMF_TRY(check_image(assoc, 4, "maskAssociationMass: assoc"));
    EMF_TRY(check_same_size(objSeg, assoc, "objSeg", "assoc"));
    if (matchMask) {
        EMF_TRY(check_image(matchMask, 1, "maskAssociationMass: matchMask"));
        EMF_TRY(check_same_size(objSeg, matchMask, "objSeg", "matchMask"));
    }
    EMF_REQUIRE_PTR(out_dev);
    MassArgs a;
    a.objSeg = img<const uint8_t>(objSeg);
    a.match = matchMask ? img<const uint8_t>(matchMask) : Img<const uint8_t>{nullptr, 0};
    a.assoc = img<const float>(assoc);
    a.w = objSeg->width;
    a.h = objSeg->height;
    a.out = out_dev;
    hipLaunchKernelGGL(k_mask_mass, dim3(kMassBlocks), dim3(256), 0, as_stream(stream), a);
    hipLaunchKernelGGL(k_mask_mass_finish, dim3(1), dim3(64), 0, as_stream(stream), out_dev);
    return launch_status("maskAssociationMass");
}

size_t emf_hip_maskAssociationMassBytes(void) { return (1 + static_cast<size_t>(kMassBlocks)) * sizeof(emf_mask_mass_t); }

int emf_hip_maskOverlap(const emf_image_t* seg, const emf_image_t* modelSeg, uint32_t* counts_dev,
                        emf_stream_t stream) {
    EMF_TRY(check_image(seg, 1, "maskOverlap: seg"));
    EMF_TRY(check_image(modelSeg, 1, "maskOverlap: modelSeg"));
    EMF_TRY(check_same_size(seg, modelSeg, "seg", "modelSeg"));
    EMF_REQUIRE_PTR(counts_dev);
    IouArgs a;
    a.seg = img<const uint8_t>(seg);
    a.modelSeg = img<const uint8_t>(modelSeg);
    a.w = seg->width;
    a.h = seg->height;
    a.counts = counts_dev;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_clear_u32, dim3(1), dim3(256), 0, s, counts_dev, 513);
    const size_t n = static_cast<size_t>(a.w) * a.h;
    hipLaunchKernelGGL(k_mask_overlap, dim3(static_cast<unsigned>(ceil_div(n, kLcBlock * 4))),
                       dim3(kLcBlock), 0, s, a);
    return launch_status("maskOverlap");
}

}  // extern "C"
