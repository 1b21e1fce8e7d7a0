// batched.hip -- level-3 ABI: one launch per stage for ALL models of a rank (background + objects),
// driven by a device-resident model table (emf_model_t[]) plus per-launch poses passed by value.
//
// The reference fans every stage out over one CUDA stream per volume (EMFusion.h:471) and
// synchronises the host between stages; on MI355X a frame is a handful of ~10-500 us kernels, so
// launch latency and host round trips would dominate.  Here each stage is a single grid:
//   k_estep             (pixel x model) lanes, likelihoods meet in LDS, normalised in the same kernel
//   k_raycast_batched   all models' 16x16 pixel tiles in one grid, XCD-aware tile order, outputs
//                       zero-filled by the kernel itself (no memsets)
//   k_integrate_batched all models' 32x8x8 voxel tiles in one grid, visibility gate read on device
// All arithmetic comes from device_core.hpp, i.e. it is the same code the per-volume kernels run.
#include <cstdlib>

#include "march_wave.hpp"
#include "peer_core.hpp"

namespace emf_hip {
namespace {

struct PoseTable {
    emf_pose_t p[EMF_MAX_BATCH];
};

__device__ __forceinline__ M33 pose_R(const emf_pose_t& p) {
    return M33{{p.R[0], p.R[1], p.R[2]}, {p.R[3], p.R[4], p.R[5]}, {p.R[6], p.R[7], p.R[8]}};
}
__device__ __forceinline__ V3 pose_t(const emf_pose_t& p) { return V3{p.t[0], p.t[1], p.t[2]}; }

// ---- fused E-step ---------------------------------------------------------------------------------

// workgroup shape, measured on the 5-model bench frame (us per E-step): 64 x 4: 13.9, 128 x 4: 14.0,
// 256 x 4: 15.0, 64 x 5: 14.9, 64 x 8: 16.8, 128 x 8: 17.6
// (round 5, measured and dropped: an XCD-banded grid -- block b on XCD b % 8 takes a band of image rows, so that the voxel
// lines neighbouring pixel rows share are fetched into one L2 instead of eight (33 MB from HBM per launch at an L2 hit rate
// of 0.31) -- 18.0 -> 19.3 us per E-step: the launch is not bound by those fetches, and the bands concentrate each
// XCD's gathers on fewer memory channels)
#ifndef EMF_ESTEP_PIXELS
#define EMF_ESTEP_PIXELS 64
#define EMF_ESTEP_LANES 4
#endif
constexpr int kEstepPixels = EMF_ESTEP_PIXELS;  // pixels per workgroup (a multiple of 64: waves stay model-uniform)
constexpr int kEstepLanes = EMF_ESTEP_LANES;    // model lanes per workgroup

struct EstepArgs {
    const emf_model_t* models;
    PoseTable poses;  // camera -> volume
    int nmodels;
    Img<const float> points;
    Img<float> norm, objSum;
    int w, h;
    int normalize;
    // the first E-step of a frame: points from depth on the way (k_compute_points' arithmetic), stored too
    Img<const float> depth;
    Img<float> pointsOut;
    float fx, fy, cx, cy;
    // sharded path over the direct peer-write transport (emf_hip_estepBatchedPeer): the per-pixel sum of the object
    // maps is this rank's contribution to the E-step's exchange and goes straight into its slot on every peer
    char* peerSlots[EMF_MAX_PEERS];
    int peerWorld;    // 0: objSum is an ordinary image
    int peerFences;   // emf_peer_t::systemFences (ranks on distinct devices): the stores are followed by a system-scope fence
    size_t peerOff;   // byte offset of this rank's slot (parity included) in a peer's receive buffer
};

__global__ __launch_bounds__(kEstepPixels* kEstepLanes) void k_estep(const EstepArgs a) {
    __shared__ float wl[EMF_MAX_BATCH][kEstepPixels];
    const int tx = threadIdx.x;
    const int x = blockIdx.x * kEstepPixels + tx, y = blockIdx.y;
    const bool inside = x < a.w;
    V3 pc = v3(0.f, 0.f, 0.f);
    if (inside) {
        if (a.depth.data) {  // (uniform) reference EMFusion.cu:39-46
            const float d = a.depth.row(y)[x];
            pc = v3((static_cast<float>(x) - a.cx) * d / a.fx, (static_cast<float>(y) - a.cy) * d / a.fy, d);
            if (threadIdx.y == 0) {
                float* po = a.pointsOut.row(y) + 3 * x;
                po[0] = pc.x;
                po[1] = pc.y;
                po[2] = pc.z;
            }
        } else {
            const float* pp = a.points.row(y) + 3 * x;
            pc = v3(pp[0], pp[1], pp[2]);
        }
    }
    // threadIdx.y is wave-uniform (64 x-lanes = one wave): keep the model index scalar
    for (int m = __builtin_amdgcn_readfirstlane(threadIdx.y); m < a.nmodels; m += kEstepLanes) {
        const emf_model_t& md = a.models[m];
        AssocModel am;
        am.tsdf = md.tsdf;
        am.fgProbs = md.fgProbs;
        am.R = pose_R(a.poses.p[m]);
        am.t = pose_t(a.poses.p[m]);
        am.n = I3{md.res[0], md.res[1], md.res[2]};
        am.voxelSize = md.voxelSize;
        am.c1 = md.assocC1;
        am.c2 = md.assocC2;
        am.alpha = md.alpha;
        am.c3 = md.assocC3;
        wl[m][tx] = inside ? assoc_weight(am, pc) : 0.f;
    }
    __syncthreads();
    if (!inside) return;
    const size_t pix = static_cast<size_t>(y) * a.w + x;
    if (a.normalize) {
        // sequential sum: background first, then objects in table (= ascending id) order,
        // exactly the order of the reference's add chain (EMFusion.cpp:654-657)
        float s = wl[0][tx];
        for (int m = 1; m < a.nmodels; ++m) s = s + wl[m][tx];
        for (int m = threadIdx.y; m < a.nmodels; m += kEstepLanes)
            a.models[m].assoc[pix] = (s != 0.f) ? wl[m][tx] / s : 0.f;  // x / 0 := 0 (Q7)
        if (threadIdx.y == 0 && a.norm.data) a.norm.row(y)[x] = s;
    } else {
        for (int m = threadIdx.y; m < a.nmodels; m += kEstepLanes)
            a.models[m].assoc[pix] = wl[m][tx];
        if (threadIdx.y == 0) {
            float s = 0.f;
            if (a.nmodels > 1) {
                s = wl[1][tx];
                for (int m = 2; m < a.nmodels; ++m) s = s + wl[m][tx];
            }
            if (a.peerWorld) {  // (uniform) write-through: the value must be in the peer's memory, not in my L2
                for (int p = 0; p < a.peerWorld; ++p)
                    __builtin_nontemporal_store(s, reinterpret_cast<float*>(a.peerSlots[p] + a.peerOff) + pix);
                if (a.peerFences) __threadfence_system();
            } else if (a.objSum.data) {  // (uniform; a chunk of a longer model list hands out no sum)
                a.objSum.row(y)[x] = s;
            }
        }
    }
}

// Normalisation of ALL maps of a model table in one launch (model lists longer than EMF_MAX_BATCH, whose chunks' E-step
// launches leave un-normalised likelihoods): the reference's add chain -- background, then the objects in table order
// (EMFusion.cpp:654-657) -- one pixel per lane, the maps' values fetched eight at a time ahead of the dependent adds; then
// every map divided (x / 0 := 0).  What emf_hip_normalizeAssociation(nsum = nmaps) computes in ceil(n / 16) + ceil(n / 16)
// launches with the map views in its kernel arguments; here the views are the table's `assoc` pointers.
__global__ __launch_bounds__(256) void k_assoc_normalize_table(const emf_model_t* __restrict__ models, int n,
                                                               float* __restrict__ norm, size_t pixels) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= pixels) return;
    float s = models[0].assoc[i];
    int m = 1;
    for (; m + 8 <= n; m += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = models[m + j].assoc[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s = s + v[j];
    }
    for (; m < n; ++m) s = s + models[m].assoc[i];
    for (m = 0; m + 8 <= n; m += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = models[m + j].assoc[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) models[m + j].assoc[i] = (s != 0.f) ? v[j] / s : 0.f;
    }
    for (; m < n; ++m) {
        const float v = models[m].assoc[i];
        models[m].assoc[i] = (s != 0.f) ? v / s : 0.f;
    }
    if (norm) norm[i] = s;
}

// ---- batched raycast -------------------------------------------------------------------------------

struct RaycastBatchArgs {
    const emf_model_t* models;
    PoseTable poses;  // camera -> volume
    int nmodels;
    int w, h;
    int tilesX, tilesY;
    int chunk;  // tiles per XCD = ceil(tilesX * tilesY / 8)
    float fx, fy, cx, cy;
    unsigned divideMask;  // bit m: model m must divide by voxelSize (pose out of the checked range)
    int firstObj;    // 1: slot 0 is the background; 0: every slot is an object (a later chunk of a model list longer than EMF_MAX_BATCH)
    int bandTile0, bandTiles;  // slot 0 only: tile rows this rank marches (bandTiles == 0: all)
    unsigned long long* stats;
    const float* farBounds;    // [model][2 tilesY][2 tilesX] per 8x8-pixel cell, or nullptr (see k_far_bounds)
    // objects (slots 1..): the tiles their volume box can project to (host-computed from the pose); only
    // those get a marching workgroup, the rest of the object's images is zero-filled 16 tiles per workgroup
    short rect[EMF_MAX_BATCH][4];      // tx0, ty0, width, height in tiles (the background's slot unused)
    int objStart[EMF_MAX_BATCH + 1];   // prefix sum of width * height over the object slots; [m] = first block of slot m
};
constexpr int kZeroTiles = 16;  // tiles per zero-fill workgroup

// 4 waves (8x8-pixel sub-tiles) per workgroup = a 16x16 tile; 1-wave workgroups measured the same
constexpr int kRbWaves = 4;
constexpr int kRbTile = 16;

#ifdef EMF_RAY_TRACE  // timeline instrumentation, trace builds only (scripts/raycast_timeline.py)
struct RayTraceRec {
    unsigned long long t0, t1;
    unsigned hw, xcc, model, tile, wave, samplesMax, samplesSum, activeLanes;
#ifdef EMF_MARCH_STAMP
    unsigned long long ckIssue, ckWait, ckRest, lateIssue, lateWait, lateRest;
    unsigned iters, hist[6], pad;
#endif
};
__device__ RayTraceRec g_rayTrace[32768];
__device__ __forceinline__ void trace_wave(unsigned long long t0, unsigned samples, int m, int tile,
                                           int wave, int lane, const MarchCount* mc = nullptr) {
    unsigned mx = samples, sm = samples;
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, static_cast<unsigned>(__shfl_xor(static_cast<int>(mx), o)));
        sm += static_cast<unsigned>(__shfl_xor(static_cast<int>(sm), o));
    }
    const unsigned long long lanes = __ballot(samples > 0);
    const unsigned slot = blockIdx.x * 4 + wave;
    if (lane == 0 && slot < 32768u) {
        RayTraceRec t;
        t.t0 = t0;
        t.t1 = wall_clock64();
        t.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        t.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        t.model = m;
        t.tile = tile;
        t.wave = wave;
        t.samplesMax = mx;
        t.samplesSum = sm;
        t.activeLanes = __popcll(lanes);
#ifdef EMF_MARCH_STAMP
        t.ckIssue = t.ckWait = t.ckRest = t.lateIssue = t.lateWait = t.lateRest = 0;
        t.iters = t.pad = 0;
        for (int k = 0; k < 6; ++k) t.hist[k] = 0;
        if (mc) {  // (wave-uniform values: any lane that marched holds them; lanes that did not hold zeros)
            t.ckIssue = mc->ckIssue; t.ckWait = mc->ckWait; t.ckRest = mc->ckRest; t.iters = mc->iters;
            t.lateIssue = mc->lateIssue; t.lateWait = mc->lateWait; t.lateRest = mc->lateRest;
            for (int k = 0; k < 6; ++k) t.hist[k] = mc->hist[k];
        }
#endif
        g_rayTrace[slot] = t;
    }
}
#else
__device__ __forceinline__ void trace_wave(unsigned long long, unsigned, int, int, int, int) {}
#endif

// Which tile of which model does workgroup `bx` of the batched raycast grid serve?  Order: the background's BORDER
// tiles first (their rays graze seen / unseen space at half-voxel steps all the way: the longest marches of the
// image), then the objects' footprints, then the background's interior -- XCD-banded: block b runs on XCD b % 8
// (observed dispatch order; used for L2 locality only) and each XCD gets a contiguous run of tiles in raster
// order, i.e. a horizontal band of the image, so the voxels its rays walk stay in that XCD's 4 MiB L2 -- then the
// zero-fill of the objects' images outside their footprints.  (Measured, frames/s of the bench: background first
// 1227, objects first 1340, this order 1385.)
// PARTS > 1 (lanes-per-ray march): a BACKGROUND tile is served by PARTS workgroups (`sub` = which part), consecutive
// on one XCD; object tiles and zero-fill keep one workgroup each (their rays are short: median 12 samples -- and the
// launch is paced by the workgroup dispatcher as soon as waves live only microseconds: with PARTS workgroups per
// object tile the background's interior started at 210 us instead of 23, round 5).  Every region starts at a multiple
// of 8 blocks.
// Returns 1: march (m, tile, sub); 2: zero-fill workgroup `tile` (index over all objects); 0: nothing to do.
struct RaycastGrid {
    int ring, ringBlocks, objBlocks, objPad, bgTileBlocks, bgBlocks, zeroBlocks;
};
__host__ __device__ __forceinline__ RaycastGrid raycast_grid(int tilesX, int tilesY, int bandTiles, int chunk, int objBlocks,
                                                           int nmodels, int parts, int firstObj) {
    RaycastGrid g;
    g.ring = (firstObj && tilesX > 2 && tilesY > 2 && bandTiles == 0) ? 2 * tilesX + 2 * (tilesY - 2) : 0;
    g.ringBlocks = (g.ring * parts + 7) / 8 * 8;
    g.objBlocks = objBlocks;
    g.objPad = (objBlocks + 7) / 8 * 8;
    g.bgTileBlocks = !firstObj ? 0 : g.ring ? 8 * (((tilesX - 2) * (tilesY - 2) + 7) / 8) : 8 * chunk;
    g.bgBlocks = g.bgTileBlocks * parts;
    g.zeroBlocks = (nmodels - firstObj) * ((tilesX * tilesY + kZeroTiles - 1) / kZeroTiles);
    return g;
}

template <int PARTS>
__device__ __forceinline__ int raycast_block_role(const RaycastBatchArgs& a, int bx, int& m, int& tile, int& sub) {
    const RaycastGrid g = raycast_grid(a.tilesX, a.tilesY, a.bandTiles, a.chunk, a.objStart[a.nmodels], a.nmodels, PARTS, a.firstObj);
    sub = 0;
    if (bx < g.ringBlocks) {
        const int b = bx / PARTS;
        sub = bx - b * PARTS;
        if (b >= g.ring) return 0;
        m = 0;
        if (b < a.tilesX) tile = b;                                              // top row
        else if (b < 2 * a.tilesX) tile = (a.tilesY - 1) * a.tilesX + (b - a.tilesX);  // bottom row
        else {
            const int k = b - 2 * a.tilesX;                                       // left / right columns
            tile = (1 + (k >> 1)) * a.tilesX + ((k & 1) ? a.tilesX - 1 : 0);
        }
    } else if (bx < g.ringBlocks + g.objPad) {  // a tile of an object's footprint
        const int o = bx - g.ringBlocks;
        if (o >= g.objBlocks) return 0;
        m = a.firstObj;
        while (m + 1 < a.nmodels && o >= a.objStart[m + 1]) ++m;
        const int i = o - a.objStart[m], rw = a.rect[m][2];
        tile = (a.rect[m][1] + i / rw) * a.tilesX + a.rect[m][0] + i % rw;
    } else {
        m = 0;
        const int j = bx - g.ringBlocks - g.objPad;
        if (j >= g.bgBlocks) {
            tile = j - g.bgBlocks;
            return 2;
        }
        const int s = j >> 3, i = ((s / PARTS) << 3) | (j & 7);  // i: the tile's block index, on XCD j % 8 like its parts
        sub = s % PARTS;
        if (g.ring) {  // interior tiles, XCD-banded like the full image
            const int inX = a.tilesX - 2, inY = a.tilesY - 2, chunkIn = (inX * inY + 7) / 8;
            const int t = (i & 7) * chunkIn + (i >> 3);
            if (t >= inX * inY) return 0;
            tile = (1 + t / inX) * a.tilesX + 1 + t % inX;
        } else {
            tile = (i & 7) * a.chunk + (i >> 3);
        }
    }
    if (tile >= a.tilesX * a.tilesY) return 0;
    // multi-GPU: the replicated background is marched in row bands, one per rank; the rows of the
    // other bands are neither marched nor written here (they arrive by all-gather)
    if (m < a.firstObj && a.bandTiles > 0) {
        const int tyy = tile / a.tilesX;
        if (tyy < a.bandTile0 || tyy >= a.bandTile0 + a.bandTiles) return 0;
    }
    return 1;
}

// zero-fill of the objects' images outside their footprints: kZeroTiles tiles per index i; a workgroup of 256
// threads writes one 16x16 tile per round, `part` of `parts` workgroups sharing the index take every parts-th tile
__device__ __forceinline__ void raycast_zero_fill(const RaycastBatchArgs& a, int i, int part, int parts) {
    const int perObj = (a.tilesX * a.tilesY + kZeroTiles - 1) / kZeroTiles;
    const int mz = a.firstObj + i / perObj;
    if (mz >= a.nmodels) return;
    const emf_model_t& mo = a.models[mz];
    const int tx0 = a.rect[mz][0], ty0 = a.rect[mz][1], tx1 = tx0 + a.rect[mz][2], ty1 = ty0 + a.rect[mz][3];
    const int px = threadIdx.x & 15, py = threadIdx.x >> 4;
    for (int t = (i % perObj) * kZeroTiles + part; t < min((i % perObj + 1) * kZeroTiles, a.tilesX * a.tilesY); t += parts) {
        const int tyz = t / a.tilesX, txz = t - tyz * a.tilesX;
        if (txz >= tx0 && txz < tx1 && tyz >= ty0 && tyz < ty1) continue;  // marched
        const int x = txz * kRbTile + px, y = tyz * kRbTile + py;
        if (x < a.w && y < a.h) {
            const size_t pix = static_cast<size_t>(y) * a.w + x;
            mo.raylengths[pix] = 0.f;
            float* pv = mo.vertices + 3 * pix;
            float* pn = mo.normals + 3 * pix;
            pv[0] = pv[1] = pv[2] = 0.f;
            pn[0] = pn[1] = pn[2] = 0.f;
            mo.hitMask[pix] = 0;
        }
    }
}

__device__ __forceinline__ RayVolume ray_volume_of(const RaycastBatchArgs& a, int m, bool flags) {
    const emf_model_t& md = a.models[m];
    RayVolume v;
    v.tsdf = md.tsdf;
    v.grads = md.grads;
    v.weights = md.weights;
    v.fg = md.fgVolMask;
    v.R = pose_R(a.poses.p[m]);
    v.cam = pose_t(a.poses.p[m]);
    v.n = I3{md.res[0], md.res[1], md.res[2]};
    // reserved bit 1: answer uniform lookups from the flags without gathering
    v.bricks = (flags && md.brickFlags) ? md.brickFlags + brick_count(v.n) : nullptr;
    v.blendFromFlags = (md.reserved & 2) != 0;
    v.voxelSize = md.voxelSize;
    v.truncdist = md.truncdist;
    v.rcpVoxel = ((a.divideMask >> m) & 1u) ? 0.f : md.rcpVoxel;
    return v;
}

// MODE 0: one lane per ray, flag-aware per-lane march (march_ray: brick flags, 64-bit offsets);
// MODE 1: one lane per ray, march_lane (march_wave.hpp), a workgroup = a 16x16-pixel tile, a wave = an 8x8 cell;
// MODE 4: the BACKGROUND with FOUR lanes per ray, march_quad<4>: a workgroup = ONE 8x8 cell (4 waves of 4x4 pixels x 4
//         rows), four workgroups per tile; objects as in MODE 1;
// MODE 2: the background with TWO lanes per ray, march_quad<2>: a workgroup = a 16x8 half tile (4 waves of 8x4 pixels x
//         2 rows), two per tile; objects as in MODE 1.
template <int MODE>
__global__ __launch_bounds__(64 * kRbWaves) void k_raycast_batched(const RaycastBatchArgs a) {
#ifdef EMF_RAY_TRACE
    const unsigned long long trace_t0 = wall_clock64();
#else
    const unsigned long long trace_t0 = 0;
#endif
    int m, tile, sub;
    const int role = raycast_block_role<(MODE >= 2 ? MODE : 1)>(a, blockIdx.x, m, tile, sub);
    if (role == 0) return;
    if (role == 2) {
        raycast_zero_fill(a, tile, 0, 1);
        return;
    }
    const int tyy = tile / a.tilesX, txx = tile - tyy * a.tilesX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const emf_model_t& md = a.models[m];
    const RayVolume v = ray_volume_of(a, m, MODE == 0);
    // incoming raylength is zero by construction (the reference zeroes it first, Q5)
    if constexpr (MODE != 0) {
        int x, y, cellX, cellY;
        const bool rows = MODE >= 2 && m < a.firstObj;  // (block-uniform) the background's rays get MODE lanes each
        if (MODE == 4 && rows) {
            cellX = 2 * txx + (sub & 1);
            cellY = 2 * tyy + (sub >> 1);
            x = cellX * 8 + (wave & 1) * 4 + (lane & 3);
            y = cellY * 8 + (wave >> 1) * 4 + ((lane >> 2) & 3);
        } else if (MODE == 2 && rows) {
            cellX = 2 * txx + (wave & 1);
            cellY = 2 * tyy + sub;
            x = cellX * 8 + (lane & 7);
            y = cellY * 8 + (wave >> 1) * 4 + ((lane >> 3) & 3);
        } else {
            cellX = 2 * txx + (wave & 1);
            cellY = 2 * tyy + (wave >> 1);
            x = cellX * 8 + (lane & 7);
            y = cellY * 8 + (lane >> 3);
        }
        const bool valid = x < a.w && y < a.h;
        const size_t pix = static_cast<size_t>(y) * a.w + x;
        auto sink = [&](float raylength, const V3& vertex, const V3& normal) {
            md.raylengths[pix] = raylength;
            float* pv = md.vertices + 3 * pix;
            float* pn = md.normals + 3 * pix;
            pv[0] = vertex.x;
            pv[1] = vertex.y;
            pv[2] = vertex.z;
            pn[0] = normal.x;
            pn[1] = normal.y;
            pn[2] = normal.z;
            md.hitMask[pix] = 1;
        };
        // far bound of this 8x8-pixel cell: beyond it no sample of any of its rays can complete
        // a hit (k_far_bounds); 0 = none of its rays can hit at all
        float cut = __builtin_inff();
        if (a.farBounds) cut = a.farBounds[(static_cast<size_t>(m) * (2 * a.tilesY) + cellY) * (2 * a.tilesX) + cellX];
        // (a cell with cut == 0 still sets its rays up: one whose first sample lies in the volume's outer shell is
        // exempt from the bound -- ray_setup -- and must be marched like in the reference)
        MarchCount c;
        bool counts = true;  // one lane per ray reports
        if constexpr (MODE >= 2) {
            if (rows) {
                c = march_wave_quad<MODE>(v, valid, x, y, a.fx, a.fy, a.cx, a.cy, 0.f, sink, cut, lane);
                counts = lane < 64 / MODE;
            } else {
                c = march_wave(v, valid, x, y, a.fx, a.fy, a.cx, a.cy, 0.f, sink, cut);
            }
        } else {
            c = march_wave(v, valid, x, y, a.fx, a.fy, a.cx, a.cy, 0.f, sink, cut);
        }
        if (valid && !c.hit && counts) {  // zeros where there is no hit
            md.raylengths[pix] = 0.f;
            float* pv = md.vertices + 3 * pix;
            float* pn = md.normals + 3 * pix;
            pv[0] = pv[1] = pv[2] = 0.f;
            pn[0] = pn[1] = pn[2] = 0.f;
            md.hitMask[pix] = 0;
        }
        add_ray_stats(a.stats, c.samples, (c.hit && counts) ? 1u : 0u, c.gathered, 0u, lane);
#ifdef EMF_MARCH_STAMP
        {   // the stamps are the wave's, but only lanes that entered the march hold them: take the maximum over lanes
            MarchCount cs = c;
            auto wmax64 = [](unsigned long long v) { for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o); v = w > v ? w : v; } return v; };
            auto wmax32 = [](unsigned v) { for (int o = 32; o > 0; o >>= 1) { const unsigned w = static_cast<unsigned>(__shfl_xor(static_cast<int>(v), o)); v = w > v ? w : v; } return v; };
            cs.ckIssue = wmax64(c.ckIssue); cs.ckWait = wmax64(c.ckWait); cs.ckRest = wmax64(c.ckRest); cs.iters = wmax32(c.iters);
            cs.lateIssue = wmax64(c.lateIssue); cs.lateWait = wmax64(c.lateWait); cs.lateRest = wmax64(c.lateRest);
            for (int k = 0; k < 6; ++k) cs.hist[k] = wmax32(c.hist[k]);
            trace_wave(trace_t0, c.samples, m, tile, wave, lane, &cs);
        }
#else
        trace_wave(trace_t0, c.samples, m, tile, wave, lane);
#endif
    } else {
        const int x = txx * kRbTile + (wave & 1) * 8 + (lane & 7);
        const int y = tyy * kRbTile + (wave >> 1) * 8 + (lane >> 3);
        const bool valid = x < a.w && y < a.h;
        RayHit r;
        r.hit = false;
        r.samples = r.gathered = r.skipped = 0;
        r.raylength = 0.f;
        r.vertex = r.normal = v3(0.f, 0.f, 0.f);
        if (valid) r = march_ray(v, x, y, a.fx, a.fy, a.cx, a.cy, 0.f);
        if (valid) {
            const size_t pix = static_cast<size_t>(y) * a.w + x;
            md.raylengths[pix] = r.raylength;  // zeros where there is no hit
            float* pv = md.vertices + 3 * pix;
            float* pn = md.normals + 3 * pix;
            pv[0] = r.vertex.x;
            pv[1] = r.vertex.y;
            pv[2] = r.vertex.z;
            pn[0] = r.normal.x;
            pn[1] = r.normal.y;
            pn[2] = r.normal.z;
            md.hitMask[pix] = r.hit ? 1 : 0;
        }
        add_ray_stats(a.stats, r.samples, r.hit ? 1u : 0u, r.gathered, r.skipped, lane);
        trace_wave(trace_t0, r.samples, m, tile, wave, lane);
    }
}

// ---- ray far bounds ----------------------------------------------------------------------------------
// Most of what the march does on a scene it has already seen is walk on after nothing can happen any
// more: rays that leave the observed surfaces behind keep sampling to the far side of the volume
// (half-voxel steps through unseen space: the 1000-sample rays that set the kernel's duration), rays
// through an object's box that miss the object cross it for nothing.  The reference's loop
// (TSDF.cu:523-572) writes an output only at a sample with a NEGATIVE blend that follows a sample with
// a POSITIVE one -- so a hit needs a negative voxel among the 8 corners of that sample's cell and a
// positive voxel among the corners of the previous sampled cell, at most one ray step (<= truncdist)
// plus the stale-sample cases discussed in march_wave.hpp away.  The integration kernels keep, per
// 32 x 8 x 8 voxel tile, two sticky bytes "holds a positive / a negative tsdf" (sign maps).  Here every
// tile that has a negative tile next to it and a positive tile within reach is projected into the image
// and raises the FAR BOUND of the 8x8-pixel cells it may cover to the largest distance from the camera
// any of its points has.  A ray's march is then cut at its cell's bound (0: never started): every
// sample it still takes is taken exactly as before -- the step sequence is replayed from the volume's
// entry -- and every sample it no longer takes could not have produced an output.  Bit-identical
// results (tests: per-volume path = no bounds vs batched path, oracle comparisons), fewer samples.
struct FarBoundArgs {
    const emf_model_t* models;
    PoseTable poses;  // camera -> volume
    int nmodels;
    int tileStart[EMF_MAX_BATCH + 1];  // prefix sum of integration tiles per model
    int w, h, cellsX, cellsY;
    float fx, fy, cx, cy;
    float* bounds;  // [model][cellsY][cellsX]
};

__global__ __launch_bounds__(256) void k_far_init(const FarBoundArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x, cells = a.cellsX * a.cellsY;
    if (i >= a.nmodels * cells) return;
    // a model without sign maps -- or with maps that nobody looks at: no list and not in the scan mask --
    // is marched to the end as before
    const int m = i / cells;
    const emf_model_t& md = a.models[m];
    const bool bounded = md.signMaps && (md.relevantTiles || a.tileStart[m + 1] != a.tileStart[m]);
    a.bounds[i] = bounded ? 0.f : __builtin_inff();
}

// Is tile (tx, ty, tz) of model md one in which a hit can be completed?  (see above)
__device__ __forceinline__ bool tile_relevant(const emf_model_t& md, int tx, int ty, int tz) {
    const int ntx = (md.res[0] + kTileX - 1) / kTileX, nty = (md.res[1] + kTileY - 1) / kTileY,
              ntz = (md.res[2] + kTileZ - 1) / kTileZ;
    const uint8_t* pos = md.signMaps;
    const uint8_t* neg = pos + static_cast<size_t>(ntx) * nty * ntz;
    auto any_in = [&](const uint8_t* map, int rx, int ry, int rz) {
        unsigned any = 0;  // OR of plain loads: no short-circuit, so the loads are independent and pipeline
        for (int z = max(tz - rz, 0); z <= min(tz + rz, ntz - 1); ++z)
            for (int y = max(ty - ry, 0); y <= min(ty + ry, nty - 1); ++y)
                for (int x = max(tx - rx, 0); x <= min(tx + rx, ntx - 1); ++x)
                    any |= map[(static_cast<size_t>(z) * nty + y) * ntx + x];
        return any != 0;
    };
    // the negative corner is in the sample's own cell (base voxel in this tile, corners up to +1): this
    // tile or a direct neighbour
    if (!any_in(neg, 1, 1, 1)) return false;
    // the positive corner belongs to the previous sample: one step back (<= truncdist), two where the
    // hit test re-used an older sample (march_wave.hpp), plus the cells' own extent
    const int reach = 2 * static_cast<int>(ceilf(md.truncdist / md.voxelSize)) + 4;
    return any_in(pos, (reach + kTileX - 1) / kTileX, (reach + kTileY - 1) / kTileY, (reach + kTileZ - 1) / kTileZ);
}

// The 64 lanes of a wave project tile (tx, ty, tz) and raise the far bounds of the cells it may cover:
// lanes 0..7 (and their copies) take the corners, then all lanes share the cells (a tile covers some tens
// of cells: one lane walking them alone pays an L2 round trip per cell).  Wave-uniform arguments.
__device__ __forceinline__ void raise_far_bounds_wave(const FarBoundArgs& a, int m, const emf_model_t& md, int tx, int ty,
                                                      int tz) {
    const I3 n = I3{md.res[0], md.res[1], md.res[2]};
    const M33 R = pose_R(a.poses.p[m]);
    const V3 cam = pose_t(a.poses.p[m]);
    const V3 half = half_extent(n);
    const int lane = threadIdx.x & 63;
    // corner (lane & 7) of the tile's voxel box widened by 1.5 voxels (a cell's far corners, rounding of the
    // march's position arithmetic), in the camera frame
    const int k = lane & 7;
    const float ix = (k & 1) ? static_cast<float>(min((tx + 1) * kTileX, n.x)) + 1.5f : static_cast<float>(tx * kTileX) - 1.5f;
    const float iy = (k & 2) ? static_cast<float>(min((ty + 1) * kTileY, n.y)) + 1.5f : static_cast<float>(ty * kTileY) - 1.5f;
    const float iz = (k & 4) ? static_cast<float>(min((tz + 1) * kTileZ, n.z)) + 1.5f : static_cast<float>(tz * kTileZ) - 1.5f;
    const V3 q = v3((ix - half.x) * md.voxelSize, (iy - half.y) * md.voxelSize, (iz - half.z) * md.voxelSize);
    const V3 d = v3(q.x - cam.x, q.y - cam.y, q.z - cam.z);
    const V3 c = v3(R.r0.x * d.x + R.r1.x * d.y + R.r2.x * d.z, R.r0.y * d.x + R.r1.y * d.y + R.r2.y * d.z,
                    R.r0.z * d.x + R.r1.z * d.y + R.r2.z * d.z);  // R^T d
    const bool behind = !(c.z > 1e-2f * md.voxelSize);  // at or behind the camera plane: cover the whole image
    float far = norm(d);
    float umin = behind ? 3e38f : a.fx * c.x / c.z + a.cx, umax = behind ? -3e38f : umin;
    float vmin = behind ? 3e38f : a.fy * c.y / c.z + a.cy, vmax = behind ? -3e38f : vmin;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {  // over the 8 corners (every group of 8 lanes holds all of them)
        far = fmaxf(far, __shfl_xor(far, o));
        umin = fminf(umin, __shfl_xor(umin, o));
        umax = fmaxf(umax, __shfl_xor(umax, o));
        vmin = fminf(vmin, __shfl_xor(vmin, o));
        vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    const bool wide = __ballot(behind) != 0ull;
    int cx0 = 0, cx1 = a.cellsX - 1, cy0 = 0, cy1 = a.cellsY - 1;
    if (!wide) {
        if (!(umax >= -2.f && vmax >= -2.f && umin <= static_cast<float>(a.w) + 1.f && vmin <= static_cast<float>(a.h) + 1.f))
            return;  // projects beside the image (wave-uniform)
        cx0 = max(static_cast<int>(floorf((umin - 2.f) / 8.f)), 0);
        cy0 = max(static_cast<int>(floorf((vmin - 2.f) / 8.f)), 0);
        cx1 = min(static_cast<int>(floorf((fminf(umax, 1e6f) + 2.f) / 8.f)), a.cellsX - 1);
        cy1 = min(static_cast<int>(floorf((fminf(vmax, 1e6f) + 2.f) / 8.f)), a.cellsY - 1);
    }
    // raylength of a sample = its distance from the camera (unit direction): a relative and an absolute margin
    const unsigned bits = __float_as_uint(far * 1.0001f + 2.f * md.voxelSize);
    unsigned* cells = reinterpret_cast<unsigned*>(a.bounds) + static_cast<size_t>(m) * a.cellsX * a.cellsY;
    const int wdt = cx1 - cx0 + 1, cnt = wdt * (cy1 - cy0 + 1);
    for (int i = lane; i < cnt; i += 64) {
        unsigned* cp = &cells[(cy0 + i / wdt) * a.cellsX + cx0 + i % wdt];
        // positive floats order as integers; most tiles raise nothing (plain read first: an atomic is a
        // trip to L2 and back whether or not it changes the value)
        if (__builtin_nontemporal_load(cp) < bits) atomicMax(cp, bits);
    }
}

// The relevance test of tile_relevant with the neighbourhood spread over the lanes of a wave.
__device__ __forceinline__ bool tile_relevant_wave(const emf_model_t& md, int tx, int ty, int tz) {
    const int ntx = (md.res[0] + kTileX - 1) / kTileX, nty = (md.res[1] + kTileY - 1) / kTileY,
              ntz = (md.res[2] + kTileZ - 1) / kTileZ;
    const uint8_t* pos = md.signMaps;
    const uint8_t* neg = pos + static_cast<size_t>(ntx) * nty * ntz;
    const int lane = threadIdx.x & 63;
    auto any_in = [&](const uint8_t* map, int rx, int ry, int rz) {
        const int wx = 2 * rx + 1, wy = 2 * ry + 1, cnt = wx * wy * (2 * rz + 1);
        bool any = false;
        for (int i = lane; i < cnt; i += 64) {
            const int x = tx - rx + i % wx, y = ty - ry + (i / wx) % wy, z = tz - rz + i / (wx * wy);
            if (x >= 0 && x < ntx && y >= 0 && y < nty && z >= 0 && z < ntz)
                any = any || map[(static_cast<size_t>(z) * nty + y) * ntx + x] != 0;
        }
        return __ballot(any) != 0ull;
    };
    if (!any_in(neg, 1, 1, 1)) return false;
    const int reach = 2 * static_cast<int>(ceilf(md.truncdist / md.voxelSize)) + 4;
    return any_in(pos, (reach + kTileX - 1) / kTileX, (reach + kTileY - 1) / kTileY, (reach + kTileZ - 1) / kTileZ);
}

// models WITHOUT a relevant-tile list (object volumes): one wave per tile of their sign maps
__global__ __launch_bounds__(256) void k_far_bounds(const FarBoundArgs a) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.tileStart[a.nmodels]) return;  // wave-uniform
    int m = 0;
    while (m + 1 < a.nmodels && i >= a.tileStart[m + 1]) ++m;  // (models outside the scan mask have no tiles here)
    const emf_model_t& md = a.models[m];
    if (!md.signMaps) return;
    const int ntx = (md.res[0] + kTileX - 1) / kTileX, nty = (md.res[1] + kTileY - 1) / kTileY;
    const int t = i - a.tileStart[m];
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    if (tile_relevant_wave(md, tx, ty, tz)) raise_far_bounds_wave(a, m, md, tx, ty, tz);
}

// models WITH a list (emf_hip_updateRelevantTiles, after each integration): one WAVE per listed tile -- a few
// thousand tiles instead of a neighbourhood scan around every tile of a 512^3 volume
constexpr int kFarListBlocks = 512;  // x 4 waves per model; the list is walked with that stride (64: 36 us for 3300 tiles)
__global__ __launch_bounds__(256) void k_far_bounds_listed(const FarBoundArgs a) {
    const int m = blockIdx.x / kFarListBlocks;
    const emf_model_t& md = a.models[m];
    if (!md.signMaps || !md.relevantTiles || a.tileStart[m + 1] != a.tileStart[m]) return;  // no maps / scanned instead
    const unsigned count = md.relevantTiles[0];
    const int ntx = (md.res[0] + kTileX - 1) / kTileX, nty = (md.res[1] + kTileY - 1) / kTileY;
    for (unsigned e = (blockIdx.x % kFarListBlocks) * 4u + (threadIdx.x >> 6); e < count; e += kFarListBlocks * 4u) {
        const int t = static_cast<int>(md.relevantTiles[1 + e]);
        raise_far_bounds_wave(a, m, md, t % ntx, (t / ntx) % nty, t / (ntx * nty));
    }
}

// (re)build the relevant-tile lists from the sign maps; the counts must be zero (emf_hip_updateRelevantTiles)
struct RelevantArgs {
    const emf_model_t* models;
    int nmodels;
    int tileStart[EMF_MAX_BATCH + 1];
};
__global__ __launch_bounds__(256) void k_relevant_tiles(const RelevantArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    int m = 0, t = 0;
    if (i < a.tileStart[a.nmodels]) {
        while (m + 1 < a.nmodels && i >= a.tileStart[m + 1]) ++m;
        const emf_model_t& md = a.models[m];
        if (md.signMaps && md.relevantTiles) {
            const int ntx = (md.res[0] + kTileX - 1) / kTileX, nty = (md.res[1] + kTileY - 1) / kTileY;
            t = i - a.tileStart[m];
            keep = tile_relevant(md, t % ntx, (t / ntx) % nty, t / (ntx * nty));
        }
    }
    // a wave may straddle two models: append per lane group with one atomic per (wave, model)
    while (true) {
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) break;
        const int lead = __ffsll(static_cast<long long>(mask)) - 1;
        const int mLead = __shfl(m, lead);
        const bool mineNow = keep && m == mLead;
        const unsigned long long grp = __ballot(mineNow);
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == lead) base = atomicAdd(&a.models[mLead].relevantTiles[0], static_cast<unsigned>(__popcll(grp)));
        base = __shfl(base, lead);
        if (mineNow) {
            a.models[m].relevantTiles[1 + base + __popcll(grp & ((1ull << lane) - 1ull))] = static_cast<unsigned>(t);
            keep = false;
        }
    }
}
__global__ void k_relevant_reset(const emf_model_t* models, int nmodels) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < nmodels && models[m].relevantTiles) models[m].relevantTiles[0] = 0;
}

// exact rebuild of a volume's sign maps from its values (after anything but the tile integration wrote them)
__global__ __launch_bounds__(256) void k_sign_maps(const float* __restrict__ tsdf, I3 n, uint8_t* __restrict__ maps) {
    const int ntx = (n.x + kTileX - 1) / kTileX, nty = (n.y + kTileY - 1) / kTileY, ntz = (n.z + kTileZ - 1) / kTileZ;
    const int t = blockIdx.x;
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int xg = threadIdx.x & 7, yy = (threadIdx.x >> 3) & 7, zs = threadIdx.x >> 6;
    bool p = false, ng = false;
    const int y = ty * kTileY + yy;
    for (int i = 0; i < 2; ++i) {
        const int z = tz * kTileZ + zs + 4 * i;
        for (int e = 0; e < 4; ++e) {
            const int x = tx * kTileX + 4 * xg + e;
            if (x < n.x && y < n.y && z < n.z) {
                const float v = tsdf[(static_cast<size_t>(z) * n.y + y) * n.x + x];
                p = p || v > 0.f;
                ng = ng || v < 0.f;
            }
        }
    }
    const int anyP = __syncthreads_or(p), anyN = __syncthreads_or(ng);
    if (threadIdx.x == 0) {
        maps[t] = anyP ? 1 : 0;
        maps[static_cast<size_t>(ntx) * nty * ntz + t] = anyN ? 1 : 0;
    }
}

// unseen-tile map (emf_model_t.unseenTiles) from the values: every weight 0 and every tsdf finite
__global__ __launch_bounds__(256) void k_unseen_tiles(const float* __restrict__ tsdf, const float* __restrict__ weights,
                                                      I3 n, uint8_t* __restrict__ map) {
    const int ntx = (n.x + kTileX - 1) / kTileX, nty = (n.y + kTileY - 1) / kTileY;
    const int t = blockIdx.x;
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int xg = threadIdx.x & 7, yy = (threadIdx.x >> 3) & 7, zs = threadIdx.x >> 6;
    bool seen = false;
    const int y = ty * kTileY + yy;
    for (int i = 0; i < 2; ++i) {
        const int z = tz * kTileZ + zs + 4 * i;
        for (int e = 0; e < 4; ++e) {
            const int x = tx * kTileX + 4 * xg + e;
            if (x < n.x && y < n.y && z < n.z) {
                const size_t v = (static_cast<size_t>(z) * n.y + y) * n.x + x;
                seen = seen || !(weights[v] == 0.f) || !(fabsf(tsdf[v]) <= 3.0e38f);
            }
        }
    }
    const int any = __syncthreads_or(seen);
    if (threadIdx.x == 0) map[t] = any ? 0 : 1;
}

// ---- batched integration ---------------------------------------------------------------------------

__device__ __forceinline__ size_t tile_count(const I3& n) {
    return static_cast<size_t>((n.x + kTileX - 1) / kTileX) * ((n.y + kTileY - 1) / kTileY) * ((n.z + kTileZ - 1) / kTileZ);
}
__device__ __forceinline__ size_t tile_index(const I3& n, int x0, int y0, int z0) {
    const int ntx = (n.x + kTileX - 1) / kTileX, nty = (n.y + kTileY - 1) / kTileY;
    return (static_cast<size_t>(z0 / kTileZ) * nty + y0 / kTileY) * ntx + x0 / kTileX;
}

struct IntegrateBatchArgs {
    const emf_model_t* models;
    PoseTable poses;  // volume -> camera
    int nmodels;
    int tileStart[EMF_MAX_BATCH + 1];  // prefix sum of tiles per model
    const int32_t* visible;
    unsigned long long* stats;
    Img<const float> depth, invLambda;
    int w, h;
    M33 K;
    bool pinhole;
};

// 6 waves per SIMD (<= 80 VGPRs): measured best; 5 and below hide less, 8 spills
// (Round 5, measured and dropped: 2 / 4 / 8 consecutive tiles per workgroup, on the suspicion that 4096 workgroups of a
// microsecond or two are paced by the dispatcher -- 50 -> 63 / 70 / 103 us for the four 128^3 objects of configs[1]: the
// launch is bound by the latency of a tile's dependent loads, and fewer, longer workgroups hide less of it.)
__attribute__((amdgpu_waves_per_eu(EMF_INT_WPE, EMF_INT_WPE)))
__global__ __launch_bounds__(256) void k_integrate_batched(const IntegrateBatchArgs a) {
    __shared__ unsigned lds[32];
    int m = 0;
    while (m + 1 < a.nmodels && static_cast<int>(blockIdx.x) >= a.tileStart[m + 1]) ++m;
    if (a.visible && a.visible[m] == 0) return;  // EMFusion.cpp:869-872, decided on the device
    const emf_model_t& md = a.models[m];
    IntegrateGeom g;
    g.depth = a.depth;
    g.invLambda = a.invLambda;
    g.assoc = Img<const float>{md.assoc, static_cast<size_t>(a.w) * sizeof(float)};
    g.w = a.w;
    g.h = a.h;
    g.R = pose_R(a.poses.p[m]);
    g.t = pose_t(a.poses.p[m]);
    g.K = a.K;
    g.pinhole = a.pinhole;
    g.n = I3{md.res[0], md.res[1], md.res[2]};
    g.voxelSize = md.voxelSize;
    g.truncdist = md.truncdist;
    g.maxWeight = md.maxWeight;
    const int b = blockIdx.x - a.tileStart[m];
    if (a.stats && b == 0 && threadIdx.x == 0)
        atomicAdd(a.stats, static_cast<unsigned long long>(g.n.x) * g.n.y * g.n.z);
    const int ntx = (g.n.x + kTileX - 1) / kTileX, nty = (g.n.y + kTileY - 1) / kTileY;
    const int tx = b % ntx, ty = (b / ntx) % nty, tz = b / (ntx * nty);
    const size_t tile = tile_index(g.n, tx * kTileX, ty * kTileY, tz * kTileZ);
    uint8_t* sp = md.signMaps ? md.signMaps + tile : nullptr;
    integrate_tile(g, md.tsdf, md.weights, md.brickFlags, tx * kTileX, ty * kTileY, tz * kTileZ,
                   lds, nullptr, nullptr, 0, nullptr, nullptr, false, sp, sp ? sp + tile_count(g.n) : nullptr,
                   md.unseenTiles ? md.unseenTiles + tile : nullptr);
}

// Models whose Nx is not a multiple of 4 (object volumes after ObjTSDF::resize, which only keeps the
// resolution even) cannot use the float4 tiles: their voxels go one per lane, in 256-voxel chunks of
// the linear index, with the same per-voxel code and the same device-side visibility gate.
__global__ __launch_bounds__(256) void k_integrate_batched_linear(const IntegrateBatchArgs a) {
    int m = 0;
    while (m + 1 < a.nmodels && static_cast<int>(blockIdx.x) >= a.tileStart[m + 1]) ++m;
    if (a.visible && a.visible[m] == 0) return;
    const emf_model_t& md = a.models[m];
    IntegrateGeom g;
    g.depth = a.depth;
    g.invLambda = a.invLambda;
    g.assoc = Img<const float>{md.assoc, static_cast<size_t>(a.w) * sizeof(float)};
    g.w = a.w;
    g.h = a.h;
    g.R = pose_R(a.poses.p[m]);
    g.t = pose_t(a.poses.p[m]);
    g.K = a.K;
    g.pinhole = a.pinhole;
    g.n = I3{md.res[0], md.res[1], md.res[2]};
    g.voxelSize = md.voxelSize;
    g.truncdist = md.truncdist;
    g.maxWeight = md.maxWeight;
    const size_t total = static_cast<size_t>(g.n.x) * g.n.y * g.n.z;
    const int b = blockIdx.x - a.tileStart[m];
    if (a.stats && b == 0 && threadIdx.x == 0) atomicAdd(a.stats, static_cast<unsigned long long>(total));
    const size_t i = static_cast<size_t>(b) * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / g.n.x;
    const int x = static_cast<int>(i - row * g.n.x);
    const int y = static_cast<int>(row % g.n.y), z = static_cast<int>(row / g.n.y);
    float samp = 0.f, aw = 0.f;
    const int kind = classify_voxel(g, half_extent(g.n), x, y, z, samp, aw);
    if (kind == kSkip) return;
    float wv = md.weights[i];
    float tv = md.tsdf[i];  // also for the "if unseen" branches: they store only a value that differs
    const int changed = apply_voxel(kind, samp, aw, g.maxWeight, tv, wv);
    if (changed & 1) md.tsdf[i] = tv;
    if (changed & 2) md.weights[i] = wv;
    if (md.brickFlags && (changed & 1))  // conservative, as in k_update_tsdf_linear
        md.brickFlags[(static_cast<size_t>(z >> kBrickShift) * bricks_along(g.n.y) + (y >> kBrickShift)) *
                          bricks_along(g.n.x) + (x >> kBrickShift)] = kBrickMixed;
}

// ---- two-level launch: cull boxes of 1x2x2 tiles first, integrate only the tiles of the survivors --------
// Most tiles of a large volume lie outside the view cone and return after the cull test -- but
// every one of them still costs a workgroup dispatch, and the dispatcher, not the CUs, then sets the
// pace of the first part of the grid.  k_integrate_cull applies the same test to boxes
// of 1x2x2 tiles, 32 x 16 x 16 voxels (one lane per box; a box whose 8 corners project beyond the same image border, in
// front of the camera, holds no voxel that lands in the image: every tile in it would be culled) and
// appends the others to a list; k_integrate_listed then runs 8 workgroups per list entry.  Its grid
// is sized by the caller from the survivor count of an EARLIER frame (the count comes back
// asynchronously); a grid that turns out too small strides over the rest, one too large exits.
#ifndef EMF_INT_BOX_X
#define EMF_INT_BOX_X 1  // in tiles: 32 x 16 x 16 voxels measured best (2,2,2: +2 %, 2,4,4: +4 %)
#define EMF_INT_BOX_Y 2
#define EMF_INT_BOX_Z 2
#endif
constexpr int kBoxX = EMF_INT_BOX_X * kTileX, kBoxY = EMF_INT_BOX_Y * kTileY, kBoxZ = EMF_INT_BOX_Z * kTileZ;
constexpr unsigned kBoxTiles = EMF_INT_BOX_X * EMF_INT_BOX_Y * EMF_INT_BOX_Z;

// Out-of-place integration (emf_hip_integrateBatchedCulledOut): per model the second copy of the
// volume that receives the result, and one byte per 32 x 8 x 8 tile saying whether the two copies
// differ there (`dirtyPrev`: left by the previous call, `dirtyNext`: written by this one).
struct IntegrateOutTable {
    float* tsdf[EMF_MAX_BATCH];
    float* weights[EMF_MAX_BATCH];
    const uint8_t* dirtyPrev[EMF_MAX_BATCH];
    uint8_t* dirtyNext[EMF_MAX_BATCH];
};

struct IntegrateCullArgs {
    IntegrateBatchArgs b;
    int boxStart[EMF_MAX_BATCH + 1];  // prefix sum of boxes per model (0 boxes for untiled models)
    unsigned* list;                    // entries: model << 24 | box index within the model
    unsigned* count;                   // survivors appended so far; count[1]: float bits of the frame's largest
                                       // depth (both zeroed by the caller's memset, made by k_integrate_cull)
    IntegrateOutTable out;             // second copies (haveOut != 0), by value like the poses
    int haveOut;
    int deepTiles;                     // let integrate_tile use count[1] (EMF_DEEP_TILES=0: not)
};


__device__ __forceinline__ IntegrateGeom geom_of(const IntegrateBatchArgs& a, int m) {
    const emf_model_t& md = a.models[m];
    IntegrateGeom g;
    g.depth = a.depth;
    g.invLambda = a.invLambda;
    g.assoc = Img<const float>{md.assoc, static_cast<size_t>(a.w) * sizeof(float)};
    g.w = a.w;
    g.h = a.h;
    g.R = pose_R(a.poses.p[m]);
    g.t = pose_t(a.poses.p[m]);
    g.K = a.K;
    g.pinhole = a.pinhole;
    g.n = I3{md.res[0], md.res[1], md.res[2]};
    g.voxelSize = md.voxelSize;
    g.truncdist = md.truncdist;
    g.maxWeight = md.maxWeight;
    return g;
}

__global__ __launch_bounds__(256) void k_integrate_cull(const IntegrateCullArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    {   // the frame's largest depth, for the deep tiles of the listed launch (integrate_tile): non-negative
        // floats order like their bits; NaN and values <= 0 never win
        float mx = 0.f;
        const int npix = a.b.w * a.b.h, stride = gridDim.x * 256;
        for (int p = i; p < npix; p += stride) {
            const float d = a.b.depth.row(p / a.b.w)[p % a.b.w];
            mx = d > mx ? d : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(a.count + 1, __float_as_uint(mx));
    }
    bool keep = false;
    unsigned entry = 0;
    if (i < a.boxStart[a.b.nmodels]) {
        int m = 0;
        while (m + 1 < a.b.nmodels && i >= a.boxStart[m + 1]) ++m;
        const bool open = !a.b.visible || a.b.visible[m] != 0;  // the gate of EMFusion.cpp:869-872
        if (open || a.haveOut) {
            const IntegrateGeom g = geom_of(a.b, m);
            const int box = i - a.boxStart[m];
            if (open && a.b.stats && box == 0)
                atomicAdd(a.b.stats, static_cast<unsigned long long>(g.n.x) * g.n.y * g.n.z);
            const int nbx = (g.n.x + kBoxX - 1) / kBoxX, nby = (g.n.y + kBoxY - 1) / kBoxY;
            const int bx = box % nbx, by = (box / nbx) % nby, bz = box / (nbx * nby);
            const int x0 = bx * kBoxX, y0 = by * kBoxY, z0 = bz * kBoxZ;
            const int x1 = min(x0 + kBoxX, g.n.x) - 1, y1 = min(y0 + kBoxY, g.n.y) - 1,
                      z1 = min(z0 + kBoxZ, g.n.z) - 1;
            const V3 half = half_extent(g.n);
            bool front = true, left = true, right = true, up = true, down = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // tile_culled's test, the 8 corners in one lane
                const V3 p = voxel_in_camera(g, half, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0, (k & 4) ? z1 : z0);
                const V3 q = mul(g.K, p);
                const float u = q.x / q.z, v = q.y / q.z;
                front = front && p.z > 1e-3f;
                left = left && u < -1.5f;
                right = right && u > static_cast<float>(g.w) + 0.5f;
                up = up && v < -1.5f;
                down = down && v > static_cast<float>(g.h) + 0.5f;
            }
            keep = open && !(front && (left || right || up || down));
            if (!keep && a.haveOut) {
                // out of place: a box outside the view cone still gets its workgroups if the previous
                // integration changed one of its tiles -- the other copy has to catch up there
                const uint8_t* dirty = a.out.dirtyPrev[m];
                for (unsigned sub = 0; sub < kBoxTiles; ++sub) {
                    const int tx0 = x0 + static_cast<int>(sub % EMF_INT_BOX_X) * kTileX,
                              ty0 = y0 + static_cast<int>((sub / EMF_INT_BOX_X) % EMF_INT_BOX_Y) * kTileY,
                              tz0 = z0 + static_cast<int>(sub / (EMF_INT_BOX_X * EMF_INT_BOX_Y)) * kTileZ;
                    if (tx0 < g.n.x && ty0 < g.n.y && tz0 < g.n.z) {
                        const size_t t = tile_index(g.n, tx0, ty0, tz0);
                        keep = keep || (dirty[t] | dirty[tile_count(g.n) + t]) != 0;
                    }
                }
            }
            entry = (static_cast<unsigned>(m) << 24) | static_cast<unsigned>(box);
        }
    }
    // one atomic per wave
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == __ffsll(static_cast<long long>(mask)) - 1) base = atomicAdd(a.count, static_cast<unsigned>(__popcll(mask)));
    base = __shfl(base, __ffsll(static_cast<long long>(mask)) - 1);
    if (keep) a.list[base + __popcll(mask & ((1ull << lane) - 1ull))] = entry;
}

// tile `sub` (0..7) of list entry e
template <bool OUT>
__device__ __forceinline__ void integrate_listed_tile(const IntegrateCullArgs& a, unsigned e, unsigned sub,
                                                      unsigned* lds) {
    const unsigned entry = a.list[e];
    const int m = static_cast<int>(entry >> 24), box = static_cast<int>(entry & 0xffffffu);
    IntegrateGeom g = geom_of(a.b, m);
    g.maxDepthBits = a.deepTiles ? a.count + 1 : nullptr;  // written by k_integrate_cull, the launch before this one
    const emf_model_t& md = a.b.models[m];
    const int nbx = (g.n.x + kBoxX - 1) / kBoxX, nby = (g.n.y + kBoxY - 1) / kBoxY;
    const int bx = box % nbx, by = (box / nbx) % nby, bz = box / (nbx * nby);
    const int dx = static_cast<int>(sub % EMF_INT_BOX_X), dy = static_cast<int>((sub / EMF_INT_BOX_X) % EMF_INT_BOX_Y),
              dz = static_cast<int>(sub / (EMF_INT_BOX_X * EMF_INT_BOX_Y));
    const int x0 = (EMF_INT_BOX_X * bx + dx) * kTileX, y0 = (EMF_INT_BOX_Y * by + dy) * kTileY,
              z0 = (EMF_INT_BOX_Z * bz + dz) * kTileZ;
    if (x0 < g.n.x && y0 < g.n.y && z0 < g.n.z) {  // block-uniform
        const size_t t = tile_index(g.n, x0, y0, z0), nt = tile_count(g.n);
        uint8_t* sp = md.signMaps ? md.signMaps + t : nullptr;
        if constexpr (OUT) {
            const uint8_t* dp = a.out.dirtyPrev[m];
            const int force = (dp[t] ? 1 : 0) | (dp[nt + t] ? 2 : 0);
            integrate_tile<true>(g, md.tsdf, md.weights, nullptr, x0, y0, z0, lds, a.out.tsdf[m],
                                 a.out.weights[m], force, a.out.dirtyNext[m] + t, a.out.dirtyNext[m] + nt + t,
                                 a.b.visible && a.b.visible[m] == 0, sp, sp ? sp + nt : nullptr,
                                 md.unseenTiles ? md.unseenTiles + t : nullptr);
        } else {
            integrate_tile(g, md.tsdf, md.weights, nullptr, x0, y0, z0, lds, nullptr, nullptr, 0, nullptr, nullptr,
                           false, sp, sp ? sp + nt : nullptr, md.unseenTiles ? md.unseenTiles + t : nullptr);
        }
    }
}

// one workgroup per (entry, tile): entries [0, min(count, grid / 8))
template <bool OUT>
__attribute__((amdgpu_waves_per_eu(EMF_INT_WPE, EMF_INT_WPE)))
__global__ __launch_bounds__(256) void k_integrate_listed(const IntegrateCullArgs a) {
    __shared__ unsigned lds[32];
    const unsigned e = blockIdx.x / kBoxTiles;
    if (e >= *a.count) return;
    integrate_listed_tile<OUT>(a, e, blockIdx.x % kBoxTiles, lds);
}

// the entries a too-small grid left over: a few workgroups stride over [first, count)
template <bool OUT>
__global__ __launch_bounds__(256) void k_integrate_listed_rest(const IntegrateCullArgs a, unsigned first) {
    __shared__ unsigned lds[32];
    const unsigned todo = *a.count;
    for (unsigned e = first + blockIdx.x / kBoxTiles; e < todo; e += gridDim.x / kBoxTiles) {
        integrate_listed_tile<OUT>(a, e, blockIdx.x % kBoxTiles, lds);
        __syncthreads();
    }
}

// refresh the dilated flags of every model that has a flag buffer (one thread per brick)
struct DilateBatchArgs {
    const emf_model_t* models;
    int nmodels;
    int brickStart[EMF_MAX_BATCH + 1];
    const int32_t* visible;
};

__global__ __launch_bounds__(256) void k_dilate_batched(const DilateBatchArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.brickStart[a.nmodels]) return;
    int m = 0;
    while (m + 1 < a.nmodels && i >= a.brickStart[m + 1]) ++m;
    if (a.visible && a.visible[m] == 0) return;  // not integrated this frame: flags unchanged
    const emf_model_t& md = a.models[m];
    if (!md.brickFlags) return;
    const int nbx = bricks_along(md.res[0]), nby = bricks_along(md.res[1]),
              nbz = bricks_along(md.res[2]);
    const int j = i - a.brickStart[m];
    const int bx = j % nbx, by = (j / nbx) % nby, bz = j / (nbx * nby);
    md.brickFlags[static_cast<size_t>(nbx) * nby * nbz + j] =
        dilated_flag(md.brickFlags, nbx, nby, nbz, bx, by, bz);
}

__global__ void k_vis_flags(const int32_t* __restrict__ counts, int nmodels, int thresh,
                            int32_t* __restrict__ visible, int32_t* __restrict__ mirror) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nmodels) return;
    visible[s] = s == 0 ? 1 : (counts[s - 1] > thresh ? 1 : 0);
    if (mirror && s > 0) mirror[s - 1] = counts[s - 1];  // host-visible copy: no copy kernel, no copy engine
}

int check_batch(const emf_model_t* models, const emf_pose_t* poses, int nmodels, const char* fn) {
    if (!models) return fail(EMF_E_NULL, "%s: models_dev is NULL", fn);
    if (!poses) return fail(EMF_E_NULL, "%s: poses are NULL", fn);
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH)
        return fail(EMF_E_LIMIT, "%s: nmodels = %d, expected 1..%d", fn, nmodels, EMF_MAX_BATCH);
    return EMF_OK;
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

namespace {
int estep_launch(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                 const emf_image_t* depth, const float* K, const emf_image_t* points, int normalize,
                 const emf_image_t* norm, const emf_image_t* objSum, emf_stream_t stream, const char* fn,
                 const emf_peer_t* group = nullptr, uint32_t seq = 0) {
    EMF_TRY(check_batch(models_dev, poseCO_host, nmodels, fn));
    EMF_TRY(check_image(points, 12, "estepBatched: points"));
    EstepArgs a;
    a.models = models_dev;
    for (int m = 0; m < nmodels; ++m) a.poses.p[m] = poseCO_host[m];
    a.nmodels = nmodels;
    a.points = img<const float>(points);
    a.w = points->width;
    a.h = points->height;
    a.normalize = normalize ? 1 : 0;
    a.norm = Img<float>{nullptr, 0};
    a.objSum = Img<float>{nullptr, 0};
    a.depth = Img<const float>{nullptr, 0};
    a.pointsOut = Img<float>{nullptr, 0};
    a.fx = a.fy = 1.f;
    a.cx = a.cy = 0.f;
    if (depth) {
        EMF_TRY(check_image(depth, 4, "estepBatchedFromDepth: depth"));
        EMF_TRY(check_same_size(depth, points, "depth", "points"));
        if (!K) return fail(EMF_E_NULL, "estepBatchedFromDepth: K is NULL");
        a.depth = img<const float>(depth);
        a.pointsOut = img<float>(points);
        a.fx = K[0];
        a.fy = K[4];
        a.cx = K[2];
        a.cy = K[5];
    }
    if (norm) {
        EMF_TRY(check_image(norm, 4, "estepBatched: norm"));
        EMF_TRY(check_same_size(norm, points, "norm", "points"));
        a.norm = img<float>(norm);
    }
    a.peerWorld = 0;
    a.peerFences = 0;
    a.peerOff = 0;
    for (int p = 0; p < EMF_MAX_PEERS; ++p) a.peerSlots[p] = nullptr;
    if (group) {
        PeerArgs pa;
        EMF_TRY(peer_args(group, pa, fn));
        if (static_cast<size_t>(a.w) * a.h * sizeof(float) > pa.slotBytes)
            return fail(EMF_E_ARG, "%s: %d x %d floats exceed the %zu-byte slot", fn, a.w, a.h, pa.slotBytes);
        a.peerWorld = pa.world;
        a.peerFences = pa.fences;
        a.peerOff = (static_cast<size_t>(seq & 1u) * pa.world + pa.rank) * pa.slotBytes;
        for (int p = 0; p < pa.world; ++p) a.peerSlots[p] = pa.slots[p];
    } else if (!normalize && objSum) {
        EMF_TRY(check_image(objSum, 4, "estepBatched: objSum"));
        EMF_TRY(check_same_size(objSum, points, "objSum", "points"));
        a.objSum = img<float>(objSum);
    }
    hipLaunchKernelGGL(k_estep, dim3(ceil_div(a.w, kEstepPixels), a.h),
                       dim3(kEstepPixels, kEstepLanes), 0, as_stream(stream), a);
    return launch_status(fn);
}
}  // namespace

int emf_hip_estepBatched(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                         const emf_image_t* points, int normalize, const emf_image_t* norm,
                         const emf_image_t* objSum, emf_stream_t stream) {
    return estep_launch(models_dev, poseCO_host, nmodels, nullptr, nullptr, points, normalize, norm, objSum,
                        stream, "estepBatched");
}

int emf_hip_estepBatchedFromDepth(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                                  const emf_image_t* depth, const float K[9], const emf_image_t* points,
                                  int normalize, const emf_image_t* norm, const emf_image_t* objSum,
                                  emf_stream_t stream) {
    if (!depth) return fail(EMF_E_NULL, "estepBatchedFromDepth: depth is NULL");
    return estep_launch(models_dev, poseCO_host, nmodels, depth, K, points, normalize, norm, objSum, stream,
                        "estepBatchedFromDepth");
}

int emf_hip_estepBatchedPeer(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, int nmodels,
                             const emf_image_t* depth, const float K[9], const emf_image_t* points,
                             const emf_peer_t* group, uint32_t seq, emf_stream_t stream) {
    if (!group) return fail(EMF_E_NULL, "estepBatchedPeer: group is NULL");
    return estep_launch(models_dev, poseCO_host, nmodels, depth, K, points, 0, nullptr, nullptr, stream,
                        "estepBatchedPeer", group, seq);
}

int emf_hip_normalizeAssociationTable(const emf_model_t* models_dev, int nmodels, int width, int height, float* norm_dev,
                                      emf_stream_t stream) {
    if (!models_dev) return fail(EMF_E_NULL, "normalizeAssociationTable: models_dev is NULL");
    if (nmodels < 1 || nmodels > EMF_MAX_MODELS) return fail(EMF_E_LIMIT, "normalizeAssociationTable: nmodels = %d", nmodels);
    if (width <= 0 || height <= 0) return fail(EMF_E_SHAPE, "normalizeAssociationTable: bad image size %d x %d", width, height);
    const size_t pixels = static_cast<size_t>(width) * height;
    hipLaunchKernelGGL(k_assoc_normalize_table, dim3(static_cast<unsigned>(ceil_div(pixels, size_t(256)))), dim3(256), 0,
                       as_stream(stream), models_dev, nmodels, norm_dev, pixels);
    return launch_status("normalizeAssociationTable");
}

size_t emf_hip_unseenTileBytes(const int32_t res[3]) { return emf_hip_signMapBytes(res) / 2; }

int emf_hip_rebuildUnseenTiles(const float* tsdf, const float* weights, const int32_t res[3], uint8_t* unseenTiles,
                               emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_REQUIRE_PTR(unseenTiles);
    EMF_REQUIRE_PTR(res);
    EMF_TRY(check_res(res));
    const size_t tiles = emf_hip_unseenTileBytes(res);
    if (tiles > 0x7fffffffu) return fail(EMF_E_LIMIT, "rebuildUnseenTiles: volume too large");
    hipLaunchKernelGGL(k_unseen_tiles, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, as_stream(stream), tsdf,
                       weights, I3{res[0], res[1], res[2]}, unseenTiles);
    return launch_status("rebuildUnseenTiles");
}

size_t emf_hip_signMapBytes(const int32_t res[3]) {
    if (!res || res[0] < 1 || res[1] < 1 || res[2] < 1) return 0;
    return 2 * static_cast<size_t>(ceil_div(res[0], kTileX)) * ceil_div(res[1], kTileY) * ceil_div(res[2], kTileZ);
}

int emf_hip_rebuildSignMaps(const float* tsdf, const int32_t res[3], uint8_t* signMaps, emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(signMaps);
    EMF_REQUIRE_PTR(res);
    EMF_TRY(check_res(res));
    const size_t tiles = emf_hip_signMapBytes(res) / 2;
    if (tiles > 0x7fffffffu) return fail(EMF_E_LIMIT, "rebuildSignMaps: volume too large");
    hipLaunchKernelGGL(k_sign_maps, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, as_stream(stream), tsdf,
                       I3{res[0], res[1], res[2]}, signMaps);
    return launch_status("rebuildSignMaps");
}

size_t emf_hip_raycastFarBoundBytes(int nmodels, int width, int height) {
    if (nmodels < 1 || width < 1 || height < 1) return 0;
    return static_cast<size_t>(nmodels) * (2 * ceil_div(width, kRbTile)) * (2 * ceil_div(height, kRbTile)) * sizeof(float);
}

int emf_hip_raycastFarBounds(const emf_model_t* models_dev, const emf_pose_t* poseCO_host, const int32_t* res_host,
                             int nmodels, int width, int height, const float K[9], uint32_t scanMask, float* bounds_dev,
                             emf_stream_t stream) {
    EMF_TRY(check_batch(models_dev, poseCO_host, nmodels, "raycastFarBounds"));
    EMF_REQUIRE_PTR(res_host);
    EMF_REQUIRE_PTR(K);
    EMF_REQUIRE_PTR(bounds_dev);
    if (width <= 0 || height <= 0) return fail(EMF_E_SHAPE, "raycastFarBounds: bad image size %d x %d", width, height);
    FarBoundArgs a;
    a.models = models_dev;
    a.nmodels = nmodels;
    a.tileStart[0] = 0;
    for (int m = 0; m < nmodels; ++m) {
        EMF_TRY(check_res(res_host + 3 * m));
        a.poses.p[m] = poseCO_host[m];
        // the scan kernel's grid only covers the models the caller asks it to scan
        const size_t tiles = ((scanMask >> m) & 1u) ? emf_hip_signMapBytes(res_host + 3 * m) / 2 : 0;
        if (tiles > static_cast<size_t>(0x7fffffff - a.tileStart[m])) return fail(EMF_E_LIMIT, "raycastFarBounds: too many tiles");
        a.tileStart[m + 1] = a.tileStart[m] + static_cast<int>(tiles);
    }
    a.w = width;
    a.h = height;
    a.cellsX = 2 * static_cast<int>(ceil_div(width, kRbTile));
    a.cellsY = 2 * static_cast<int>(ceil_div(height, kRbTile));
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.bounds = bounds_dev;
    hipLaunchKernelGGL(k_far_init, dim3(ceil_div(nmodels * a.cellsX * a.cellsY, 256)), dim3(256), 0, as_stream(stream), a);
    if (a.tileStart[nmodels] > 0)  // models with sign maps but no relevant-tile list
        hipLaunchKernelGGL(k_far_bounds, dim3(ceil_div(a.tileStart[nmodels], 4)), dim3(256), 0, as_stream(stream), a);
    hipLaunchKernelGGL(k_far_bounds_listed, dim3(nmodels * kFarListBlocks), dim3(256), 0, as_stream(stream), a);
    return launch_status("raycastFarBounds");
}

size_t emf_hip_relevantTileBytes(const int32_t res[3]) {
    if (!res || res[0] < 1 || res[1] < 1 || res[2] < 1) return 0;
    return (1 + emf_hip_signMapBytes(res) / 2) * sizeof(uint32_t);
}

int emf_hip_updateRelevantTiles(const emf_model_t* models_dev, const int32_t* res_host, int nmodels, emf_stream_t stream) {
    if (!models_dev) return fail(EMF_E_NULL, "updateRelevantTiles: models_dev is NULL");
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH) return fail(EMF_E_LIMIT, "updateRelevantTiles: nmodels = %d", nmodels);
    EMF_REQUIRE_PTR(res_host);
    RelevantArgs a;
    a.models = models_dev;
    a.nmodels = nmodels;
    a.tileStart[0] = 0;
    for (int m = 0; m < nmodels; ++m) {
        EMF_TRY(check_res(res_host + 3 * m));
        const size_t tiles = emf_hip_signMapBytes(res_host + 3 * m) / 2;
        if (tiles > static_cast<size_t>(0x7fffffff - a.tileStart[m])) return fail(EMF_E_LIMIT, "updateRelevantTiles: too many tiles");
        a.tileStart[m + 1] = a.tileStart[m] + static_cast<int>(tiles);
    }
    hipLaunchKernelGGL(k_relevant_reset, dim3(1), dim3(64), 0, as_stream(stream), models_dev, nmodels);
    hipLaunchKernelGGL(k_relevant_tiles, dim3(ceil_div(a.tileStart[nmodels], 256)), dim3(256), 0, as_stream(stream), a);
    return launch_status("updateRelevantTiles");
}

namespace {
int raycast_batched_launch(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                           const int32_t* res_host, int nmodels, int width, int height,
                           const float K[9], int useBrickFlags, int bgBandRow0, int bgBandRows,
                           const float* farBounds_dev, const float* voxelSizes_host, uint64_t* stats,
                           emf_stream_t stream, int firstObj, int rowsPerRay) {
    EMF_TRY(check_batch(models_dev, poseCO_host, nmodels, "raycastBatched"));
    if (bgBandRows < 0 || bgBandRow0 < 0 || bgBandRow0 % kRbTile || bgBandRows % kRbTile)
        return fail(EMF_E_ARG, "raycastBatched: band [%d, +%d) must be non-negative multiples of %d rows",
                    bgBandRow0, bgBandRows, kRbTile);
    EMF_REQUIRE_PTR(res_host);
    EMF_REQUIRE_PTR(K);
    if (width <= 0 || height <= 0)
        return fail(EMF_E_SHAPE, "raycastBatched: bad image size %d x %d", width, height);
    RaycastBatchArgs a;
    a.models = models_dev;
    a.divideMask = 0;
    bool offsets32 = true;
    for (int m = 0; m < nmodels; ++m) {
        a.poses.p[m] = poseCO_host[m];
        EMF_TRY(check_res(res_host + 3 * m));
        offsets32 = offsets32 && fits_offsets32(res_host + 3 * m);
        if (usable_reciprocal(1.f, poseCO_host[m].t) == 0.f) a.divideMask |= 1u << m;
    }
    a.nmodels = nmodels;
    a.w = width;
    a.h = height;
    a.tilesX = static_cast<int>(ceil_div(width, kRbTile));
    a.tilesY = static_cast<int>(ceil_div(height, kRbTile));
    a.chunk = static_cast<int>(ceil_div(static_cast<size_t>(a.tilesX) * a.tilesY, 8));
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.stats = reinterpret_cast<unsigned long long*>(stats);
    a.farBounds = farBounds_dev;
    a.bandTile0 = bgBandRow0 / kRbTile;
    a.bandTiles = bgBandRows / kRbTile;
    a.firstObj = firstObj;
    // footprints of the objects: the tiles the (slightly enlarged) volume box projects to
    a.objStart[0] = a.objStart[1] = 0;
    for (int m = firstObj; m < nmodels; ++m) {
        const int32_t* r = res_host + 3 * m;
        const emf_pose_t& p = poseCO_host[m];
        // (the voxel size lives in the device table: without the host's copy the object keeps the whole image)
        const float vs = voxelSizes_host ? voxelSizes_host[m] : 0.f;
        int tx0 = 0, ty0 = 0, tx1 = a.tilesX, ty1 = a.tilesY;
        if (vs > 0.f) {
            float umin = 3e38f, umax = -3e38f, vmin = 3e38f, vmax = -3e38f;
            bool wide = false;
            for (int k = 0; k < 8 && !wide; ++k) {
                const float q[3] = {((k & 1) ? .5f : -.5f) * (r[0] + 2) * vs, ((k & 2) ? .5f : -.5f) * (r[1] + 2) * vs,
                                    ((k & 4) ? .5f : -.5f) * (r[2] + 2) * vs};
                const float d[3] = {q[0] - p.t[0], q[1] - p.t[1], q[2] - p.t[2]};
                const float c[3] = {p.R[0] * d[0] + p.R[3] * d[1] + p.R[6] * d[2], p.R[1] * d[0] + p.R[4] * d[1] + p.R[7] * d[2],
                                    p.R[2] * d[0] + p.R[5] * d[1] + p.R[8] * d[2]};  // R^T d
                if (!(c[2] > 1e-2f * vs)) {
                    wide = true;
                } else {
                    const float u = a.fx * c[0] / c[2] + a.cx, v = a.fy * c[1] / c[2] + a.cy;
                    umin = std::fmin(umin, u); umax = std::fmax(umax, u);
                    vmin = std::fmin(vmin, v); vmax = std::fmax(vmax, v);
                }
            }
            if (!wide) {
                if (!(umax >= -2.f && vmax >= -2.f && umin <= width + 1.f && vmin <= height + 1.f)) {
                    tx1 = tx0 = ty1 = ty0 = 0;  // beside the image: nothing to march
                } else {
                    tx0 = std::max(static_cast<int>(std::floor((umin - 2.f) / kRbTile)), 0);
                    ty0 = std::max(static_cast<int>(std::floor((vmin - 2.f) / kRbTile)), 0);
                    tx1 = std::min(static_cast<int>(std::floor((std::fmin(umax, 1e6f) + 2.f) / kRbTile)) + 1, a.tilesX);
                    ty1 = std::min(static_cast<int>(std::floor((std::fmin(vmax, 1e6f) + 2.f) / kRbTile)) + 1, a.tilesY);
                }
            }
        }
        a.rect[m][0] = static_cast<short>(tx0);
        a.rect[m][1] = static_cast<short>(ty0);
        a.rect[m][2] = static_cast<short>(std::max(tx1 - tx0, 0));
        a.rect[m][3] = static_cast<short>(std::max(ty1 - ty0, 0));
        a.objStart[m + 1] = a.objStart[m] + a.rect[m][2] * a.rect[m][3];
    }
    if (firstObj) a.rect[0][0] = a.rect[0][1] = a.rect[0][2] = a.rect[0][3] = 0;
    // rowsPerRay = 1 / 2 / 4 lanes per BACKGROUND ray (march_lane / march_quad<2> / march_quad<4>); same images
    const int rows = (useBrickFlags || !offsets32 || !firstObj) ? 1 : rowsPerRay;
    const int parts = rows == 4 ? 4 : rows == 2 ? 2 : 1;
    // background: border ring + interior rounded up to whole XCD chunks (or all of it when banded), `parts` workgroups per
    // tile; objects: footprint tiles, and ceil(tiles / kZeroTiles) zero-fill workgroups each (raycast_grid)
    const RaycastGrid g = raycast_grid(a.tilesX, a.tilesY, a.bandTiles, a.chunk, a.objStart[nmodels], nmodels, parts, firstObj);
    if (g.ringBlocks + g.objPad + g.bgBlocks + g.zeroBlocks == 0) return EMF_OK;
    const dim3 grid(static_cast<unsigned>(g.ringBlocks + g.objPad + g.bgBlocks + g.zeroBlocks));
    if (useBrickFlags || !offsets32)  // the wave marches address with 32-bit byte offsets
        hipLaunchKernelGGL(k_raycast_batched<0>, grid, dim3(64 * kRbWaves), 0, as_stream(stream), a);
    else if (parts == 4)
        hipLaunchKernelGGL(k_raycast_batched<4>, grid, dim3(64 * kRbWaves), 0, as_stream(stream), a);
    else if (parts == 2)
        hipLaunchKernelGGL(k_raycast_batched<2>, grid, dim3(64 * kRbWaves), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(k_raycast_batched<1>, grid, dim3(64 * kRbWaves), 0, as_stream(stream), a);
    return launch_status("raycastBatched");
}

}  // namespace

int emf_hip_raycastBatched(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                           const int32_t* res_host, int nmodels, int width, int height,
                           const float K[9], int useBrickFlags, int bgBandRow0, int bgBandRows,
                           const float* farBounds_dev, const float* voxelSizes_host, uint64_t* stats,
                           emf_stream_t stream) {
    return raycast_batched_launch(models_dev, poseCO_host, res_host, nmodels, width, height, K, useBrickFlags, bgBandRow0,
                                  bgBandRows, farBounds_dev, voxelSizes_host, stats, stream, 1, 1);
}

int emf_hip_raycastBatchedLanes(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                                const int32_t* res_host, int nmodels, int width, int height,
                                const float K[9], int useBrickFlags, int bgBandRow0, int bgBandRows,
                                const float* farBounds_dev, const float* voxelSizes_host, int lanesPerBgRay,
                                uint64_t* stats, emf_stream_t stream) {
    if (lanesPerBgRay != 1 && lanesPerBgRay != 2 && lanesPerBgRay != 4)
        return fail(EMF_E_ARG, "raycastBatchedLanes: %d lanes per background ray (1, 2 or 4)", lanesPerBgRay);
    return raycast_batched_launch(models_dev, poseCO_host, res_host, nmodels, width, height, K, useBrickFlags, bgBandRow0,
                                  bgBandRows, farBounds_dev, voxelSizes_host, stats, stream, 1, lanesPerBgRay);
}

int emf_hip_raycastBatchedObjects(const emf_model_t* models_dev, const emf_pose_t* poseCO_host,
                                  const int32_t* res_host, int nmodels, int width, int height,
                                  const float K[9], int useBrickFlags, const float* farBounds_dev,
                                  const float* voxelSizes_host, uint64_t* stats, emf_stream_t stream) {
    return raycast_batched_launch(models_dev, poseCO_host, res_host, nmodels, width, height, K, useBrickFlags, 0, 0,
                                  farBounds_dev, voxelSizes_host, stats, stream, 0, 1);
}

#ifdef EMF_RAY_TRACE
int emf_hip_debugFetchRayTrace(void* host, size_t bytes) {
    (void)hipDeviceSynchronize();
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rayTrace),
                                                bytes < sizeof(g_rayTrace) ? bytes : sizeof(g_rayTrace)));
}
#endif

int emf_hip_integrateBatched(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                             const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                             const emf_image_t* depth, const emf_image_t* invLambda,
                             const float K[9], int maintainBrickFlags, uint64_t* stats,
                             emf_stream_t stream) {
    EMF_TRY(check_batch(models_dev, poseOC_host, nmodels, "integrateBatched"));
    EMF_REQUIRE_PTR(res_host);
    EMF_TRY(check_image(depth, 4, "integrateBatched: depth"));
    if (invLambda) {
        EMF_TRY(check_image(invLambda, 4, "integrateBatched: invLambda"));
        EMF_TRY(check_same_size(depth, invLambda, "depth", "invLambda"));
    }
    EMF_REQUIRE_PTR(K);
    IntegrateBatchArgs a;
    int linStart[EMF_MAX_BATCH + 1];  // 256-voxel chunks of the models that cannot be tiled
    a.models = models_dev;
    a.nmodels = nmodels;
    a.tileStart[0] = 0;
    linStart[0] = 0;
    for (int m = 0; m < nmodels; ++m) {
        const int32_t* r = res_host + 3 * m;
        EMF_TRY(check_res(r));
        a.poses.p[m] = poseOC_host[m];
        const bool tiled = r[0] % 4 == 0;  // float4 tiles; otherwise one voxel per lane
        const size_t voxels = static_cast<size_t>(r[0]) * r[1] * r[2];
        if (!tiled && ceil_div(voxels, size_t(256)) > size_t(0x7fffffff) - linStart[m])
            return fail(EMF_E_LIMIT, "integrateBatched: model %d (Nx = %d, not a multiple of 4) is too "
                        "large for the one-voxel-per-lane launch", m, r[0]);
        a.tileStart[m + 1] = a.tileStart[m] + (tiled ? static_cast<int>(ceil_div(r[0], kTileX)) *
                                                           static_cast<int>(ceil_div(r[1], kTileY)) *
                                                           static_cast<int>(ceil_div(r[2], kTileZ))
                                                     : 0);
        linStart[m + 1] = linStart[m] + (tiled ? 0 : static_cast<int>(ceil_div(voxels, size_t(256))));
    }
    a.visible = visible_dev;
    a.stats = reinterpret_cast<unsigned long long*>(stats);
    a.depth = img<const float>(depth);
    a.invLambda = invLambda ? img<const float>(invLambda) : Img<const float>{nullptr, 0};
    a.w = depth->width;
    a.h = depth->height;
    a.K = m33_from(K);
    a.pinhole = is_pinhole(a.K);
    if (a.tileStart[nmodels] > 0)
        hipLaunchKernelGGL(k_integrate_batched, dim3(static_cast<unsigned>(a.tileStart[nmodels])),
                           dim3(256), 0, as_stream(stream), a);
    if (linStart[nmodels] > 0) {
        for (int m = 0; m <= nmodels; ++m) a.tileStart[m] = linStart[m];
        hipLaunchKernelGGL(k_integrate_batched_linear, dim3(static_cast<unsigned>(linStart[nmodels])),
                           dim3(256), 0, as_stream(stream), a);
    }
    if (!maintainBrickFlags) return launch_status("integrateBatched");
    DilateBatchArgs d;
    d.models = models_dev;
    d.nmodels = nmodels;
    d.visible = visible_dev;
    d.brickStart[0] = 0;
    for (int m = 0; m < nmodels; ++m) {
        const int32_t* r = res_host + 3 * m;
        d.brickStart[m + 1] = d.brickStart[m] + bricks_along(r[0]) * bricks_along(r[1]) *
                                                    bricks_along(r[2]);
    }
    hipLaunchKernelGGL(k_dilate_batched, dim3(ceil_div(d.brickStart[nmodels], 256)), dim3(256), 0,
                       as_stream(stream), d);
    return launch_status("integrateBatched");
}

size_t emf_hip_integrateCullScratchBytes(const int32_t* res_host, int nmodels) {
    size_t boxes = 0;
    for (int m = 0; res_host && m < nmodels; ++m) {
        const int32_t* r = res_host + 3 * m;
        if (r[0] < 2 || r[1] < 2 || r[2] < 2) return 0;
        boxes += static_cast<size_t>(ceil_div(r[0], kBoxX)) * ceil_div(r[1], kBoxY) * ceil_div(r[2], kBoxZ);
    }
    return (boxes + 4) * sizeof(unsigned);
}

size_t emf_hip_integrateDirtyMapBytes(const int32_t res[3]) {
    if (!res || res[0] < 1 || res[1] < 1 || res[2] < 1) return 0;
    // one byte per tile for the tsdf array, then one per tile for the weights
    return 2 * static_cast<size_t>(ceil_div(res[0], kTileX)) * ceil_div(res[1], kTileY) * ceil_div(res[2], kTileZ);
}

int emf_hip_integrateBatchedCulled(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                                   const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                                   const emf_image_t* depth, const emf_image_t* invLambda,
                                   const float K[9], void* scratch_dev, uint32_t launchBoxes,
                                   uint32_t* survivors_out_dev, uint64_t* stats, emf_stream_t stream) {
    return emf_hip_integrateBatchedCulledOut(models_dev, poseOC_host, res_host, nmodels, visible_dev, depth,
                                             invLambda, K, nullptr, 0, scratch_dev, launchBoxes, survivors_out_dev,
                                             stats, stream);
}

int emf_hip_integratePrepareOut(const emf_volume_out_t* out_host, const int32_t* res_host, int nmodels,
                                void* scratch_dev, emf_stream_t stream) {
    EMF_REQUIRE_PTR(res_host);
    EMF_REQUIRE_PTR(scratch_dev);
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH) return fail(EMF_E_LIMIT, "integratePrepareOut: nmodels = %d", nmodels);
    hipError_t e = hipMemsetAsync(scratch_dev, 0, 2 * sizeof(unsigned), as_stream(stream));
    for (int m = 0; e == hipSuccess && out_host && m < nmodels; ++m) {
        if (!out_host[m].dirtyNext) return fail(EMF_E_NULL, "integratePrepareOut: model %d has no dirtyNext map", m);
        EMF_TRY(check_res(res_host + 3 * m));
        e = hipMemsetAsync(out_host[m].dirtyNext, 0, emf_hip_integrateDirtyMapBytes(res_host + 3 * m), as_stream(stream));
    }
    if (e != hipSuccess) {
        set_error("integratePrepareOut: memset: %s", hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return EMF_OK;
}

int emf_hip_integrateBatchedCulledOut(const emf_model_t* models_dev, const emf_pose_t* poseOC_host,
                                      const int32_t* res_host, int nmodels, const int32_t* visible_dev,
                                      const emf_image_t* depth, const emf_image_t* invLambda,
                                      const float K[9], const emf_volume_out_t* out_host, int prepared,
                                      void* scratch_dev, uint32_t launchBoxes, uint32_t* survivors_out_dev,
                                      uint64_t* stats, emf_stream_t stream) {
    EMF_TRY(check_batch(models_dev, poseOC_host, nmodels, "integrateBatchedCulled"));
    EMF_REQUIRE_PTR(res_host);
    EMF_REQUIRE_PTR(scratch_dev);
    EMF_TRY(check_image(depth, 4, "integrateBatchedCulled: depth"));
    if (invLambda) {
        EMF_TRY(check_image(invLambda, 4, "integrateBatchedCulled: invLambda"));
        EMF_TRY(check_same_size(depth, invLambda, "depth", "invLambda"));
    }
    EMF_REQUIRE_PTR(K);
    IntegrateCullArgs a;
    a.b.models = models_dev;
    a.b.nmodels = nmodels;
    a.boxStart[0] = 0;
    for (int m = 0; m < nmodels; ++m) {
        const int32_t* r = res_host + 3 * m;
        EMF_TRY(check_res(r));
        if (r[0] % 4 != 0)
            return fail(EMF_E_SHAPE, "integrateBatchedCulled: model %d has Nx = %d, needs Nx %% 4 == 0 "
                        "(use emf_hip_integrateBatched)", m, r[0]);
        a.b.poses.p[m] = poseOC_host[m];
        a.b.tileStart[m] = 0;
        a.boxStart[m + 1] = a.boxStart[m] + static_cast<int>(ceil_div(r[0], kBoxX)) *
                                                static_cast<int>(ceil_div(r[1], kBoxY)) *
                                                static_cast<int>(ceil_div(r[2], kBoxZ));
    }
    a.b.visible = visible_dev;
    a.b.stats = reinterpret_cast<unsigned long long*>(stats);
    a.b.depth = img<const float>(depth);
    a.b.invLambda = invLambda ? img<const float>(invLambda) : Img<const float>{nullptr, 0};
    a.b.w = depth->width;
    a.b.h = depth->height;
    a.b.K = m33_from(K);
    a.b.pinhole = is_pinhole(a.b.K);
    a.count = static_cast<unsigned*>(scratch_dev);
    a.list = a.count + 4;
    a.out = IntegrateOutTable{};
    a.haveOut = out_host ? 1 : 0;
    const char* dt = std::getenv("EMF_DEEP_TILES");  // (per call: A/B switch, same results)
    const bool deepTiles = !(dt && dt[0] == '0');
    a.deepTiles = deepTiles ? 1 : 0;
    const unsigned total = static_cast<unsigned>(a.boxStart[nmodels]);
    const hipError_t e = prepared ? hipSuccess : hipMemsetAsync(a.count, 0, 2 * sizeof(unsigned), as_stream(stream));
    if (e != hipSuccess) {
        set_error("integrateBatchedCulled: memset: %s", hipGetErrorString(e));
        return static_cast<int>(e);
    }
    if (out_host) {
        // the next-dirty maps start out clean
        for (int m = 0; m < nmodels; ++m) {
            const emf_volume_out_t& o = out_host[m];
            if (!o.tsdf || !o.weights || !o.dirtyPrev || !o.dirtyNext)
                return fail(EMF_E_NULL, "integrateBatchedCulledOut: model %d: tsdf / weights / dirtyPrev / dirtyNext "
                            "of the second copy are all required", m);
            a.out.tsdf[m] = o.tsdf;
            a.out.weights[m] = o.weights;
            a.out.dirtyPrev[m] = o.dirtyPrev;
            a.out.dirtyNext[m] = o.dirtyNext;
            const hipError_t c = prepared ? hipSuccess
                                          : hipMemsetAsync(o.dirtyNext, 0, emf_hip_integrateDirtyMapBytes(res_host + 3 * m),
                                                           as_stream(stream));
            if (c != hipSuccess) {
                set_error("integrateBatchedCulledOut: memset: %s", hipGetErrorString(c));
                return static_cast<int>(c);
            }
        }
    }
    hipLaunchKernelGGL(k_integrate_cull, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), a);
    const unsigned boxes = launchBoxes == 0 || launchBoxes > total ? total : launchBoxes;
    if (out_host) {
        hipLaunchKernelGGL(k_integrate_listed<true>, dim3(kBoxTiles * boxes), dim3(256), 0, as_stream(stream), a);
        if (boxes < total)
            hipLaunchKernelGGL(k_integrate_listed_rest<true>, dim3(kBoxTiles * 64u), dim3(256), 0, as_stream(stream), a, boxes);
    } else {
        hipLaunchKernelGGL(k_integrate_listed<false>, dim3(kBoxTiles * boxes), dim3(256), 0, as_stream(stream), a);
        if (boxes < total)  // an estimate: whatever it missed is swept by a small strided grid
            hipLaunchKernelGGL(k_integrate_listed_rest<false>, dim3(kBoxTiles * 64u), dim3(256), 0, as_stream(stream), a, boxes);
    }
    if (survivors_out_dev) {
        const hipError_t c = hipMemcpyAsync(survivors_out_dev, a.count, sizeof(unsigned), hipMemcpyDeviceToDevice,
                                            as_stream(stream));
        if (c != hipSuccess) {
            set_error("integrateBatchedCulled: count copy: %s", hipGetErrorString(c));
            return static_cast<int>(c);
        }
    }
    return launch_status("integrateBatchedCulled");
}

int emf_hip_visibilityFlags(const int32_t* visCounts, int nmodels, int visibilityThresh,
                            int32_t* visible_dev, int32_t* countsMirror, emf_stream_t stream) {
    EMF_REQUIRE_PTR(visible_dev);
    if (nmodels < 1 || nmodels > EMF_MAX_MODELS)
        return fail(EMF_E_LIMIT, "visibilityFlags: nmodels = %d", nmodels);
    if (nmodels > 1) EMF_REQUIRE_PTR(visCounts);
    hipLaunchKernelGGL(k_vis_flags, dim3(ceil_div(nmodels, 64)), dim3(64), 0, as_stream(stream),
                       visCounts, nmodels, visibilityThresh, visible_dev, countsMirror);
    return launch_status("visibilityFlags");
}

}  // extern "C"
