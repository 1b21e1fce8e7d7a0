// multigpu.hip -- kernels around the second cross-GPU exchange of the object-sharded path
// (SURVEY.md section 8e): merging the nearest raycast hit over the objects of ALL ranks.
//
// The reference composites object raycasts sequentially in list (creation) order with a strict
// '<' (EMFusion.cpp:760-771): the first object reaching the minimum raylength keeps the pixel.
// That is the lexicographic minimum of (raylength, list position), which for positive raylengths
// is the unsigned minimum of the packed key  (float_bits(raylength) << 32) | position.  Each rank
// packs the keys of the objects it owns, ONE all-reduce(min, u64, W*H) over xGMI merges them, and
// every rank finishes the composite (background override, visibility counts) locally.
// Deviation, documented in DESIGN.md: hits with raylength <= 0 (surface behind a camera that sits
// inside an object volume) lose against every positive hit here, whereas the reference's
// `raylength <= 0 || ...` rule lets later objects overwrite them.
#include "peer_core.hpp"

namespace emf_hip {
namespace {

constexpr int kTileX = 64, kTileY = 4;
constexpr unsigned long long kNoHit = ~0ull;

__device__ __forceinline__ bool pixel_of(int w, int h, int& x, int& y) {
    x = blockIdx.x * kTileX + threadIdx.x;
    y = blockIdx.y * kTileY + threadIdx.y;
    return x < w && y < h;
}
inline dim3 pixel_grid(int w, int h) { return dim3(ceil_div(w, kTileX), ceil_div(h, kTileY)); }
inline dim3 pixel_block() { return dim3(kTileX, kTileY); }

constexpr int kLocalMax = EMF_MAX_BATCH;  // objects of one rank handled per launch

struct KeyPackTable {
    Img<const float> ray[kLocalMax];
    Img<const uint8_t> seg[kLocalMax];
    unsigned pos[kLocalMax];  // position of the object in the global creation-order list
    int count;
};

__global__ __launch_bounds__(256) void k_pack_keys(const KeyPackTable t,
                                                   unsigned long long* __restrict__ keys, int w,
                                                   int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    unsigned long long key = kNoHit;
    for (int k = 0; k < t.count; ++k) {
        if (t.seg[k].row(y)[x] == 0) continue;
        const float r = t.ray[k].row(y)[x];
        // positive floats order like their bit patterns; non-positive hits sort last (see header)
        const unsigned bits = r > 0.f ? __float_as_uint(r) : 0xFFFFFFFEu;
        const unsigned long long cand = (static_cast<unsigned long long>(bits) << 32) | t.pos[k];
        key = cand < key ? cand : key;
    }
    keys[static_cast<size_t>(y) * w + x] = key;
}

struct IdTable {
    uint8_t id[EMF_MAX_MODELS];  // saturated object id by list position
};

struct LocalTable {
    Img<const float> vert[kLocalMax], nrm[kLocalMax], ray[kLocalMax];
    unsigned pos[kLocalMax];
    int count;
};

struct FromKeysArgs {
    const unsigned long long* keys;
    Img<const float> bgRay, bgVert, bgNorm;
    Img<const uint8_t> bgMask;
    Img<float> ray, vert, nrm, diff;
    Img<uint8_t> seg, noObj;
    int w, h;
};

__global__ __launch_bounds__(256) void k_composite_keys(const FromKeysArgs a, const IdTable ids,
                                                        const LocalTable loc) {
    int x, y;
    if (!pixel_of(a.w, a.h, x, y)) return;
    const unsigned long long key = a.keys[static_cast<size_t>(y) * a.w + x];
    float r = 0.f;
    V3 vv = v3(0.f, 0.f, 0.f), nn = v3(0.f, 0.f, 0.f);
    uint8_t s = 0;
    if (key != kNoHit) {
        const unsigned pos = static_cast<unsigned>(key & 0xFFFFFFFFull);
        const unsigned bits = static_cast<unsigned>(key >> 32);
        s = ids.id[pos];
        r = __uint_as_float(bits);
        for (int k = 0; k < loc.count; ++k) {
            if (loc.pos[k] != pos) continue;
            // the winner lives on this rank: its vertex / normal / exact raylength are at hand
            r = loc.ray[k].row(y)[x];
            const float* pv = loc.vert[k].row(y) + 3 * x;
            const float* pn = loc.nrm[k].row(y) + 3 * x;
            vv = v3(pv[0], pv[1], pv[2]);
            nn = v3(pn[0], pn[1], pn[2]);
        }
    }
    // from here on identical to the single-GPU composite (EMFusion.cpp:773-794)
    float d = a.diff.row(y)[x];
    if (a.bgMask.row(y)[x]) {
        d = r - a.bgRay.row(y)[x];
        a.diff.row(y)[x] = d;
    }
    if (d > 0.05f) s = 0;
    const uint8_t no = s == 0 ? 255 : 0;
    if (no) {
        const float* pv = a.bgVert.row(y) + 3 * x;
        const float* pn = a.bgNorm.row(y) + 3 * x;
        vv = v3(pv[0], pv[1], pv[2]);
        nn = v3(pn[0], pn[1], pn[2]);
    }
    a.noObj.row(y)[x] = no;
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

struct SlotTable {
    int16_t slot[256];
};

// same counting scheme as pixel_ops.hip's k_vis_counts (EMFusion.cpp:778-791)
__global__ __launch_bounds__(256) void k_vis_counts_all(Img<const uint8_t> seg, int w, int h,
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

struct GateTable {
    int idx[EMF_MAX_BATCH + 1];  // index into counts for each model slot (slot 0 = the background, unused; then <= EMF_MAX_BATCH local objects)
};
__global__ void k_vis_flags_indexed(const int32_t* __restrict__ counts, int nmodels, int thresh,
                                    const GateTable g, int32_t* __restrict__ visible) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nmodels) return;
    visible[s] = s == 0 ? 1 : (counts[g.idx[s]] > thresh ? 1 : 0);
}


// ---- the raycast's exchange over the direct peer-write transport, fused into the path (round 4) ------------
// Slot layout per sender (peer_exchange.hip): [P u64 keys][P f32 background raylengths][P u8 background hit mask],
// P = W * H; of the two background images a sender fills only the rows of its band.
struct PackPeer {
    char* slots[EMF_MAX_PEERS];
    int world;
    int fences;  // emf_peer_t::systemFences (ranks on distinct devices): the stores are followed by a system-scope fence
    size_t off;  // this rank's slot (parity included) in a peer's receive buffer
    size_t pixels;
    int band0, bandRows;
    Img<const float> bgRay;
    Img<const uint8_t> bgMask;
};

__global__ __launch_bounds__(256) void k_pack_keys_peer(const KeyPackTable t, const PackPeer pp, int w, int h) {
    int x, y;
    if (!pixel_of(w, h, x, y)) return;
    unsigned long long key = kNoHit;
    for (int k = 0; k < t.count; ++k) {
        if (t.seg[k].row(y)[x] == 0) continue;
        const float r = t.ray[k].row(y)[x];
        const unsigned bits = r > 0.f ? __float_as_uint(r) : 0xFFFFFFFEu;  // as k_pack_keys
        const unsigned long long cand = (static_cast<unsigned long long>(bits) << 32) | t.pos[k];
        key = cand < key ? cand : key;
    }
    const size_t pix = static_cast<size_t>(y) * w + x;
    for (int p = 0; p < pp.world; ++p)
        __builtin_nontemporal_store(key, reinterpret_cast<unsigned long long*>(pp.slots[p] + pp.off) + pix);
    if (y >= pp.band0 && y < pp.band0 + pp.bandRows) {  // my band of the replicated background's raycast
        const float r = pp.bgRay.row(y)[x];
        const uint8_t m = pp.bgMask.row(y)[x];
        for (int p = 0; p < pp.world; ++p) {
            __builtin_nontemporal_store(r, reinterpret_cast<float*>(pp.slots[p] + pp.off + 8 * pp.pixels) + pix);
            __builtin_nontemporal_store(m, reinterpret_cast<uint8_t*>(pp.slots[p] + pp.off + 12 * pp.pixels) + pix);
        }
    }
    if (pp.fences) __threadfence_system();
}

struct FromKeysPeerArgs {
    Img<float> bgRay;     // foreign bands are filled in from their owners' slots
    Img<uint8_t> bgMask;
    Img<const float> bgVert, bgNorm;
    Img<float> ray, vert, nrm, diff;
    Img<uint8_t> seg, noObj;
    int w, h;
    int bandRowsPerRank;  // 0: the background images are complete locally
    int boundary;
    int32_t* counts;      // zero at the launch's start
};

// A resident grid of at most kPollGroups workgroups walks the 64 x 4 tiles (every workgroup polls the flags at its
// start: peer_exchange.hip, k_peer_normalize); the visibility histogram is kept in LDS over all of a workgroup's tiles.
constexpr int kPollGroups = 256;
__global__ __launch_bounds__(256) void k_composite_keys_peer(const PeerArgs pa, uint32_t seq, const FromKeysPeerArgs a,
                                                             const IdTable ids, const LocalTable loc, const SlotTable slots) {
    __shared__ int lh[256];
    const int tid = threadIdx.y * kTileX + threadIdx.x;
    lh[tid] = 0;
    if (!peer_arrive(pa, seq, tid, blockIdx.x == 0)) return;  // (uniform)
    __syncthreads();
    const int tilesX = (a.w + kTileX - 1) / kTileX, tiles = tilesX * ((a.h + kTileY - 1) / kTileY);
    const size_t P = static_cast<size_t>(a.w) * a.h;
    const char* base = pa.slots[pa.rank];
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int x = (tile % tilesX) * kTileX + threadIdx.x, y = (tile / tilesX) * kTileY + threadIdx.y;
        if (x >= a.w || y >= a.h) continue;
        const size_t pix = static_cast<size_t>(y) * a.w + x;
        unsigned long long key = load_slot8(base + slot_offset(pa, 0, seq) + 8 * pix);
        for (int r = 1; r < pa.world; ++r) {
            const unsigned long long v = load_slot8(base + slot_offset(pa, r, seq) + 8 * pix);
            key = v < key ? v : key;
        }
        float bgR;
        uint8_t bgM;
        const int owner = a.bandRowsPerRank > 0 ? y / a.bandRowsPerRank : pa.rank;
        if (owner != pa.rank && owner < pa.world) {
            const char* so = base + slot_offset(pa, owner, seq);
            bgR = load_slot4(so + 8 * P + 4 * pix);
            bgM = load_slot1(so + 12 * P + pix);
            a.bgRay.row(y)[x] = bgR;
            a.bgMask.row(y)[x] = bgM;
        } else {
            bgR = a.bgRay.row(y)[x];
            bgM = a.bgMask.row(y)[x];
        }
        uint8_t s = 0;
        float r = 0.f;
        V3 vv = v3(0.f, 0.f, 0.f), nn = v3(0.f, 0.f, 0.f);
        if (key != kNoHit) {
            const unsigned pos = static_cast<unsigned>(key & 0xFFFFFFFFull);
            const unsigned bits = static_cast<unsigned>(key >> 32);
            s = ids.id[pos];
            r = __uint_as_float(bits);
            for (int k = 0; k < loc.count; ++k) {
                if (loc.pos[k] != pos) continue;
                r = loc.ray[k].row(y)[x];  // the winner lives on this rank (as k_composite_keys)
                const float* pv = loc.vert[k].row(y) + 3 * x;
                const float* pn = loc.nrm[k].row(y) + 3 * x;
                vv = v3(pv[0], pv[1], pv[2]);
                nn = v3(pn[0], pn[1], pn[2]);
            }
        }
        // from here on identical to the single-GPU composite (EMFusion.cpp:773-794)
        float d = a.diff.row(y)[x];
        if (bgM) {
            d = r - bgR;
            a.diff.row(y)[x] = d;
        }
        if (d > 0.05f) s = 0;
        const uint8_t no = s == 0 ? 255 : 0;
        if (no) {
            const float* pv = a.bgVert.row(y) + 3 * x;
            const float* pn = a.bgNorm.row(y) + 3 * x;
            vv = v3(pv[0], pv[1], pv[2]);
            nn = v3(pn[0], pn[1], pn[2]);
        }
        a.noObj.row(y)[x] = no;
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
        // visibility counts on the values just written (k_vis_counts_all's scheme, EMFusion.cpp:778-791)
        if (s && x >= a.boundary && x < a.w - a.boundary && y >= a.boundary && y < a.h - a.boundary) atomicAdd(&lh[s], 1);
    }
    __syncthreads();
    const int k = slots.slot[tid];
    if (k >= 0 && lh[tid]) atomicAdd(&a.counts[k], lh[tid]);
}

// gate of the owned objects (as k_vis_flags_indexed) + the counts of ALL objects mirrored to host-visible memory,
// visCounts left cleared for the next composite; one workgroup
__global__ __launch_bounds__(256) void k_vis_flags_mirror(int32_t* __restrict__ counts, int nall, int nmodels, int thresh,
                                                          const GateTable g, int32_t* __restrict__ visible,
                                                          int32_t* __restrict__ mirror) {
    const int s = threadIdx.x;
    const int32_t own = (s > 0 && s < nmodels) ? counts[g.idx[s]] : 0;
    const int32_t c = s < nall ? counts[s] : 0;
    __syncthreads();
    if (s < nmodels) visible[s] = s == 0 ? 1 : (own > thresh ? 1 : 0);
    if (s < nall) {
        if (mirror) mirror[s] = c;
        counts[s] = 0;
    }
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

int emf_hip_packHitKeys(int nlocal, const int32_t* listPos_host, const emf_image_t* objRay_host,
                        const emf_image_t* objSeg_host, uint64_t* keys, int width, int height,
                        emf_stream_t stream) {
    EMF_REQUIRE_PTR(keys);
    if (nlocal < 0 || nlocal > kLocalMax)
        return fail(EMF_E_LIMIT, "packHitKeys: nlocal = %d, expected 0..%d", nlocal, kLocalMax);
    if (width <= 0 || height <= 0) return fail(EMF_E_SHAPE, "packHitKeys: bad image size");
    KeyPackTable t;
    t.count = nlocal;
    if (nlocal > 0) {
        EMF_REQUIRE_PTR(listPos_host);
        EMF_REQUIRE_PTR(objRay_host);
        EMF_REQUIRE_PTR(objSeg_host);
    }
    for (int k = 0; k < nlocal; ++k) {
        EMF_TRY(check_image(&objRay_host[k], 4, "packHitKeys: objRay"));
        EMF_TRY(check_image(&objSeg_host[k], 1, "packHitKeys: objSeg"));
        if (objRay_host[k].width != width || objRay_host[k].height != height ||
            objSeg_host[k].width != width || objSeg_host[k].height != height)
            return fail(EMF_E_SHAPE, "packHitKeys: image %d is not %d x %d", k, width, height);
        if (listPos_host[k] < 0 || listPos_host[k] >= EMF_MAX_MODELS)
            return fail(EMF_E_ARG, "packHitKeys: list position %d out of range", listPos_host[k]);
        t.ray[k] = img<const float>(&objRay_host[k]);
        t.seg[k] = img<const uint8_t>(&objSeg_host[k]);
        t.pos[k] = static_cast<unsigned>(listPos_host[k]);
    }
    hipLaunchKernelGGL(k_pack_keys, pixel_grid(width, height), pixel_block(), 0, as_stream(stream),
                       t, reinterpret_cast<unsigned long long*>(keys), width, height);
    return launch_status("packHitKeys");
}

int emf_hip_compositeFromKeys(const uint64_t* keys, int nall, const int32_t* ids_host, int nlocal,
                              const int32_t* listPos_host, const emf_image_t* objRay_host,
                              const emf_image_t* objVert_host, const emf_image_t* objNorm_host,
                              const emf_image_t* bgRay, const emf_image_t* bgVert,
                              const emf_image_t* bgNorm, const emf_image_t* bgMask,
                              const emf_image_t* ray, const emf_image_t* vert,
                              const emf_image_t* norm, const emf_image_t* seg,
                              const emf_image_t* diff, const emf_image_t* noObj, int boundary,
                              int32_t* visCounts, emf_stream_t stream) {
    EMF_REQUIRE_PTR(keys);
    if (nall < 0 || nall > EMF_MAX_MODELS - 1)
        return fail(EMF_E_LIMIT, "compositeFromKeys: nall = %d", nall);
    if (nlocal < 0 || nlocal > kLocalMax || nlocal > nall)
        return fail(EMF_E_LIMIT, "compositeFromKeys: nlocal = %d", nlocal);
    if (nall > 0) {
        EMF_REQUIRE_PTR(ids_host);
        EMF_REQUIRE_PTR(visCounts);
    }
    if (nlocal > 0) {
        EMF_REQUIRE_PTR(listPos_host);
        EMF_REQUIRE_PTR(objRay_host);
        EMF_REQUIRE_PTR(objVert_host);
        EMF_REQUIRE_PTR(objNorm_host);
    }
    EMF_TRY(check_image(bgRay, 4, "compositeFromKeys: bgRay"));
    EMF_TRY(check_image(bgVert, 12, "compositeFromKeys: bgVert"));
    EMF_TRY(check_image(bgNorm, 12, "compositeFromKeys: bgNorm"));
    EMF_TRY(check_image(bgMask, 1, "compositeFromKeys: bgMask"));
    EMF_TRY(check_image(ray, 4, "compositeFromKeys: ray"));
    EMF_TRY(check_image(vert, 12, "compositeFromKeys: vert"));
    EMF_TRY(check_image(norm, 12, "compositeFromKeys: norm"));
    EMF_TRY(check_image(seg, 1, "compositeFromKeys: seg"));
    EMF_TRY(check_image(diff, 4, "compositeFromKeys: diff"));
    EMF_TRY(check_image(noObj, 1, "compositeFromKeys: noObj"));
    const emf_image_t* all[] = {bgVert, bgNorm, bgMask, ray, vert, norm, seg, diff, noObj};
    for (const emf_image_t* im : all) EMF_TRY(check_same_size(im, bgRay, "image", "bgRay"));
    if (boundary < 0) return fail(EMF_E_ARG, "compositeFromKeys: boundary < 0");
    const int w = bgRay->width, h = bgRay->height;

    FromKeysArgs a;
    a.keys = reinterpret_cast<const unsigned long long*>(keys);
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
    IdTable ids;
    SlotTable slots;
    for (int v = 0; v < 256; ++v) {
        ids.id[v] = 0;
        slots.slot[v] = -1;
    }
    for (int k = 0; k < nall; ++k) {
        const int id = ids_host[k];
        ids.id[k] = static_cast<uint8_t>(id < 0 ? 0 : (id > 255 ? 255 : id));
        if (id >= 1 && id <= 255 && slots.slot[id] < 0) slots.slot[id] = static_cast<int16_t>(k);
    }
    LocalTable loc;
    loc.count = nlocal;
    for (int k = 0; k < nlocal; ++k) {
        EMF_TRY(check_image(&objRay_host[k], 4, "compositeFromKeys: objRay"));
        EMF_TRY(check_image(&objVert_host[k], 12, "compositeFromKeys: objVert"));
        EMF_TRY(check_image(&objNorm_host[k], 12, "compositeFromKeys: objNorm"));
        EMF_TRY(check_same_size(&objRay_host[k], bgRay, "objRay", "bgRay"));
        EMF_TRY(check_same_size(&objVert_host[k], bgRay, "objVert", "bgRay"));
        EMF_TRY(check_same_size(&objNorm_host[k], bgRay, "objNorm", "bgRay"));
        if (listPos_host[k] < 0 || listPos_host[k] >= nall)
            return fail(EMF_E_ARG, "compositeFromKeys: list position %d out of range",
                        listPos_host[k]);
        loc.ray[k] = img<const float>(&objRay_host[k]);
        loc.vert[k] = img<const float>(&objVert_host[k]);
        loc.nrm[k] = img<const float>(&objNorm_host[k]);
        loc.pos[k] = static_cast<unsigned>(listPos_host[k]);
    }
    const dim3 g = pixel_grid(w, h), b = pixel_block();
    hipLaunchKernelGGL(k_composite_keys, g, b, 0, as_stream(stream), a, ids, loc);
    EMF_TRY(launch_status("compositeFromKeys"));
    if (nall > 0) {
        const hipError_t e = hipMemsetAsync(visCounts, 0, sizeof(int32_t) * nall, as_stream(stream));
        if (e != hipSuccess) {
            set_error("compositeFromKeys: memset visCounts: %s", hipGetErrorString(e));
            return static_cast<int>(e);
        }
        hipLaunchKernelGGL(k_vis_counts_all, g, b, 0, as_stream(stream), img<const uint8_t>(seg), w,
                           h, boundary, slots, visCounts);
        return launch_status("compositeFromKeys: visibility");
    }
    return EMF_OK;
}

int emf_hip_visibilityFlagsIndexed(const int32_t* visCounts, int nmodels,
                                   const int32_t* countIndex_host, int visibilityThresh,
                                   int32_t* visible_dev, emf_stream_t stream) {
    EMF_REQUIRE_PTR(visible_dev);
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH + 1)
        return fail(EMF_E_LIMIT, "visibilityFlagsIndexed: nmodels = %d", nmodels);
    GateTable g;
    for (int s = 0; s <= EMF_MAX_BATCH; ++s) g.idx[s] = 0;
    if (nmodels > 1) {
        EMF_REQUIRE_PTR(visCounts);
        EMF_REQUIRE_PTR(countIndex_host);
        for (int s = 1; s < nmodels; ++s) g.idx[s] = countIndex_host[s];
    }
    hipLaunchKernelGGL(k_vis_flags_indexed, dim3(1), dim3(64), 0, as_stream(stream), visCounts,
                       nmodels, visibilityThresh, g, visible_dev);
    return launch_status("visibilityFlagsIndexed");
}

size_t emf_hip_peerRaycastSlotBytes(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const size_t P = static_cast<size_t>(width) * height;
    return (13 * P + 15) / 16 * 16;
}

namespace {
int fill_pack_table(KeyPackTable& t, int nlocal, const int32_t* listPos_host, const emf_image_t* objRay_host,
                    const emf_image_t* objSeg_host, int width, int height, const char* fn) {
    if (nlocal < 0 || nlocal > kLocalMax) return fail(EMF_E_LIMIT, "%s: nlocal = %d, expected 0..%d", fn, nlocal, kLocalMax);
    if (width <= 0 || height <= 0) return fail(EMF_E_SHAPE, "%s: bad image size", fn);
    t.count = nlocal;
    if (nlocal > 0 && (!listPos_host || !objRay_host || !objSeg_host)) return fail(EMF_E_NULL, "%s: NULL object table", fn);
    for (int k = 0; k < nlocal; ++k) {
        EMF_TRY(check_image(&objRay_host[k], 4, "packHitKeys: objRay"));
        EMF_TRY(check_image(&objSeg_host[k], 1, "packHitKeys: objSeg"));
        if (objRay_host[k].width != width || objRay_host[k].height != height ||
            objSeg_host[k].width != width || objSeg_host[k].height != height)
            return fail(EMF_E_SHAPE, "%s: image %d is not %d x %d", fn, k, width, height);
        if (listPos_host[k] < 0 || listPos_host[k] >= EMF_MAX_MODELS)
            return fail(EMF_E_ARG, "%s: list position %d out of range", fn, listPos_host[k]);
        t.ray[k] = img<const float>(&objRay_host[k]);
        t.seg[k] = img<const uint8_t>(&objSeg_host[k]);
        t.pos[k] = static_cast<unsigned>(listPos_host[k]);
    }
    return EMF_OK;
}
}  // namespace

int emf_hip_packHitKeysPeer(int nlocal, const int32_t* listPos_host, const emf_image_t* objRay_host,
                            const emf_image_t* objSeg_host, const emf_image_t* bgRay, const emf_image_t* bgMask,
                            int bandRow0, int bandRows, const emf_peer_t* group, uint32_t seq, emf_stream_t stream) {
    EMF_TRY(check_image(bgRay, 4, "packHitKeysPeer: bgRay"));
    EMF_TRY(check_image(bgMask, 1, "packHitKeysPeer: bgMask"));
    EMF_TRY(check_same_size(bgMask, bgRay, "bgMask", "bgRay"));
    const int w = bgRay->width, h = bgRay->height;
    KeyPackTable t;
    EMF_TRY(fill_pack_table(t, nlocal, listPos_host, objRay_host, objSeg_host, w, h, "packHitKeysPeer"));
    PeerArgs pa;
    EMF_TRY(peer_args(group, pa, "packHitKeysPeer"));
    if (emf_hip_peerRaycastSlotBytes(w, h) > pa.slotBytes)
        return fail(EMF_E_ARG, "packHitKeysPeer: a %d x %d raycast needs %zu-byte slots, the group has %zu", w, h,
                    emf_hip_peerRaycastSlotBytes(w, h), pa.slotBytes);
    if (bandRow0 < 0 || bandRows < 0) return fail(EMF_E_ARG, "packHitKeysPeer: band %d + %d", bandRow0, bandRows);
    PackPeer pp;
    for (int p = 0; p < EMF_MAX_PEERS; ++p) pp.slots[p] = p < pa.world ? pa.slots[p] : nullptr;
    pp.world = pa.world;
    pp.fences = pa.fences;
    pp.off = (static_cast<size_t>(seq & 1u) * pa.world + pa.rank) * pa.slotBytes;
    pp.pixels = static_cast<size_t>(w) * h;
    pp.band0 = bandRow0;
    pp.bandRows = bandRows;
    pp.bgRay = img<const float>(bgRay);
    pp.bgMask = img<const uint8_t>(bgMask);
    hipLaunchKernelGGL(k_pack_keys_peer, pixel_grid(w, h), pixel_block(), 0, as_stream(stream), t, pp, w, h);
    return launch_status("packHitKeysPeer");
}

int emf_hip_compositeFromKeysPeer(const emf_peer_t* group, uint32_t seq, int bandRowsPerRank, int nall,
                                  const int32_t* ids_host, int nlocal, const int32_t* listPos_host,
                                  const emf_image_t* objRay_host, const emf_image_t* objVert_host,
                                  const emf_image_t* objNorm_host, const emf_image_t* bgRay, const emf_image_t* bgVert,
                                  const emf_image_t* bgNorm, const emf_image_t* bgMask, const emf_image_t* ray,
                                  const emf_image_t* vert, const emf_image_t* norm, const emf_image_t* seg,
                                  const emf_image_t* diff, const emf_image_t* noObj, int boundary, int32_t* visCounts,
                                  emf_stream_t stream) {
    PeerArgs pa;
    EMF_TRY(peer_args(group, pa, "compositeFromKeysPeer"));
    EMF_REQUIRE_PTR(visCounts);
    if (nall < 0 || nall > EMF_MAX_MODELS - 1) return fail(EMF_E_LIMIT, "compositeFromKeysPeer: nall = %d", nall);
    if (nlocal < 0 || nlocal > kLocalMax || nlocal > nall) return fail(EMF_E_LIMIT, "compositeFromKeysPeer: nlocal = %d", nlocal);
    if (nall > 0) EMF_REQUIRE_PTR(ids_host);
    if (nlocal > 0) {
        EMF_REQUIRE_PTR(listPos_host);
        EMF_REQUIRE_PTR(objRay_host);
        EMF_REQUIRE_PTR(objVert_host);
        EMF_REQUIRE_PTR(objNorm_host);
    }
    EMF_TRY(check_image(bgRay, 4, "compositeFromKeysPeer: bgRay"));
    EMF_TRY(check_image(bgVert, 12, "compositeFromKeysPeer: bgVert"));
    EMF_TRY(check_image(bgNorm, 12, "compositeFromKeysPeer: bgNorm"));
    EMF_TRY(check_image(bgMask, 1, "compositeFromKeysPeer: bgMask"));
    EMF_TRY(check_image(ray, 4, "compositeFromKeysPeer: ray"));
    EMF_TRY(check_image(vert, 12, "compositeFromKeysPeer: vert"));
    EMF_TRY(check_image(norm, 12, "compositeFromKeysPeer: norm"));
    EMF_TRY(check_image(seg, 1, "compositeFromKeysPeer: seg"));
    EMF_TRY(check_image(diff, 4, "compositeFromKeysPeer: diff"));
    EMF_TRY(check_image(noObj, 1, "compositeFromKeysPeer: noObj"));
    const emf_image_t* all[] = {bgVert, bgNorm, bgMask, ray, vert, norm, seg, diff, noObj};
    for (const emf_image_t* im : all) EMF_TRY(check_same_size(im, bgRay, "image", "bgRay"));
    if (boundary < 0 || bandRowsPerRank < 0) return fail(EMF_E_ARG, "compositeFromKeysPeer: boundary / band rows < 0");
    const int w = bgRay->width, h = bgRay->height;
    if (emf_hip_peerRaycastSlotBytes(w, h) > pa.slotBytes)
        return fail(EMF_E_ARG, "compositeFromKeysPeer: a %d x %d raycast needs %zu-byte slots, the group has %zu", w, h,
                    emf_hip_peerRaycastSlotBytes(w, h), pa.slotBytes);
    FromKeysPeerArgs a;
    a.bgRay = img<float>(bgRay);
    a.bgMask = img<uint8_t>(bgMask);
    a.bgVert = img<const float>(bgVert);
    a.bgNorm = img<const float>(bgNorm);
    a.ray = img<float>(ray);
    a.vert = img<float>(vert);
    a.nrm = img<float>(norm);
    a.diff = img<float>(diff);
    a.seg = img<uint8_t>(seg);
    a.noObj = img<uint8_t>(noObj);
    a.w = w;
    a.h = h;
    a.bandRowsPerRank = bandRowsPerRank;
    a.boundary = boundary;
    a.counts = visCounts;
    IdTable ids;
    SlotTable slots;
    for (int v = 0; v < 256; ++v) {
        ids.id[v] = 0;
        slots.slot[v] = -1;
    }
    for (int k = 0; k < nall; ++k) {
        const int id = ids_host[k];
        ids.id[k] = static_cast<uint8_t>(id < 0 ? 0 : (id > 255 ? 255 : id));
        if (id >= 1 && id <= 255 && slots.slot[id] < 0) slots.slot[id] = static_cast<int16_t>(k);
    }
    LocalTable loc;
    loc.count = nlocal;
    for (int k = 0; k < nlocal; ++k) {
        EMF_TRY(check_image(&objRay_host[k], 4, "compositeFromKeysPeer: objRay"));
        EMF_TRY(check_image(&objVert_host[k], 12, "compositeFromKeysPeer: objVert"));
        EMF_TRY(check_image(&objNorm_host[k], 12, "compositeFromKeysPeer: objNorm"));
        EMF_TRY(check_same_size(&objRay_host[k], bgRay, "objRay", "bgRay"));
        EMF_TRY(check_same_size(&objVert_host[k], bgRay, "objVert", "bgRay"));
        EMF_TRY(check_same_size(&objNorm_host[k], bgRay, "objNorm", "bgRay"));
        if (listPos_host[k] < 0 || listPos_host[k] >= nall)
            return fail(EMF_E_ARG, "compositeFromKeysPeer: list position %d out of range", listPos_host[k]);
        loc.ray[k] = img<const float>(&objRay_host[k]);
        loc.vert[k] = img<const float>(&objVert_host[k]);
        loc.nrm[k] = img<const float>(&objNorm_host[k]);
        loc.pos[k] = static_cast<unsigned>(listPos_host[k]);
    }
    EMF_TRY(peer_wait_in_front(group, seq, stream));
    const unsigned tiles = ceil_div(w, kTileX) * ceil_div(h, kTileY);
    hipLaunchKernelGGL(k_composite_keys_peer, dim3(group->waitInFront || tiles < kPollGroups ? tiles : kPollGroups), pixel_block(), 0,
                       as_stream(stream), pa, seq, a, ids, loc, slots);
    return launch_status("compositeFromKeysPeer");
}

int emf_hip_visibilityFlagsMirror(int32_t* visCounts, int nall, int nmodels, const int32_t* countIndex_host,
                                  int visibilityThresh, int32_t* visible_dev, int32_t* countsMirror, emf_stream_t stream) {
    EMF_REQUIRE_PTR(visible_dev);
    EMF_REQUIRE_PTR(visCounts);
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH + 1) return fail(EMF_E_LIMIT, "visibilityFlagsMirror: nmodels = %d", nmodels);
    if (nall < 0 || nall > EMF_MAX_MODELS - 1) return fail(EMF_E_LIMIT, "visibilityFlagsMirror: nall = %d", nall);
    GateTable g;
    for (int s = 0; s <= EMF_MAX_BATCH; ++s) g.idx[s] = 0;
    if (nmodels > 1) {
        EMF_REQUIRE_PTR(countIndex_host);
        for (int s = 1; s < nmodels; ++s) {
            if (countIndex_host[s] < 0 || countIndex_host[s] >= nall)
                return fail(EMF_E_ARG, "visibilityFlagsMirror: count index %d out of range", countIndex_host[s]);
            g.idx[s] = countIndex_host[s];
        }
    }
    hipLaunchKernelGGL(k_vis_flags_mirror, dim3(1), dim3(256), 0, as_stream(stream), visCounts, nall, nmodels,
                       visibilityThresh, g, visible_dev, countsMirror);
    return launch_status("visibilityFlagsMirror");
}

}  // extern "C"
