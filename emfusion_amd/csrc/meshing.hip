// meshing.hip -- marching cubes over a TSDF volume: the mesh that
// TSDF::getMesh / ObjTSDF::getMesh hand to the PLY writer (reference TSDF.cu:855-1152,
// TSDF.cpp:356-373, ObjTSDF.cpp:247-268).
//
// The reference classifies every cube into three N^3-sized buffers (class u8, vertex count i32,
// triangle count i32 -- 9 bytes per voxel, 1.2 GB for the 512^3 background), sums them, runs two
// device-wide thrust::exclusive_scan passes over them and reads them back in the emit kernel.
// Here nothing per cube is stored.  The cube anchored at voxel (x, y, z) is visited in voxel order
// (= the reference's buffer order with empty slots where x, y or z is the last index) in chunks of
// 252 positions -- 4 waves of 63 cubes; lane 63 only lends its voxel to lane 62 -- and only per-chunk
// numbers go through memory (9 bytes per chunk):
//   k_mesh_count  a workgroup classifies 32 chunks (8 for small volumes).  A lane loads ITS voxel of the four rows
//                 (y, z), (y+1, z), (y, z+1), (y+1, z+1) -- 8 coalesced loads instead of 16
//                 gathers -- and the "observed" and "negative" predicates become 64-bit wave masks
//                 (a v_cmp each); the x+1 neighbour is the mask shifted by one, so which cubes are
//                 complete and which hold a sign change is a handful of SCALAR and / or / shift
//                 instructions per wave.  Only lanes with surface (rare) build their class.  Per
//                 chunk the packed (vertices, triangles) total -> chunkTot[g], per workgroup their
//                 sum -> blockSums[b], and the ids of the chunks that hold surface -> list[]
//   k_mesh_scan   one workgroup: exclusive scan of blockSums in place (17 k pairs at 512^3), totals
//   k_mesh_emit   a fixed grid walks list[]: classify the chunk again, scan inside the workgroup (wave
//                 shuffles + LDS) on top of blockSums[g / 8] + the chunk totals before it, and write
//                 vertices, normals and triangles where the reference puts them
// so the volume is streamed once (count) plus the surface chunks once more (emit; a workgroup per
// chunk that returns when its total is 0 measured 0.57 ms at 512^3 for launching 0.5 M workgroups
// alone).  Counting workgroups are mapped so that each XCD walks a contiguous eighth of the volume
// (its own z-slabs stay in its L2).
// The output is element-for-element the reference's: cubes in (z, y, x) order, a cube's vertices
// in edge-bit order, triangles as (3, i0, i1, i2) in table order.  Normals are the interpolated RAW
// gradients: the reference's `ns[i] /= norm(ns[i])` and `normals[..] /= norm(..)` call an
// operator/= that takes its left side by const reference and returns the quotient (common.cuh:170-173),
// so nothing is normalised (quirk Q19).
#include "mesh_core.hpp"

#include "mc_tables.h"

namespace emf_hip {
namespace {

constexpr int kMcBlock = 256;
constexpr int kMcWaveCubes = 63;                               // positions per wave (lanes 0..62)
constexpr int kMcChunk = kMcWaveCubes * (kMcBlock / 64);       // 252 positions per chunk
// chunks per counting workgroup: 32 for large volumes (the one-workgroup scan over the per-workgroup
// sums shrinks: 0.52 -> 0.455 ms count + scan at 512^3), 8 for small ones (a 128^3 object would
// otherwise be 260 workgroups of 32 serial steps: 0.020 -> 0.055 ms)
constexpr int kMcChunksLarge = 32, kMcChunksSmall = 8;
constexpr size_t kMcLargeVoxels = size_t(1) << 24;
inline int chunks_for(size_t nvox) { return nvox >= kMcLargeVoxels ? kMcChunksLarge : kMcChunksSmall; }
constexpr unsigned kXcds = 8;

// Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2.  A cube needs the
// planes z and z + 1, so every plane is read twice, one plane's worth of workgroups apart -- 2 MiB of
// tsdf + weights per 512^2 plane, which does not survive in an L2 that streams the whole plane.
// Each XCD therefore takes the same BAND of every plane (an eighth of its rows) and walks z: the data
// it has to keep between the two uses is an eighth of a plane.  `wpp` = workgroups per plane.
__device__ __forceinline__ unsigned logical_block(unsigned nblocks, unsigned wpp) {
    const unsigned band = (wpp + kXcds - 1) / kXcds;
    const unsigned k = blockIdx.x % kXcds, i = blockIdx.x / kXcds;
    const unsigned col = k * band + i % band;
    const unsigned b = (i / band) * wpp + col;
    return col < wpp ? b : nblocks;  // nblocks = nothing to do
}

struct MeshArgs {
    MeshSource src;
    const float* grads;  // N^3 x 3 gradient volume, or nullptr: forward differences on the fly
    uint2* blockSums;    // per counting workgroup (vertices, triangles); after k_mesh_scan their exclusive scan
    unsigned* chunkTot;  // per chunk: vertices | triangles << 16 (at most 3072 and 1280)
    unsigned* list;      // ids of the chunks with a non-zero total, in no particular order
    unsigned* listCount;
    emf_mesh_counts_t* counts;
    float* vertices;
    float* normals;
    int32_t* triangles;
    unsigned nblocks;
    unsigned wpp;     // counting workgroups per z plane (at least 1)
    unsigned chunks;  // chunks per counting workgroup (kMcChunksLarge or kMcChunksSmall)
};

struct Cube {
    int x, y, z;
    size_t base;
    unsigned cls;  // 0 when the cube is masked out or carries no surface
};

// Voxel coordinates of a linear position.  Dividing is done once per workgroup (wave-uniform
// values); chunks, waves and lanes are reached from there by carrying -- two 64-bit divisions per
// cube were most of the first counting kernel's time.
struct Origin {
    unsigned x, y, z;
};

__device__ __forceinline__ Origin origin_of(const I3& n, size_t p) {
    const unsigned nx = n.x, ny = n.y;
    const size_t r = p / nx;
    return Origin{static_cast<unsigned>(p - r * nx), static_cast<unsigned>(r % ny), static_cast<unsigned>(r / ny)};
}

__device__ __forceinline__ Origin advanced(const I3& n, Origin o, unsigned by) {
    const unsigned nx = n.x, ny = n.y;
    o.x += by;
    while (o.x >= nx) {  // at most once per wave when Nx >= 64
        o.x -= nx;
        ++o.y;
    }
    while (o.y >= ny) {
        o.y -= ny;
        ++o.z;
    }
    return o;
}

// the cube anchored at voxel `o` (emit pass: its 16 corner values are gathered by the lane itself)
__device__ __forceinline__ Cube classify(const MeshSource& s, const Origin& o, bool inRange) {
    Cube q{0, 0, 0, 0, 0u};
    const I3 n = s.n;
    if (!inRange || o.x + 1 >= static_cast<unsigned>(n.x) || o.y + 1 >= static_cast<unsigned>(n.y) ||
        o.z + 1 >= static_cast<unsigned>(n.z))
        return q;
    q.x = static_cast<int>(o.x);
    q.y = static_cast<int>(o.y);
    q.z = static_cast<int>(o.z);
    const size_t sy = static_cast<size_t>(n.x), sz = sy * n.y;
    q.base = static_cast<size_t>(q.z) * sz + static_cast<size_t>(q.y) * sy + q.x;
    bool valid = true;  // kernel_classifyCubes: all 8 corners observed (and foreground)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int dx, dy, dz;
        cube_corner(i, dx, dy, dz);
        const size_t idx = q.base + dx + dy * sy + dz * sz;
        valid = valid && s.weights[idx] > 0.f && (!s.fg || s.fg[idx] != 0);
    }
    if (!valid) return q;
    unsigned cls = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int dx, dy, dz;
        cube_corner(i, dx, dy, dz);
        cls |= (s.tsdf[q.base + dx + dy * sy + dz * sz] < 0.f ? 1u : 0u) << i;
    }
    q.cls = cls == 255u ? 0u : cls;
    return q;
}

// The same decision for the 63 cubes of a wave at once (count pass).  `ow` / `pw`: coordinates and
// linear position of the wave's first voxel (wave-uniform); lane l holds voxel pw + l.  Returns the
// lane's class (0: no surface here); all lanes of the wave call together.
__device__ __forceinline__ unsigned classify_wave(const MeshSource& s, const Origin& ow, size_t pw, size_t nvox,
                                                  int lane) {
    const I3 n = s.n;
    const unsigned nx = n.x, ny = n.y, nz = n.z;
    const size_t sy = static_cast<size_t>(nx), sz = sy * ny;
    unsigned x = ow.x + lane, y = ow.y, z = ow.z;
    if (nx >= 64u) {  // a wave wraps at most once
        if (x >= nx) {
            x -= nx;
            ++y;
        }
        if (y >= ny) {
            y -= ny;
            ++z;
        }
    } else {
        while (x >= nx) {
            x -= nx;
            ++y;
        }
        while (y >= ny) {
            y -= ny;
            ++z;
        }
    }
    const bool in = pw + lane < nvox;
    const bool rowY = in && y + 1 < ny, rowZ = in && z + 1 < nz;
    const bool row[4] = {in, rowY, rowZ, rowY && rowZ};
    const size_t rowOff[4] = {0, sy, sz, sy + sz};
    float w[4], t[4];
    unsigned char g[4];
    if (pw + 64 + sy + sz <= nvox) {
        // every address of the wave lies inside the arrays (all but the last plane): scalar row bases,
        // the lane index as the only per-lane part of the address.  A row that does not exist for a
        // lane (y + 1 == Ny, z + 1 == Nz) yields a neighbouring row's values; V masks them out.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w[r] = (s.weights + pw + rowOff[r])[lane];
            t[r] = (s.tsdf + pw + rowOff[r])[lane];
            g[r] = s.fg ? (s.fg + pw + rowOff[r])[lane] : 1;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // clamped addresses, still unconditional loads
            const size_t q = row[r] ? pw + lane + rowOff[r] : 0;
            w[r] = s.weights[q];
            t[r] = s.tsdf[q];
            g[r] = s.fg ? s.fg[q] : 1;
        }
    }
    unsigned long long V[4], N[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        V[r] = __ballot(row[r] && w[r] > 0.f && g[r] != 0);
        N[r] = __ballot(t[r] < 0.f);  // only read where V says the cube is complete
    }
    // lanes 0..62 whose x + 1 exists; the rows' existence is already in V
    unsigned long long complete = __ballot(x + 1 < nx) & 0x7fffffffffffffffull;
    unsigned long long all = ~0ull, any = 0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        complete &= V[r] & (V[r] >> 1);
        all &= N[r] & (N[r] >> 1);
        any |= N[r] | (N[r] >> 1);
    }
    const unsigned long long surface = complete & any & ~all;  // both signs among the 8 corners
    if (surface == 0ull) return 0u;  // wave-uniform: the common case
    if (!((surface >> lane) & 1ull)) return 0u;
    auto bit = [&](unsigned long long m, int d) { return static_cast<unsigned>((m >> (lane + d)) & 1ull); };
    // corners 0:(0,0,0) 1:(1,0,0) 2:(1,0,1) 3:(0,0,1) 4:(0,1,0) 5:(1,1,0) 6:(1,1,1) 7:(0,1,1); rows (dy, dz):
    // N[0] = (0,0), N[1] = (1,0), N[2] = (0,1), N[3] = (1,1)
    return bit(N[0], 0) | bit(N[0], 1) << 1 | bit(N[2], 1) << 2 | bit(N[2], 0) << 3 | bit(N[1], 0) << 4 |
           bit(N[1], 1) << 5 | bit(N[3], 1) << 6 | bit(N[3], 0) << 7;
}

__device__ __forceinline__ unsigned triangles_of(unsigned cls) {
    unsigned n = 0;
    while (n < 5 && emf_mc_tri_table[cls][3 * n] >= 0) ++n;
    return n;
}

// sums over the workgroup (all lanes get the totals) and the exclusive prefix of this lane
__device__ __forceinline__ uint2 block_scan(uint2 v, uint2& total, uint2* lds /* [8] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned ax = __shfl_up(inc.x, o), ay = __shfl_up(inc.y, o);
        if (lane >= o) {
            inc.x += ax;
            inc.y += ay;
        }
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint2 before = make_uint2(0u, 0u);
    total = make_uint2(0u, 0u);
#pragma unroll
    for (int w = 0; w < kMcBlock / 64; ++w) {
        const uint2 t = lds[w];
        if (w < wave) {
            before.x += t.x;
            before.y += t.y;
        }
        total.x += t.x;
        total.y += t.y;
    }
    __syncthreads();
    return make_uint2(before.x + inc.x - v.x, before.y + inc.y - v.y);
}

template <int kMcChunks>
__global__ __launch_bounds__(kMcBlock) void k_mesh_count(const MeshArgs a) {
    constexpr int kMcSpan = kMcChunk * kMcChunks;
    __shared__ unsigned lds[kMcBlock / 64][kMcChunks];
    const unsigned b = logical_block(a.nblocks, a.wpp);
    if (b >= a.nblocks) return;
    const I3 n = a.src.n;
    const size_t nvox = static_cast<size_t>(n.x) * n.y * n.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned p[kMcChunks];
    size_t pw = static_cast<size_t>(b) * kMcSpan + static_cast<size_t>(wave) * kMcWaveCubes;  // this wave, chunk 0
    Origin o = origin_of(n, pw);
#pragma unroll
    for (int c = 0; c < kMcChunks; ++c) {
        const unsigned cls = classify_wave(a.src, o, pw, nvox, lane);
        p[c] = cls ? __popc(active_edges(cls)) | (triangles_of(cls) << 16) : 0u;
        pw += kMcChunk;
        o = advanced(n, o, kMcChunk);
    }
#pragma unroll
    for (int c = 0; c < kMcChunks; ++c) {
        if (__ballot(p[c] != 0u) != 0ull) {  // wave-uniform: most waves hold no surface
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) p[c] += __shfl_xor(p[c], o2);  // fields cannot carry: <= 756, 315 per wave
        }
        if (lane == 0) lds[wave][c] = p[c];
    }
    __syncthreads();
    if (threadIdx.x < kMcChunks) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < kMcBlock / 64; ++w) t += lds[w][threadIdx.x];
        const unsigned g = b * kMcChunks + threadIdx.x;
        a.chunkTot[g] = t;
        if (t) a.list[atomicAdd(a.listCount, 1u)] = g;
        uint2 sum = make_uint2(t & 0xffffu, t >> 16);
#pragma unroll
        for (int o2 = 1; o2 < kMcChunks; o2 <<= 1) {
            sum.x += __shfl_xor(sum.x, o2);
            sum.y += __shfl_xor(sum.y, o2);
        }
        if (threadIdx.x == 0) a.blockSums[b] = sum;
    }
}

// one workgroup walks the per-workgroup sums in chunks of 1024 (a 512^3 volume has 65 k of them)
__global__ __launch_bounds__(1024) void k_mesh_scan(const MeshArgs a) {
    __shared__ uint2 lds[16];
    __shared__ uint2 carry;
    if (threadIdx.x == 0) carry = make_uint2(0u, 0u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned start = 0; start < a.nblocks; start += 1024) {
        const unsigned i = start + threadIdx.x;
        const uint2 v = i < a.nblocks ? a.blockSums[i] : make_uint2(0u, 0u);
        uint2 inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned ax = __shfl_up(inc.x, o), ay = __shfl_up(inc.y, o);
            if (lane >= o) {
                inc.x += ax;
                inc.y += ay;
            }
        }
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        uint2 before = carry, total = make_uint2(0u, 0u);
        for (int w = 0; w < 16; ++w) {
            const uint2 t = lds[w];
            if (w < wave) {
                before.x += t.x;
                before.y += t.y;
            }
            total.x += t.x;
            total.y += t.y;
        }
        if (i < a.nblocks) a.blockSums[i] = make_uint2(before.x + inc.x - v.x, before.y + inc.y - v.y);
        __syncthreads();
        if (threadIdx.x == 0) {
            carry.x += total.x;
            carry.y += total.y;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.counts->vertices = carry.x;
        a.counts->triangles = carry.y;
    }
}

// gradient of the corner voxel: the gradient volume if there is one, else what
// kernel_computeTSDFGrads would have stored there (forward differences, zero on the last planes)
__device__ __forceinline__ V3 corner_gradient(const MeshArgs& a, size_t idx, int x, int y, int z) {
    if (a.grads) return v3(a.grads[3 * idx], a.grads[3 * idx + 1], a.grads[3 * idx + 2]);
    const I3 n = a.src.n;
    if (x >= n.x - 1 || y >= n.y - 1 || z >= n.z - 1) return v3(0.f, 0.f, 0.f);
    const size_t sy = static_cast<size_t>(n.x), sz = sy * n.y;
    const float t = a.src.tsdf[idx];
    return v3(a.src.tsdf[idx + 1] - t, a.src.tsdf[idx + sy] - t, a.src.tsdf[idx + sz] - t);
}

__device__ __forceinline__ void emit_cube(const MeshArgs& a, const Cube& q, unsigned edges, unsigned ntris,
                                          unsigned vertBase, unsigned triBase);

__global__ __launch_bounds__(kMcBlock) void k_mesh_emit(const MeshArgs a) {
    __shared__ uint2 lds[8];
    const I3 n = a.src.n;
    const size_t nvox = static_cast<size_t>(n.x) * n.y * n.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned todo = *a.listCount;
    for (unsigned i = blockIdx.x; i < todo; i += gridDim.x) {
        const unsigned g = a.list[i];
        const unsigned first = g & ~(a.chunks - 1u);
        uint2 base = a.blockSums[g / a.chunks];
        for (unsigned c = first; c < g; ++c) {
            const unsigned t = a.chunkTot[c];
            base.x += t & 0xffffu;
            base.y += t >> 16;
        }
        const size_t p = static_cast<size_t>(g) * kMcChunk + static_cast<size_t>(wave) * kMcWaveCubes + lane;
        const Cube q = classify(a.src, origin_of(n, p), lane < kMcWaveCubes && p < nvox);
        const unsigned edges = q.cls ? active_edges(q.cls) : 0u;
        uint2 v = make_uint2(0u, 0u);
        if (q.cls) v = make_uint2(__popc(edges), triangles_of(q.cls));
        uint2 total;
        const uint2 mine = block_scan(v, total, lds);
        // (3, i0, i1, i2) per triangle
        if (q.cls) emit_cube(a, q, edges, v.y, base.x + mine.x, 4u * (base.y + mine.y));
    }
}

__device__ __forceinline__ void emit_cube(const MeshArgs& a, const Cube& q, unsigned edges, unsigned ntris,
                                          unsigned vertBase, unsigned triBase) {
    const I3 n = a.src.n;
    const size_t sy = static_cast<size_t>(n.x), sz = sy * n.y;
    const V3 half = half_extent(n);
    int offsets[12];
    unsigned k = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        offsets[e] = 0;
        if (!((edges >> e) & 1u)) continue;
        V3 p[2], g[2];
        float val[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int dx, dy, dz;
            cube_corner(emf_mc_edge_corner[e][s], dx, dy, dz);
            const size_t idx = q.base + dx + dy * sy + dz * sz;
            val[s] = a.src.tsdf[idx];
            // ( x + 1 - ( volSize.x - 1 ) / 2.f ) * voxelSize  (TSDF.cu:945-968)
            p[s] = v3((static_cast<float>(q.x + dx) - half.x) * a.src.voxelSize,
                      (static_cast<float>(q.y + dy) - half.y) * a.src.voxelSize,
                      (static_cast<float>(q.z + dz) - half.z) * a.src.voxelSize);
            g[s] = corner_gradient(a, idx, q.x + dx, q.y + dy, q.z + dz);
        }
        const V3 pv = vertex_interp(p[0], p[1], val[0], val[1]);
        const V3 nv = vertex_interp(g[0], g[1], val[0], val[1]);  // not normalised: Q19
        float* vo = a.vertices + 3 * static_cast<size_t>(vertBase + k);
        float* no = a.normals + 3 * static_cast<size_t>(vertBase + k);
        vo[0] = pv.x;
        vo[1] = pv.y;
        vo[2] = pv.z;
        no[0] = nv.x;
        no[1] = nv.y;
        no[2] = nv.z;
        offsets[e] = static_cast<int>(k++);
    }
    for (unsigned t = 0; t < ntris; ++t) {
        int32_t* to = a.triangles + triBase + 4 * t;
        to[0] = 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = emf_mc_tri_table[q.cls][3 * t + j];
            int o = 0;  // offsets[e] without a dynamically indexed register array
#pragma unroll
            for (int i = 0; i < 12; ++i) o = e == i ? offsets[i] : o;
            to[1 + j] = static_cast<int32_t>(vertBase) + o;
        }
    }
}

// grid of the counting pass: 8 XCDs x ceil(wpp / 8) columns x planes (see logical_block)
unsigned launch_blocks(unsigned nblocks, unsigned wpp) {
    const unsigned band = (wpp + kXcds - 1) / kXcds, rows = (nblocks + wpp - 1) / wpp;
    return kXcds * band * rows;
}

int fill_args(MeshArgs& a, const float* tsdf, const float* weights, const uint8_t* fg,
              const int32_t res[3], float voxelSize, void* scratch) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(weights);
    EMF_REQUIRE_PTR(scratch);
    EMF_TRY(check_res(res));
    const size_t nvox = static_cast<size_t>(res[0]) * res[1] * res[2];
    if ((nvox + kMcChunk - 1) / kMcChunk > 0x7ffffff0ull)
        return fail(EMF_E_LIMIT, "mesh: %zu voxels exceed one launch", nvox);
    const unsigned chunks = static_cast<unsigned>(chunks_for(nvox));
    const size_t span = static_cast<size_t>(kMcChunk) * chunks;
    const unsigned nblocks = static_cast<unsigned>((nvox + span - 1) / span);
    a.src = MeshSource{tsdf, weights, fg, i3_from(res), voxelSize};
    a.grads = nullptr;
    a.blockSums = static_cast<uint2*>(scratch);
    a.chunkTot = reinterpret_cast<unsigned*>(a.blockSums + nblocks);
    a.list = a.chunkTot + static_cast<size_t>(nblocks) * chunks;
    a.listCount = a.list + static_cast<size_t>(nblocks) * chunks;
    a.chunks = chunks;
    a.counts = nullptr;
    a.vertices = a.normals = nullptr;
    a.triangles = nullptr;
    a.nblocks = nblocks;
    const size_t plane = static_cast<size_t>(res[0]) * res[1];
    a.wpp = static_cast<unsigned>(plane / span > 0 ? plane / span : 1);
    return EMF_OK;
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

size_t emf_hip_meshScratchBytes(const int32_t res[3]) {
    if (!res || res[0] < 2 || res[1] < 2 || res[2] < 2) return 0;
    const size_t nvox = static_cast<size_t>(res[0]) * res[1] * res[2];
    const size_t chunks = static_cast<size_t>(chunks_for(nvox)), span = kMcChunk * chunks;
    return ((nvox + span - 1) / span) * (sizeof(uint2) + 2 * chunks * sizeof(unsigned)) + 16;
}

int emf_hip_meshCount(const float* tsdf, const float* weights, const uint8_t* fgVolMask,
                      const int32_t res[3], void* scratch_dev, emf_mesh_counts_t* counts_dev,
                      emf_stream_t stream) {
    MeshArgs a;
    EMF_TRY(fill_args(a, tsdf, weights, fgVolMask, res, 1.f, scratch_dev));
    EMF_REQUIRE_PTR(counts_dev);
    a.counts = counts_dev;
    const hipError_t e = hipMemsetAsync(a.listCount, 0, sizeof(unsigned), as_stream(stream));
    if (e != hipSuccess) {
        set_error("meshCount: memset: %s", hipGetErrorString(e));
        return static_cast<int>(e);
    }
    if (a.chunks == static_cast<unsigned>(kMcChunksLarge))
        hipLaunchKernelGGL(k_mesh_count<kMcChunksLarge>, dim3(launch_blocks(a.nblocks, a.wpp)), dim3(kMcBlock), 0,
                           as_stream(stream), a);
    else
        hipLaunchKernelGGL(k_mesh_count<kMcChunksSmall>, dim3(launch_blocks(a.nblocks, a.wpp)), dim3(kMcBlock), 0,
                           as_stream(stream), a);
    hipLaunchKernelGGL(k_mesh_scan, dim3(1), dim3(1024), 0, as_stream(stream), a);
    return launch_status("meshCount");
}

int emf_hip_meshEmit(const float* tsdf, const float* grads, const float* weights,
                     const uint8_t* fgVolMask, const int32_t res[3], float voxelSize,
                     const void* scratch_dev, float* vertices, float* normals, int32_t* triangles,
                     emf_stream_t stream) {
    MeshArgs a;
    EMF_TRY(fill_args(a, tsdf, weights, fgVolMask, res, voxelSize, const_cast<void*>(scratch_dev)));
    EMF_REQUIRE_PTR(vertices);
    EMF_REQUIRE_PTR(normals);
    EMF_REQUIRE_PTR(triangles);
    a.grads = grads;
    a.vertices = vertices;
    a.normals = normals;
    a.triangles = triangles;
    const unsigned nchunks = a.nblocks * a.chunks;  // fixed grid over the list of surface chunks
    hipLaunchKernelGGL(k_mesh_emit, dim3(nchunks < 4096u ? nchunks : 4096u), dim3(kMcBlock), 0, as_stream(stream), a);
    return launch_status("meshEmit");
}

}  // extern "C"
