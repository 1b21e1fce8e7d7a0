/*
 * emf_oracle.c -- CPU restatement of EM-Fusion's per-frame volumetric hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY -- see emf_oracle.h.  PARITY UNPINNED (no reference tests / golden
 * vectors exist and the reference cannot be built in this image) -- see emf_oracle.h.
 *
 * Each function cites the reference lines (relative to the reference repository root) whose
 * behaviour it restates.  Nothing here is copied: the loops, names and structure are our own, the
 * *arithmetic order* is the reference's, because ray-march step decisions and pixel roundings
 * depend on it.  Compile with -ffp-contract=off.
 */
#include "emf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
typedef struct { v3 r0, r1, r2; } m33;

static int g_threads = 1;

int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n < 1) n = omp_get_num_procs();
    g_threads = n;
#else
    (void)n;
    g_threads = 1;
#endif
    return g_threads;
}

/* ---- vector helpers: operation order of common.cuh:92-202 ---------------------------------- */

static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline m33 mk33(const float* a) {
    m33 m = {{a[0], a[1], a[2]}, {a[3], a[4], a[5]}, {a[6], a[7], a[8]}};
    return m;
}
/* dot: (a.x*b.x + a.y*b.y) + a.z*b.z  (common.cuh:92-94) */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* matrix * vector = row dots (common.cuh:102-105) */
static inline v3 mulmv(m33 m, v3 v) { return mk(dot3(m.r0, v), dot3(m.r1, v), dot3(m.r2, v)); }
static inline v3 add3(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scale3(v3 a, float f) { return mk(a.x * f, a.y * f, a.z * f); }
static inline v3 div3(v3 a, float f) { return mk(a.x / f, a.y / f, a.z / f); }
static inline float norm3(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static inline m33 transp(m33 m) {
    m33 t = {{m.r0.x, m.r1.x, m.r2.x}, {m.r0.y, m.r1.y, m.r2.y}, {m.r0.z, m.r1.z, m.r2.z}};
    return t;
}

/* voxel-space coordinate of an object-frame point: p / voxelSize + (N - 1) / 2.f
 * (TSDF.cu:507-508, 525, 620, 680; int3 - int, int3 / float: common.cuh:144-145,185-189) */
static inline v3 to_voxel(v3 p, float voxelSize, const int res[3]) {
    v3 q = div3(p, voxelSize);
    v3 c = mk((float)(res[0] - 1) / 2.f, (float)(res[1] - 1) / 2.f, (float)(res[2] - 1) / 2.f);
    return add3(q, c);
}

/* interpolateTrilinear: TSDF.cuh:65-97.  Blend x, then y, then z, each (1 - f)*a + f*b.
 * `ch` interleaved channels, `c` the channel wanted. */
static inline float trilinear(const float* vol, int ch, int c, v3 idx, const int res[3]) {
    const int lx = (int)idx.x, ly = (int)idx.y, lz = (int)idx.z;
    const int hx = lx + 1, hy = ly + 1, hz = lz + 1;
    const float fx = idx.x - (float)lx, fy = idx.y - (float)ly, fz = idx.z - (float)lz;
    const size_t nx = (size_t)res[0], ny = (size_t)res[1];
#define VOX(zz, yy, xx) vol[(((size_t)(zz) * ny + (size_t)(yy)) * nx + (size_t)(xx)) * (size_t)ch + (size_t)c]
    float vs[8] = {VOX(lz, ly, lx), VOX(lz, ly, hx), VOX(lz, hy, lx), VOX(lz, hy, hx),
                   VOX(hz, ly, lx), VOX(hz, ly, hx), VOX(hz, hy, lx), VOX(hz, hy, hx)};
#undef VOX
    for (int i = 0; i < 4; ++i) vs[i] = (1 - fx) * vs[2 * i] + fx * vs[2 * i + 1];
    for (int i = 0; i < 2; ++i) vs[i] = (1 - fy) * vs[2 * i] + fy * vs[2 * i + 1];
    return (1 - fz) * vs[0] + fz * vs[1];
}

/* trilinear blend of weights as seen through the foreground mask (ObjTSDF.cpp:209-210) */
static inline float trilinear_w(const float* w, const uint8_t* fg, v3 idx, const int res[3]) {
    if (!fg) return trilinear(w, 1, 0, idx, res);
    const int lx = (int)idx.x, ly = (int)idx.y, lz = (int)idx.z;
    const int hx = lx + 1, hy = ly + 1, hz = lz + 1;
    const float fx = idx.x - (float)lx, fy = idx.y - (float)ly, fz = idx.z - (float)lz;
    const size_t nx = (size_t)res[0], ny = (size_t)res[1];
#define IDX(zz, yy, xx) (((size_t)(zz) * ny + (size_t)(yy)) * nx + (size_t)(xx))
#define WV(zz, yy, xx) (fg[IDX(zz, yy, xx)] ? w[IDX(zz, yy, xx)] : 0.f)
    float vs[8] = {WV(lz, ly, lx), WV(lz, ly, hx), WV(lz, hy, lx), WV(lz, hy, hx),
                   WV(hz, ly, lx), WV(hz, ly, hx), WV(hz, hy, lx), WV(hz, hy, hx)};
#undef WV
#undef IDX
    for (int i = 0; i < 4; ++i) vs[i] = (1 - fx) * vs[2 * i] + fx * vs[2 * i + 1];
    for (int i = 0; i < 2; ++i) vs[i] = (1 - fy) * vs[2 * i] + fy * vs[2 * i + 1];
    return (1 - fz) * vs[0] + fz * vs[1];
}

/* forward-difference gradient component at a voxel (kernel_computeTSDFGrads, TSDF.cu:429-448):
 * zero on the last index planes (the kernel returns there after setTo(0)) */
static inline float fwd_grad(const float* tsdf, int x, int y, int z, int c, const int res[3]) {
    if (x >= res[0] - 1 || y >= res[1] - 1 || z >= res[2] - 1) return 0.f;
    const size_t nx = (size_t)res[0], ny = (size_t)res[1];
    const size_t i = ((size_t)z * ny + (size_t)y) * nx + (size_t)x;
    const float t = tsdf[i];
    if (c == 0) return tsdf[i + 1] - t;
    if (c == 1) return tsdf[i + nx] - t;
    return tsdf[i + nx * ny] - t;
}

/* trilinear blend of the (virtual) gradient volume, one channel */
static inline float trilinear_grad(const float* tsdf, int c, v3 idx, const int res[3]) {
    const int lx = (int)idx.x, ly = (int)idx.y, lz = (int)idx.z;
    const int hx = lx + 1, hy = ly + 1, hz = lz + 1;
    const float fx = idx.x - (float)lx, fy = idx.y - (float)ly, fz = idx.z - (float)lz;
    float vs[8] = {fwd_grad(tsdf, lx, ly, lz, c, res), fwd_grad(tsdf, hx, ly, lz, c, res),
                   fwd_grad(tsdf, lx, hy, lz, c, res), fwd_grad(tsdf, hx, hy, lz, c, res),
                   fwd_grad(tsdf, lx, ly, hz, c, res), fwd_grad(tsdf, hx, ly, hz, c, res),
                   fwd_grad(tsdf, lx, hy, hz, c, res), fwd_grad(tsdf, hx, hy, hz, c, res)};
    for (int i = 0; i < 4; ++i) vs[i] = (1 - fx) * vs[2 * i] + fx * vs[2 * i + 1];
    for (int i = 0; i < 2; ++i) vs[i] = (1 - fy) * vs[2 * i] + fy * vs[2 * i + 1];
    return (1 - fz) * vs[0] + fz * vs[1];
}

/* ---- a1: computePoints (EMFusion.cu:29-61) -------------------------------------------------- */

void orc_computePoints(const float* depth, float* points, int w, int h, const float K[9]) {
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float d = depth[(size_t)y * w + x];
            float* p = points + ((size_t)y * w + x) * 3;
            p[0] = ((float)x - cx) * d / fx; /* EMFusion.cu:40-41: ((x - cx) * d) / fx */
            p[1] = ((float)y - cy) * d / fy;
            p[2] = d;
        }
}

/* ---- a7: updateTSDF (TSDF.cu:327-401) ------------------------------------------------------- */

void orc_updateTSDF(const float* depth, const float* assoc, int w, int h, float* tsdf,
                    float* weights, const float R_OC[9], const float t_OC[3], const float K[9],
                    const int res[3], float voxelSize, float truncdist, float maxWeight) {
    const m33 R = mk33(R_OC), Km = mk33(K);
    const v3 t = mk(t_OC[0], t_OC[1], t_OC[2]);
    const int nx = res[0], ny = res[1], nz = res[2];
    const long rows = (long)ny * nz;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long row = 0; row < rows; ++row) {
        const int y = (int)(row % ny), z = (int)(row / ny);
        float* trow = tsdf + (size_t)row * nx;
        float* wrow = weights + (size_t)row * nx;
        for (int x = 0; x < nx; ++x) {
            /* TSDF.cu:345-349 */
            const v3 pobj = mk(((float)x - (float)(nx - 1) / 2.f) * voxelSize,
                               ((float)y - (float)(ny - 1) / 2.f) * voxelSize,
                               ((float)z - (float)(nz - 1) / 2.f) * voxelSize);
            const v3 pcam = add3(mulmv(R, pobj), t);
            if (pcam.z <= 0.f) { /* TSDF.cu:351-356 */
                if (wrow[x] == 0) trow[x] = 0;
                continue;
            }
            const v3 proj = mulmv(Km, pcam);
            const int px = (int)lrintf(proj.x / proj.z); /* __float2int_rn, TSDF.cu:360-361 */
            const int py = (int)lrintf(proj.y / proj.z);
            if (px < 0 || px >= w || py < 0 || py >= h) continue;
            const float d = depth[(size_t)py * w + px];
            if (d <= 0.f) { /* TSDF.cu:367-372 */
                if (wrow[x] == 0) trow[x] = 0;
                continue;
            }
            /* TSDF.cu:374-379: lambda from the rounded pixel */
            const float lambda = norm3(mk(((float)px - K[2]) / K[0], ((float)py - K[5]) / K[4], 1.f));
            const float sdf = d - (1.f / lambda) * norm3(pcam);
            const float pw = wrow[x];
            if (sdf >= -truncdist) { /* TSDF.cu:382-397 */
                const float tv = copysignf(fminf(1.f, fabsf(sdf / truncdist)), sdf);
                const float pt = trow[x];
                const float aw = sdf < truncdist ? assoc[(size_t)py * w + px] : 1.f;
                if (pw + aw > 0) {
                    trow[x] = (pw * pt + aw * tv) / (pw + aw);
                    wrow[x] = fminf(pw + aw, maxWeight);
                }
            } else if (pw == 0) { /* TSDF.cu:398-400 */
                trow[x] = -1;
            }
        }
    }
}

/* ---- a8: updateGradients (TSDF.cpp:120-123, TSDF.cu:429-448) -------------------------------- */

void orc_computeTSDFGrads(const float* tsdf, float* grads, const int res[3]) {
    const int nx = res[0], ny = res[1], nz = res[2];
    const long rows = (long)ny * nz;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long row = 0; row < rows; ++row) {
        const int y = (int)(row % ny), z = (int)(row / ny);
        float* g = grads + (size_t)row * nx * 3;
        for (int x = 0; x < nx; ++x) {
            g[3 * x + 0] = fwd_grad(tsdf, x, y, z, 0, res);
            g[3 * x + 1] = fwd_grad(tsdf, x, y, z, 1, res);
            g[3 * x + 2] = fwd_grad(tsdf, x, y, z, 2, res);
        }
    }
}

/* ---- a10/a11: raycastTSDF (TSDF.cu:466-573, TSDF.cuh:31-63) --------------------------------- */

/* enterVolStep: TSDF.cuh:31-46 */
static inline float enter_step(v3 dir, v3 cam, v3 bb) {
    const float sx = ((dir.x > 0.f ? -bb.x : bb.x) - cam.x) / dir.x;
    const float sy = ((dir.y > 0.f ? -bb.y : bb.y) - cam.y) / dir.y;
    const float sz = ((dir.z > 0.f ? -bb.z : bb.z) - cam.z) / dir.z;
    return fmaxf(fmaxf(sx, sy), sz);
}
/* exitVolStep: TSDF.cuh:48-63 */
static inline float exit_step(v3 dir, v3 cam, v3 bb) {
    const float sx = ((dir.x > 0.f ? bb.x : -bb.x) - cam.x) / dir.x;
    const float sy = ((dir.y > 0.f ? bb.y : -bb.y) - cam.y) / dir.y;
    const float sz = ((dir.z > 0.f ? bb.z : -bb.z) - cam.z) / dir.z;
    return fminf(fminf(sx, sy), sz);
}

static inline int out_of(v3 v, float pad, const int res[3]) {
    return v.x < 0 || v.x + pad >= (float)res[0] || v.y < 0 || v.y + pad >= (float)res[1] ||
           v.z < 0 || v.z + pad >= (float)res[2];
}

void orc_raycastTSDF(const float* tsdfVol, const float* grads, const float* weights,
                     const uint8_t* fgmask, float* raylengths, float* vertices, float* normals,
                     uint8_t* mask, int w, int h, const float R_CO[9], const float t_CO[3],
                     const float K[9], const int res[3], float voxelSize, float truncdist,
                     uint32_t* steps) {
    const m33 R = mk33(R_CO);
    const m33 Rt = transp(R); /* rot_OC, TSDF.cu:561 */
    const v3 cam = mk(t_CO[0], t_CO[1], t_CO[2]);
    /* TSDF.cu:490: (volSize - 1) / 2 is INTEGER division (Q2) */
    const v3 bb = mk((float)((res[0] - 1) / 2) * voxelSize, (float)((res[1] - 1) / 2) * voxelSize,
                     (float)((res[2] - 1) / 2) * voxelSize);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t pi = (size_t)y * w + x;
            uint32_t nsteps = 0;
            if (steps) steps[pi] = 0;
            const v3 unproj = mk(((float)x - K[2]) / K[0], ((float)y - K[5]) / K[4], 1.f);
            const v3 ray = mulmv(R, unproj);
            const v3 dir = div3(ray, norm3(ray));
            float raylength = enter_step(dir, cam, bb);
            float maxRay = exit_step(dir, cam, bb);
            const float old = raylengths[pi]; /* TSDF.cu:496 */
            raylength += voxelSize;
            maxRay -= voxelSize;
            if (old != 0) maxRay = fminf(old, maxRay);
            if (raylength >= maxRay) continue; /* NaN compares false, like the reference */
            float raystep = truncdist;
            v3 v = to_voxel(add3(cam, scale3(dir, raylength)), voxelSize, res);
            while (out_of(v, 1.f, res) && raylength < maxRay) { /* TSDF.cu:509-514 */
                raylength += raystep;
                v = to_voxel(add3(cam, scale3(dir, raylength)), voxelSize, res);
            }
            /* Q4: the reference reads out of bounds here when the search ran out; the value can
             * only influence a loop that then never executes (raylength >= maxRay), so skip. */
            if (out_of(v, 1.f, res)) continue;
            float tsdf = trilinear(tsdfVol, 1, 0, v, res);
            if (fabsf(tsdf) < 1.f) raystep = voxelSize;
            if (fabsf(tsdf) < .8f) raystep = 0.5f * voxelSize;
            while ((raylength += raystep) <= maxRay) { /* TSDF.cu:523-572 */
                v = to_voxel(add3(cam, scale3(dir, raylength)), voxelSize, res);
                if (out_of(v, 2.f, res)) continue;
                ++nsteps;
                const float next = trilinear(tsdfVol, 1, 0, v, res);
                float wgt = trilinear_w(weights, fgmask, v, res);
                if (tsdf < 0 && next > 0 && wgt > 0.f) break;
                if (fabsf(next) < 1.f) raystep = voxelSize;
                if (fabsf(next) < .8f) raystep = 0.5f * voxelSize;
                if (tsdf > 0 && next < 0) {
                    const float tstar = raylength - raystep * tsdf / (next - tsdf); /* Q1 */
                    const v3 vs = to_voxel(add3(cam, scale3(dir, tstar)), voxelSize, res);
                    if (out_of(vs, 2.f, res)) continue; /* tsdf NOT advanced (TSDF.cu:547-550) */
                    wgt = trilinear_w(weights, fgmask, vs, res);
                    if (wgt > 0.f) {
                        v3 g;
                        if (grads) {
                            g = mk(trilinear(grads, 3, 0, vs, res), trilinear(grads, 3, 1, vs, res),
                                   trilinear(grads, 3, 2, vs, res));
                        } else {
                            g = mk(trilinear_grad(tsdfVol, 0, vs, res),
                                   trilinear_grad(tsdfVol, 1, vs, res),
                                   trilinear_grad(tsdfVol, 2, vs, res));
                        }
                        raylengths[pi] = tstar;
                        const v3 vert = mulmv(Rt, scale3(dir, tstar));
                        const v3 nrm = mulmv(Rt, div3(g, norm3(g))); /* Q16: 0/0 -> NaN */
                        vertices[3 * pi + 0] = vert.x;
                        vertices[3 * pi + 1] = vert.y;
                        vertices[3 * pi + 2] = vert.z;
                        normals[3 * pi + 0] = nrm.x;
                        normals[3 * pi + 1] = nrm.y;
                        normals[3 * pi + 2] = nrm.z;
                        mask[pi] = 1;
                        break;
                    }
                }
                tsdf = next;
            }
            if (steps) steps[pi] = nsteps;
        }
}

/* ---- a2: getVolumeVals (TSDF.cu:662-726) ---------------------------------------------------- */

void orc_getVolumeVals(const float* vol, int channels, const float* points, int w, int h,
                       const float R_CO[9], const float t_CO[3], const int res[3],
                       float voxelSize, float* vals) {
    const m33 R = mk33(R_CO);
    const v3 t = mk(t_CO[0], t_CO[1], t_CO[2]);
    memset(vals, 0, (size_t)w * h * channels * sizeof(float)); /* TSDF.cu:705 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t pi = (size_t)y * w + x;
            const v3 pc = mk(points[3 * pi], points[3 * pi + 1], points[3 * pi + 2]);
            if (pc.z <= 0) continue;
            const v3 p = add3(mulmv(R, pc), t);
            const v3 v = to_voxel(p, voxelSize, res);
            if (out_of(v, 1.f, res)) continue;
            for (int c = 0; c < channels; ++c)
                vals[pi * channels + c] = trilinear(vol, channels, c, v, res);
        }
}

/* ---- a13: updateFgBgProbs (ObjTSDF.cu:29-80) ------------------------------------------------ */

void orc_updateFgBgProbs(const uint8_t* mask, const uint8_t* occluded, int w, int h,
                         const float* tsdf, const float* weights, float* fgbg, const float Rm[9],
                         const float tv[3], const float K[9], const int res[3], float voxelSize) {
    const m33 R = mk33(Rm), Km = mk33(K);
    const v3 t = mk(tv[0], tv[1], tv[2]);
    const int nx = res[0], ny = res[1], nz = res[2];
    const long rows = (long)ny * nz;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long row = 0; row < rows; ++row) {
        const int y = (int)(row % ny), z = (int)(row / ny);
        for (int x = 0; x < nx; ++x) {
            const size_t i = (size_t)row * nx + x;
            if (fabsf(tsdf[i]) >= 1.f || weights[i] == 0.f) continue; /* ObjTSDF.cu:49-50 */
            const v3 pobj = mk(((float)x - (float)(nx - 1) / 2.f) * voxelSize,
                               ((float)y - (float)(ny - 1) / 2.f) * voxelSize,
                               ((float)z - (float)(nz - 1) / 2.f) * voxelSize);
            const v3 pcam = add3(mulmv(R, pobj), t);
            if (pcam.z <= 0.f) continue;
            const v3 proj = mulmv(Km, pcam);
            const int px = (int)lrintf(proj.x / proj.z);
            const int py = (int)lrintf(proj.y / proj.z);
            if (px < 0 || px >= w || py < 0 || py >= h) continue;
            const size_t pi = (size_t)py * w + px;
            if (!occluded[pi]) { /* ObjTSDF.cu:74-79; mask is read as bool */
                const int m = mask[pi] ? 1 : 0;
                fgbg[2 * i + 0] = fgbg[2 * i + 0] + (float)m;
                fgbg[2 * i + 1] = fgbg[2 * i + 1] + (float)(1 - m);
            }
        }
    }
}

/* ---- a14: computeFgProbs (ObjTSDF.cpp:218-226) ---------------------------------------------- */

void orc_computeFgProbs(const float* fgbg, float* fgProbs, uint8_t* fgVolMask, const int res[3]) {
    const long n = (long)res[0] * res[1] * res[2];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long i = 0; i < n; ++i) {
        const float fg = fgbg[2 * i], bg = fgbg[2 * i + 1];
        const float s = fg + bg;             /* cv::cuda::add            :220 */
        float p = (s != 0.f) ? fg / s : 0.f; /* cv::cuda::divide, x/0 := 0 (Q7) :222 */
        if (p != p) p = 0.f;                 /* compare(NE) + setTo(0)   :223-224 */
        fgProbs[i] = p;
        fgVolMask[i] = p > 0.5f ? 255 : 0;   /* compare(GT) -> 0/255     :225 */
    }
}

/* ---- a11 literal: ObjTSDF::raycast weight masking (ObjTSDF.cpp:209-210) --------------------- */

void orc_maskRaycastWeights(const float* weights, const uint8_t* fgVolMask, float* raycastWeights,
                            const int res[3]) {
    const long n = (long)res[0] * res[1] * res[2];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long i = 0; i < n; ++i) raycastWeights[i] = fgVolMask[i] ? weights[i] : 0.f;
}

/* ---- a3-a5: computeLaplace + computeAssociation (TSDF.cpp:125-156, ObjTSDF.cpp:181-201) ----- */

void orc_computeAssociation(const float* tsdf, const float* fgProbs, const float* points, int w,
                            int h, const float R_CO[9], const float t_CO[3], const int res[3],
                            float voxelSize, float truncdist, float assocSigma, float alpha,
                            float uniPrior, float* out) {
    const size_t n = (size_t)w * h;
    float* s = (float*)malloc(n * sizeof(float));
    float* f = fgProbs ? (float*)malloc(n * sizeof(float)) : NULL;
    orc_getVolumeVals(tsdf, 1, points, w, h, R_CO, t_CO, res, voxelSize, s);
    if (fgProbs) orc_getVolumeVals(fgProbs, 1, points, w, h, R_CO, t_CO, res, voxelSize, f);
    const float c1 = -truncdist / assocSigma;     /* TSDF.cpp:151, float on the host */
    const float c2 = 1.f / (2.f * assocSigma);    /* TSDF.cpp:154 */
    const float c3 = (1 - alpha) * uniPrior;      /* TSDF.cpp:133 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const float sv = s[i];
        const int invalid = (sv == 0.f);          /* compare(==0) -> associationMask, Q6 */
        float L = fabsf(sv);
        L = L * c1;
        L = expf(L);
        L = L * c2;
        if (f) L = L * f[i];                      /* ObjTSDF.cpp:192-193 */
        float wgt = L * alpha;
        wgt = wgt + c3;
        out[i] = invalid ? 0.f : wgt;             /* setTo(0, associationMask) */
    }
    free(s);
    free(f);
}

/* ---- a6: normalisation (EMFusion.cpp:653-665) ----------------------------------------------- */

void orc_normalizeAssociation(float* const* maps, int nmaps, int w, int h, float* norm) {
    const long n = (long)w * h;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long i = 0; i < n; ++i) {
        float s = maps[0][i];                       /* copyTo            :654 */
        for (int k = 1; k < nmaps; ++k) s = s + maps[k][i]; /* sequential adds :655-657 */
        norm[i] = s;
        for (int k = 0; k < nmaps; ++k)             /* divide, x/0 := 0  :659-665 */
            maps[k][i] = (s != 0.f) ? maps[k][i] / s : 0.f;
    }
}

/* ---- a12: raycast compositing (EMFusion.cpp:760-794) ---------------------------------------- */

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void orc_compositeRaycast(int nobj, const int* ids, const float* const* objRay,
                          const float* const* objVert, const float* const* objNorm,
                          const uint8_t* const* objSeg, const float* bgRay, const float* bgVert,
                          const float* bgNorm, const uint8_t* bgMask, float* ray, float* vert,
                          float* norm, uint8_t* seg, float* diff, uint8_t* noObj, int w, int h,
                          int boundary, int* visCounts) {
    const long n = (long)w * h;
    for (int k = 0; k < nobj; ++k) visCounts[k] = 0;
    for (long i = 0; i < n; ++i) {
        /* zeroed at the top of EMFusion::raycast (:727-733) */
        float r = 0.f;
        float vx = 0.f, vy = 0.f, vz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        uint8_t s = 0;
        for (int k = 0; k < nobj; ++k) { /* :760-771, list order, strict '<' (Q15) */
            const uint8_t m0 = (r <= 0) ? 255 : 0;
            const uint8_t m1 = (objRay[k][i] < r) ? 255 : 0;
            const uint8_t m = (uint8_t)(objSeg[k][i] & (m0 | m1));
            if (m) {
                r = objRay[k][i];
                vx = objVert[k][3 * i]; vy = objVert[k][3 * i + 1]; vz = objVert[k][3 * i + 2];
                nx = objNorm[k][3 * i]; ny = objNorm[k][3 * i + 1]; nz = objNorm[k][3 * i + 2];
                s = sat_u8(ids[k]);
            }
        }
        if (bgMask[i]) diff[i] = r - bgRay[i];  /* masked subtract, else stale (Q12) :773 */
        if (diff[i] > 0.05f) s = 0;             /* :774-775 */
        const uint8_t no = (s == 0) ? 255 : 0;  /* :776 */
        if (no) {                               /* :793-794 */
            vx = bgVert[3 * i]; vy = bgVert[3 * i + 1]; vz = bgVert[3 * i + 2];
            nx = bgNorm[3 * i]; ny = bgNorm[3 * i + 1]; nz = bgNorm[3 * i + 2];
        }
        ray[i] = r; /* composite raylength is NOT replaced by the background's */
        vert[3 * i] = vx; vert[3 * i + 1] = vy; vert[3 * i + 2] = vz;
        norm[3 * i] = nx; norm[3 * i + 1] = ny; norm[3 * i + 2] = nz;
        seg[i] = s;
        noObj[i] = no;
        const int px = (int)(i % w), py = (int)(i / w);
        if (s && px >= boundary && px < w - boundary && py >= boundary && py < h - boundary)
            for (int k = 0; k < nobj; ++k) /* :778-791 */
                if ((int)s == ids[k]) ++visCounts[k]; /* compare(seg, id): ids > 255 never match */
    }
}

/* ---- integrateMasks occlusion (EMFusion.cpp:897-900) ---------------------------------------- */

void orc_occludedMask(const uint8_t* objSeg, const uint8_t* seg, int id, uint8_t* occluded, int w,
                      int h) {
    const long n = (long)w * h;
    for (long i = 0; i < n; ++i) {
        const int own = ((int)seg[i] == id) ? 255 : 0;   /* compare(EQ) -> 0/255 */
        occluded[i] = sat_u8((int)objSeg[i] - own);      /* saturating u8 subtract */
    }
}

/* ==== f-1: weighted LM-ICP tracking ================================= */

/* cuda::TSDF::computePoseGradients / kernel_computePoseGradients (TSDF.cu:603-660).
 * grads6: (W*H) x 6, zero-filled first (TSDF.cu:655); per pixel with p_cam.z > 0 and the sample
 * inside [0, N-2): [ grad_tsdf (3), skew(p) * grad_tsdf (3) ], grad_tsdf = trilinear(tsdfGrads) /
 * voxelSize.  gradsVol: the N^3 x 3 gradient volume, or NULL = forward differences of `tsdf`
 * blended on the fly (the same values, see trilinear_grad). */
void orc_computePoseGradients(const float* tsdf, const float* gradsVol, const float* points, int w,
                              int h, const float R_CO[9], const float t_CO[3], const int res[3],
                              float voxelSize, float* grads6) {
    const m33 R = mk33(R_CO);
    const v3 t = mk(t_CO[0], t_CO[1], t_CO[2]);
    memset(grads6, 0, (size_t)w * h * 6 * sizeof(float));
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t pi = (size_t)y * w + x;
            const v3 pc = mk(points[3 * pi], points[3 * pi + 1], points[3 * pi + 2]);
            if (pc.z <= 0) continue;
            const v3 p = add3(mulmv(R, pc), t);
            const v3 v = to_voxel(p, voxelSize, res);
            if (out_of(v, 2.f, res)) continue;  /* TSDF.cu:623-626 */
            v3 g;
            if (gradsVol)
                g = mk(trilinear(gradsVol, 3, 0, v, res), trilinear(gradsVol, 3, 1, v, res),
                       trilinear(gradsVol, 3, 2, v, res));
            else
                g = mk(trilinear_grad(tsdf, 0, v, res), trilinear_grad(tsdf, 1, v, res),
                       trilinear_grad(tsdf, 2, v, res));
            g = div3(g, voxelSize);
            /* make_float33(0,-p.z,p.y, p.z,0,-p.x, -p.y,p.x,0) * grad_tsdf (TSDF.cu:630-632) */
            const v3 r0 = mk(0.f, -p.z, p.y), r1 = mk(p.z, 0.f, -p.x), r2 = mk(-p.y, p.x, 0.f);
            const v3 gr = mk(dot3(r0, g), dot3(r1, g), dot3(r2, g));
            float* o = grads6 + 6 * pi;
            o[0] = g.x; o[1] = g.y; o[2] = g.z;
            o[3] = gr.x; o[4] = gr.y; o[5] = gr.z;
        }
}

/* TSDF::computeHuberWeights + normalizeTSDFWeights + combineWeights (TSDF.cpp:218-252).
 * tsdfVals, intWeightsRaw: the two getVolumeVals lookups; assoc: association weights.
 * trackWeights = min(huber / |tsdfVals|, 1) with OpenCV's x / 0 := 0 (Q7);
 * intWeights   = min(raw, maxWeight) scaled by 1 / max (cv::cuda::normalize, NORM_INF, alpha 1:
 *                scale = norm > DBL_EPSILON ? 1 / norm : 0, applied in float), then
 *                trackWeights * intWeights, then * assoc (two separate multiplies). */
void orc_trackingWeights(const float* tsdfVals, const float* intWeightsRaw, const float* assoc,
                         int n, float huberThresh, float maxWeight, float* trackWeights,
                         float* intWeights) {
    float mx = 0.f;
    for (int i = 0; i < n; ++i) {
        const float a = fabsf(tsdfVals[i]);
        float tw = a != 0.f ? huberThresh / a : 0.f;
        trackWeights[i] = fminf(tw, 1.0f);
        const float c = fminf(intWeightsRaw[i], maxWeight);
        intWeights[i] = c;
        if (fabsf(c) > mx) mx = fabsf(c);
    }
    const float scale = (double)mx > 2.220446049250313e-16 ? (float)(1.0 / (double)mx) : 0.f;
    for (int i = 0; i < n; ++i) {
        float v = intWeights[i] * scale;
        v = trackWeights[i] * v;
        v = v * assoc[i];
        intWeights[i] = v;
    }
}

/* computeAb + multSingletonCol + column reduce (TSDF.cu:729-766, 821-853, TSDF.cpp:254-262,
 * 375-388): A = sum_i w_i g_i g_i^T (all 36 entries, row-major), b = sum_i w_i r_i g_i.
 * Each product is formed in float exactly as the kernels do -- (g_j * g_k) * w, (r * g_j) * w --
 * the SUM is accumulated in double and rounded once: cv::cuda::reduce's order is unspecified,
 * so this is the reference value any order approximates. */
void orc_reduceAb(const float* grads6, const float* tsdfVals, const float* intWeights, int n,
                  float A[36], float b[6]) {
    double As[36] = {0}, bs[6] = {0};
    for (int i = 0; i < n; ++i) {
        const float* g = grads6 + 6 * (size_t)i;
        const float w = intWeights[i], r = tsdfVals[i];
        for (int j = 0; j < 6; ++j) {
            for (int k = 0; k < 6; ++k) {
                const float a = g[j] * g[k];
                As[6 * j + k] += (double)(a * w);
            }
            const float bb = r * g[j];
            bs[j] += (double)(bb * w);
        }
    }
    for (int j = 0; j < 36; ++j) A[j] = (float)As[j];
    for (int j = 0; j < 6; ++j) b[j] = (float)bs[j];
}

/* TSDF::computeError (TSDF.cpp:390-394): sum(tsdfVals^2 * intWeights), double accumulation of
 * float products (cv::cuda::sum accumulates in double). */
double orc_trackingError(const float* tsdfVals, const float* intWeights, int n) {
    double e = 0;
    for (int i = 0; i < n; ++i) {
        float v = tsdfVals[i] * tsdfVals[i];
        v = v * intWeights[i];
        e += (double)v;
    }
    return e;
}

/* ==== f-2: depth pre-processing (EMFusion::preprocessDepth, EMFusion.cpp:294-305) ============== */

static inline int reflect101(int i, int n) { /* cv::BORDER_REFLECT_101, the BORDER_DEFAULT */
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

/* cv::cuda::bilateralFilter (third-party, OpenCV cudaimgproc bilateral_filter.cu -- not in the
 * reference tree; restated from its published algorithm, parity unpinned): window ksz x ksz
 * clipped to the disc of radius ksz / 2, weight = exp(space2 * (-0.5 / sigma_spatial^2) +
 * (value - centre)^2 * (-0.5 / sigma_color^2)), out = sum(weight * value) / sum(weight), borders
 * reflected.  Then the two patches of the reference: NaN -> 0, and 0 wherever the raw depth is 0. */
void orc_preprocessDepth(const float* raw, int w, int h, int ksz, float sigmaDepth,
                         float sigmaSpatial, float* out) {
    const int r = ksz / 2;
    const float r2 = (float)(r * r);
    const float ss = -0.5f / (sigmaSpatial * sigmaSpatial);
    const float sc = -0.5f / (sigmaDepth * sigmaDepth);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float center = raw[(size_t)y * w + x];
            float sum1 = 0.f, sum2 = 0.f;
            for (int cy = y - r; cy < y - r + ksz; ++cy)
                for (int cx = x - r; cx < x - r + ksz; ++cx) {
                    const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                    if (space2 > r2) continue;
                    const float v = raw[(size_t)reflect101(cy, h) * w + reflect101(cx, w)];
                    const float d = fabsf(v - center);
                    const float wgt = expf(space2 * ss + (d * d) * sc);
                    sum1 = sum1 + wgt * v;
                    sum2 = sum2 + wgt;
                }
            float o = sum1 / sum2;
            if (o != o) o = 0.f;         /* compare(depth, depth, NE) -> setTo(0) */
            if (center == 0.f) o = 0.f;  /* compare(depth_raw, 0, EQ) -> setTo(0) */
            out[(size_t)y * w + x] = o;
        }
}

/* ==== f-4: marching cubes (cuda::TSDF::marchingCubes, TSDF.cu:855-1152) ========================= */

/* the lookup table is shared with the device code (a data table, checked against the digest of the
 * reference's in tests/test_mc_tables.py); everything else below is restated independently */
#include "../emfusion_amd/csrc/mc_tables.h"

static inline int mc_edge_bits(int cls) { /* edgeTable[cls]: bit e set iff edge e changes sign */
    int m = 0;
    for (int e = 0; e < 12; ++e)
        m |= (((cls >> emf_mc_edge_corner[e][0]) ^ (cls >> emf_mc_edge_corner[e][1])) & 1) << e;
    return m;
}

/* corner i of the cube at (x, y, z): the reference indexes tsdf(y_ + dy + Ny * dz, x + dx) with
 * (dx, dy, dz) = 0:(0,0,0) 1:(1,0,0) 2:(1,0,1) 3:(0,0,1) 4:(0,1,0) 5:(1,1,0) 6:(1,1,1) 7:(0,1,1) */
static const int mc_dx[8] = {0, 1, 1, 0, 0, 1, 1, 0};
static const int mc_dy[8] = {0, 0, 0, 0, 1, 1, 1, 1};
static const int mc_dz[8] = {0, 0, 1, 1, 0, 0, 1, 1};

/* kernel_classifyCubes (TSDF.cu:872-907): 0 for masked-out cubes (the buffers are pre-zeroed) */
static int mc_classify(const float* tsdf, const float* weights, const uint8_t* fg, const int res[3],
                       int x, int y, int z) {
    const size_t sy = (size_t)res[0], sz = sy * res[1];
    const size_t base = (size_t)z * sz + (size_t)y * sy + x;
    for (int i = 0; i < 8; ++i) { /* mask = weights > 0 [& fgVolMask] */
        const size_t idx = base + ((i >> 2) & 1) * sz + ((i >> 1) & 1) * sy + (i & 1);
        if (!(weights[idx] > 0.f) || (fg && !fg[idx])) return 0;
    }
    int cls = 0;
    for (int i = 0; i < 8; ++i)
        cls |= (tsdf[base + mc_dx[i] + mc_dy[i] * sy + mc_dz[i] * sz] < 0.f) << i;
    return cls;
}

static v3 mc_interp(v3 p1, v3 p2, float v1, float v2) { /* vertexInterp, TSDF.cu:909-920 */
    if (fabs(v1) < 0.00001) return p1;
    if (fabs(v2) < 0.00001) return p2;
    if (fabs(v1 - v2) < 0.00001) return p1;
    const float mu = -v1 / (v2 - v1);
    return add3(p1, scale3(mk(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z), mu));
}

void orc_marchingCubesCount(const float* tsdf, const float* weights, const uint8_t* fg,
                            const int res[3], int* numVerts, int* numTris) {
    emf_mc_init();
    int nv = 0, nt = 0;
    for (int z = 0; z < res[2] - 1; ++z)
        for (int y = 0; y < res[1] - 1; ++y)
            for (int x = 0; x < res[0] - 1; ++x) {
                const int cls = mc_classify(tsdf, weights, fg, res, x, y, z);
                for (int m = mc_edge_bits(cls); m; m >>= 1) nv += m & 1; /* countVerts */
                for (int i = 0; emf_mc_tri_table[cls][i] != -1; i += 3) ++nt; /* countTris */
            }
    *numVerts = nv;
    *numTris = nt;
}

/* kernel_createTriangles (TSDF.cu:922-1108) over the cubes in buffer order (z, y, x), which is the
 * order the exclusive scans assign.  grads: N^3 x 3 gradient volume, or NULL for the values
 * kernel_computeTSDFGrads would have stored.  The normals are NOT normalised: `ns[i] /= norm(ns[i])`
 * and `normals[..] /= norm(..)` call operator/=(const float3&, float), which returns the quotient
 * and leaves its left side alone (common.cuh:170-173). */
void orc_marchingCubes(const float* tsdf, const float* grads, const float* weights,
                       const uint8_t* fg, const int res[3], float voxelSize, float* vertices,
                       float* normals, int* triangles) {
    emf_mc_init();
    const size_t sy = (size_t)res[0], sz = sy * res[1];
    int vertBase = 0, triBase = 0;
    for (int z = 0; z < res[2] - 1; ++z)
        for (int y = 0; y < res[1] - 1; ++y)
            for (int x = 0; x < res[0] - 1; ++x) {
                const int cls = mc_classify(tsdf, weights, fg, res, x, y, z);
                const int edges = mc_edge_bits(cls);
                if (edges == 0) continue;
                const size_t base = (size_t)z * sz + (size_t)y * sy + x;
                v3 ps[8], ns[8];
                float vals[8];
                for (int i = 0; i < 8; ++i) {
                    const int cx = x + mc_dx[i], cy = y + mc_dy[i], cz = z + mc_dz[i];
                    const size_t idx = base + mc_dx[i] + mc_dy[i] * sy + mc_dz[i] * sz;
                    ps[i] = mk(((float)cx - (float)(res[0] - 1) / 2.f) * voxelSize,
                               ((float)cy - (float)(res[1] - 1) / 2.f) * voxelSize,
                               ((float)cz - (float)(res[2] - 1) / 2.f) * voxelSize);
                    if (grads)
                        ns[i] = mk(grads[3 * idx], grads[3 * idx + 1], grads[3 * idx + 2]);
                    else if (cx < res[0] - 1 && cy < res[1] - 1 && cz < res[2] - 1)
                        ns[i] = mk(tsdf[idx + 1] - tsdf[idx], tsdf[idx + sy] - tsdf[idx],
                                   tsdf[idx + sz] - tsdf[idx]);
                    else
                        ns[i] = mk(0.f, 0.f, 0.f);
                    vals[i] = tsdf[idx];
                }
                int offsets[12], offset = 0;
                for (int e = 0; e < 12; ++e) {
                    if (!((edges >> e) & 1)) continue;
                    const int a = emf_mc_edge_corner[e][0], b = emf_mc_edge_corner[e][1];
                    const v3 p = mc_interp(ps[a], ps[b], vals[a], vals[b]);
                    const v3 n = mc_interp(ns[a], ns[b], vals[a], vals[b]);
                    float* vo = vertices + 3 * (size_t)(vertBase + offset);
                    float* no = normals + 3 * (size_t)(vertBase + offset);
                    vo[0] = p.x; vo[1] = p.y; vo[2] = p.z;
                    no[0] = n.x; no[1] = n.y; no[2] = n.z;
                    offsets[e] = offset++;
                }
                int j = 0;
                for (int i = 0; emf_mc_tri_table[cls][i] != -1; i += 3, j += 4) {
                    triangles[triBase + j] = 3;
                    triangles[triBase + j + 1] = vertBase + offsets[emf_mc_tri_table[cls][i]];
                    triangles[triBase + j + 2] = vertBase + offsets[emf_mc_tri_table[cls][i + 1]];
                    triangles[triBase + j + 3] = vertBase + offsets[emf_mc_tri_table[cls][i + 2]];
                }
                vertBase += offset;
                triBase += j;
            }
}

/* ==== f-4: kernel_renderPhong (EMFusion.cu:100-186) ============================================= */

static float orc_fastpow(float base, int exp) {
    float result = 1;
    while (exp) {
        if (exp & 1) result *= base;
        base *= base;
        exp >>= 1;
    }
    return result;
}

/* static_cast<uchar>(float): undefined outside [0, 256) in the reference; negative and NaN -> 0 */
static uint8_t orc_to_u8(float v) { return v >= 0.f ? (uint8_t)(v < 255.f ? (int)v : 255) : 0; }

/* image (H x W x 3) is fully written: 0 where the vertex is (0, 0, 0) (image.setTo(0) + early out) */
void orc_renderPhong(const float* points, const float* normals, const uint8_t* seg,
                     const uint8_t* colorMap, const float lightPos[3], int w, int h, uint8_t* image) {
    const float ka = 0.3f, kd = 0.5f, ks = 0.2f;
    const int alpha = 20;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const v3 p = mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        const v3 n = mk(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
        uint8_t* out = image + 3 * i;
        out[0] = out[1] = out[2] = 0;
        if (p.x == 0.f && p.y == 0.f && p.z == 0.f) continue;
        const uint8_t* c = colorMap + 3 * seg[i]; /* LookUpTable on the 3-channel copy of the labels */
        const v3 Rd = mk((float)c[0] / 255.f, (float)c[1] / 255.f, (float)c[2] / 255.f);
        v3 l = mk(lightPos[0] - p.x, lightPos[1] - p.y, lightPos[2] - p.z);
        l = div3(l, norm3(l));
        const v3 v = div3(mk(-p.x, -p.y, -p.z), norm3(p));
        const v3 two = scale3(n, 2.f * dot3(l, n));
        v3 r = mk(two.x - l.x, two.y - l.y, two.z - l.z);
        r = div3(r, norm3(r));
        const float diff = dot3(n, l), spec = orc_fastpow(dot3(r, v), alpha);
        const float I[3] = {ka * 1.f + (kd * Rd.x) * diff + (ks * 1.f) * spec,
                            ka * 1.f + (kd * Rd.y) * diff + (ks * 1.f) * spec,
                            ka * 1.f + (kd * Rd.z) * diff + (ks * 1.f) * spec};
        for (int k = 0; k < 3; ++k) out[k] = orc_to_u8(I[k] * 255.f);
    }
}
