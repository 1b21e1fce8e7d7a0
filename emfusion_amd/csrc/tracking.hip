// tracking.hip -- weighted Levenberg-Marquardt ICP on the TSDF.
//
// Reference: TSDF::prepareTracking ... computePoseUpdate (TSDF.cpp:170-344, 375-395), driven by
// EMFusion::performTracking (EMFusion.cpp:672-724); kernels computePoseGradients, getVolumeVals,
// computeAb, multSingletonCol (TSDF.cu:603-660, 662-726, 729-766, 821-853).  One LM iteration
// there is ~15 single-operator launches on 4 streams, a 44 MB (W*H x 36) `As` buffer that is
// written, re-read, scaled and column-reduced, two device->host copies with stream syncs, a 6x6
// solve on the host and -- on every trial step -- another lookup pass and a third sync.
//
// Here an iteration is ONE launch and no host round trip; all Levenberg-Marquardt state lives in
// device memory (emf_track_state_t), so a stage's iterations are enqueued back to back and the host
// reads the final pose once:
//   k_track_maxw   first launch of a stage only: the clamped integration-weight lookup per pixel and
//                  its image maximum (the NORM_INF of cv::cuda::normalize)
//   k_track_step   [prologue | per-pixel body] -- see the comment above the kernel.  One workgroup per CU.  Prologue,
//                  in every workgroup: the previous launch's partial sums added in a fixed order, the gain ratio and
//                  accept / reject of the pending trial step, damping, convergence tests, (A + mu I) x = b, the next
//                  trial pose exp(-x) * pose.  Body: residual of the trial pose under the current weights, the
//                  integration weights and their maximum there, and the pose gradient (6), Huber x normalised weight x
//                  association weight and the 21 + 6 + 1 sums A = sum w g g^T, b = sum w r g, sum r^2 w the NEXT
//                  iteration needs if the step is accepted -- all of a pixel's loads in one batch, reduced in
//                  registers (v_permlane swaps), through LDS, to one row of partials per 1216 pixels; `As` is never
//                  materialised.
// The four launches this replaced (sums | solve | trial error | verdict; round 1) cost 46 us per iteration, of which
// 20 were the 168 ds_bpermute shuffles of the 28 wave sums; rounds 2-3: 15-23 us in a 64-register kernel with two
// workgroups per CU; round 4: 14-15 us in-kernel (DESIGN.md section 10 has the table of what each change removed).
// Measured and dropped: finishing the scalar part in the LAST workgroup of the per-pixel kernel (ticket counter,
// __threadfence) -- on this multi-XCD part a device-scope release writes the XCD's L2 back, 1200 times per kernel:
// 51 -> 325 us per iteration; a persistent kernel with a grid barrier of relaxed agent-scope atomics: 8 us per
// exchange against 5 for a kernel boundary; look-ahead past rejected steps, slot masks, LDS-held points (round 4).
//
// Parity: per-pixel quantities follow the reference's operations one by one (the pose gradient
// is bit-identical to the oracle; tests/test_gpu_tracking.py).  Sums are formed in a different --
// but fixed, run-to-run deterministic -- order than cv::cuda::reduce's unspecified one, and the
// SE(3) exponential / QR re-orthonormalisation restate Sophus / Eigen (absent from the reference
// tree, versions unpinned): these agree with the oracle to float rounding, not bit for bit.
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <algorithm>

#include "device_core.hpp"

namespace emf_hip {
namespace {

#ifndef EMF_TRACK_BLOCK
#define EMF_TRACK_BLOCK 1024  // pixels per workgroup = per row of partial sums.  Round 4, resident grid, stage of the tracked bench
                              // (scripts/ab_track_block.sh): 512: 1.34 ms, 640: 1.04, 768: 1.03, 1024: 0.96 -- fewer, fatter workgroups win:
                              // every workgroup pays the prologue, and its cost grows with the number of rows
#endif
constexpr int kTrackBlock = EMF_TRACK_BLOCK;   // threads per workgroup
#ifndef EMF_TRACK_ROW_EXTRA
#define EMF_TRACK_ROW_EXTRA 192  // 640 x 480 pixels in rows of 1024 are 300 rows for 256 CUs: 44 CUs carry two workgroups and the
                                 // launch ends with them (20.4 us against 16.3).  Rows of 1024 + 192 pixels are 253: a workgroup
                                 // per CU, three waves take a second pixel
#endif
constexpr int kRowExtra = EMF_TRACK_ROW_EXTRA;          // pixels of a row beyond the workgroup's lanes (whole waves)
constexpr int kRowPixels = kTrackBlock + kRowExtra;     // pixels per row of partial sums
static_assert(kRowExtra % 64 == 0 && kRowExtra >= 0 && kRowExtra <= kTrackBlock, "whole waves take a second pixel");
constexpr int kCols = 30;          // partial-sum columns per workgroup: 21 (upper triangle of A) + 6 (b) + 1
                                   // (error), + the trial step's error + max |integration weight| at the
                                   // trial pose (k_track_step)

struct TrackFrame {
    const emf_model_t* models;
    emf_track_state_t* states;
    int nmodels;
    Img<const float> points;
    int w, h, nblocks;
    emf_track_params_t prm;
    char* scratch;          // per model: [w images 0, 1][iw images 0, 1][partials 2 x 30 x nblocks][shadow state]
    size_t scratchStride;   // bytes per model
    int launch;             // index of the launch within the trackIterate call (parity of the double buffers)
    int iterations;         // LM iterations the call asks for
    int rescale;            // an accepted step whose weight maximum moved rescales its sums (0: makes them anew, one launch more)
    uint32_t* watch;        // host memory (or null): [0] <- seq, [1 + m] <- model m is done (emf_hip_trackStep)
    emf_track_state_t* finalStates;  // host memory (or null): [m] <- model m's state, in front of watch[1 + m]
    uint32_t seq;
};

__device__ __forceinline__ float* scratch_base(const TrackFrame& f, int m) {
    return reinterpret_cast<float*>(f.scratch + f.scratchStride * m);
}
// two per-pixel weight images (Huber x normalised integration weight x association): [wSel] belongs to
// the current pose, the other one is filled at the trial pose
__device__ __forceinline__ float* scratch_w(const TrackFrame& f, int m, int sel) {
    return scratch_base(f, m) + static_cast<size_t>(f.w) * f.h * sel;
}
// two clamped integration-weight images, likewise ([iwSel])
__device__ __forceinline__ float* scratch_iw(const TrackFrame& f, int m, int sel) {
    return scratch_base(f, m) + static_cast<size_t>(f.w) * f.h * (2 + sel);
}
// two sets of partial sums, by launch parity
__device__ __forceinline__ float* scratch_partials(const TrackFrame& f, int m, int parity) {
    return scratch_base(f, m) + static_cast<size_t>(f.w) * f.h * 4 + static_cast<size_t>(f.nblocks) * kCols * parity;
}
// the state, by launch parity: [0] is the caller's array, [1] a shadow in the scratch
__device__ __forceinline__ emf_track_state_t* state_buf(const TrackFrame& f, int m, int parity) {
    if (parity == 0) return f.states + m;
    return reinterpret_cast<emf_track_state_t*>(scratch_base(f, m) + static_cast<size_t>(f.w) * f.h * 4 +
                                                static_cast<size_t>(f.nblocks) * kCols * 2);
}

__device__ __forceinline__ M33 state_R(const float* R) {
    return M33{{R[0], R[1], R[2]}, {R[3], R[4], R[5]}, {R[6], R[7], R[8]}};
}

// pixel of this lane of k_track_maxw: the image in runs of kTrackBlock pixels, blockIdx.y = model
__device__ __forceinline__ bool load_point(const TrackFrame& f, size_t& pix, V3& pc) {
    pix = static_cast<size_t>(blockIdx.x) * kTrackBlock + threadIdx.x;
    pc = v3(0.f, 0.f, 0.f);
    if (pix >= static_cast<size_t>(f.w) * f.h) return false;
    const int y = static_cast<int>(pix / f.w), x = static_cast<int>(pix - static_cast<size_t>(y) * f.w);
    const float* p = f.points.row(y) + 3 * x;
    pc = v3(p[0], p[1], p[2]);
    return true;
}

// getVolumeVals of one channel at one point (TSDF.cu:662-688): 0 outside [0, N - 1)
__device__ __forceinline__ float lookup1(const float* vol, const M33& R, const V3& t, const V3& pc,
                                         const I3& n, float voxelSize) {
    if (!(pc.z > 0)) return 0.f;
    const V3 v = to_voxel(mul(R, pc) + t, voxelSize, half_extent(n));
    if (outside(v, 1.f, n)) return 0.f;
    return trilinear1(vol, cell_of(v, n), n);
}

// kernel_computePoseGradients for one point (TSDF.cu:603-637): g[0..2] = trilinear(gradient) /
// voxelSize, g[3..5] = skew(p) * g[0..2]; zeros where the reference returns without writing
__device__ __forceinline__ void pose_gradient(const float* tsdf, const float* grads, const M33& R,
                                              const V3& t, const V3& pc, const I3& n,
                                              float voxelSize, float g[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = 0.f;
    if (!(pc.z > 0)) return;
    const V3 p = mul(R, pc) + t;
    const V3 v = to_voxel(p, voxelSize, half_extent(n));
    if (outside(v, 2.f, n)) return;
    RayVolume rv;
    rv.tsdf = tsdf;
    rv.grads = grads;
    rv.n = n;
    const V3 gt = gradient_at(rv, cell_of(v, n)) / voxelSize;
    // make_float33(0,-p.z,p.y, p.z,0,-p.x, -p.y,p.x,0) * grad_tsdf, evaluated as the full
    // matrix-vector product of the reference (products with the literal zeros included)
    const M33 S{{0.f, -p.z, p.y}, {p.z, 0.f, -p.x}, {-p.y, p.x, 0.f}};
    const V3 gr = mul(S, gt);
    g[0] = gt.x; g[1] = gt.y; g[2] = gt.z;
    g[3] = gr.x; g[4] = gr.y; g[5] = gr.z;
}

// The sums of 16 per-lane values over the wave, with 17 cross-lane moves instead of 16 x 6: at the
// level that pairs lane L with L ^ o each lane keeps the half of the values its bit `o` selects,
// adds the partner's copy of those and hands the other half over.  Every value still goes through
// the plain xor tree `v += shfl_xor(v, o)`, o = 32, 16, ..., 1 (a + b on one side is b + a on the
// other), so the results are bit-identical to it; afterwards lane L holds the total of value L >> 2
// in s[0].  (16 at a time, not all 29 sums of a pixel at once: that many live registers spill.)
__device__ __forceinline__ void wave_sum16(float (&s)[16], int lane) {
    // lane ^ 32 and lane ^ 16: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half
    // (the odd 16-lane rows) of the first register with the lower half (the even rows) of the second
    // in the VALU -- afterwards the two registers hold, lane by lane, the two addends
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[k]), __float_as_uint(s[k + 8]), false, false);
        s[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[k]), __float_as_uint(s[k + 4]), false, false);
        s[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int half = 2, o = 8; half >= 1; half >>= 1, o >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            float a = s[k], b = s[k + half];
            // (opaque: otherwise the two selects become one dynamically indexed read of s[], which
            // the backend expands into a compare chain over every register of the array)
            asm("" : "+v"(a), "+v"(b));
            const float keep = up ? b : a;
            const float send = up ? a : b;
            s[k] = keep + __shfl_xor(send, o);
        }
    }
    s[0] += __shfl_xor(s[0], 2);
    s[0] += __shfl_xor(s[0], 1);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- per-pixel kernels ---------------------------------------------------------------------------

__global__ __launch_bounds__(kTrackBlock) void k_track_maxw(const TrackFrame f) {
    const int m = blockIdx.y;
    emf_track_state_t& st = f.states[m];
    // only the first iteration of a stage: afterwards the weights of an accepted pose were already
    // looked up by k_track_step when that pose was the trial (TSDF.cpp:212-214, 234)
    if (st.converged || !st.firstIteration) return;
    const emf_model_t& md = f.models[m];
    size_t pix;
    V3 pc;
    float iw = 0.f;
    if (load_point(f, pix, pc)) {
        iw = lookup1(md.weights, state_R(st.R), v3(st.t[0], st.t[1], st.t[2]), pc,
                     I3{md.res[0], md.res[1], md.res[2]}, md.voxelSize);
        iw = fminf(iw, f.prm.maxWeight);  // cv::cuda::min(intWeights, maxTSDFWeight), TSDF.cpp:234
        scratch_iw(f, m, st.iwSel)[pix] = iw;
    }
    __shared__ float red[kTrackBlock / 64];
    const float wmx = wave_max(fabsf(iw));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = red[0];
        for (int i = 1; i < kTrackBlock / 64; ++i) mx = fmaxf(mx, red[i]);
        // non-negative floats order like their bit patterns; most workgroups find the maximum
        // (the weight cap, after a few frames) already there and skip the same-address atomic
        const unsigned bits = __float_as_uint(mx);
        if (bits > __atomic_load_n(&st.maxIwBits, __ATOMIC_RELAXED)) atomicMax(&st.maxIwBits, bits);
    }
}

// ---- SE(3) helpers (float; restating Sophus::SE3f::exp / log used at TSDF.cpp:300-313) -----------

struct Se3 {
    M33 R;
    V3 t;
};

__host__ __device__ inline M33 mat_mul(const M33& a, const M33& b) {
    const M33 bt = transpose(b);
    return M33{{dot(a.r0, bt.r0), dot(a.r0, bt.r1), dot(a.r0, bt.r2)},
               {dot(a.r1, bt.r0), dot(a.r1, bt.r1), dot(a.r1, bt.r2)},
               {dot(a.r2, bt.r0), dot(a.r2, bt.r1), dot(a.r2, bt.r2)}};
}

// exp of the twist (upsilon, omega) -- translation part first, as Sophus orders it
__host__ __device__ inline Se3 se3_exp(const float x[6]) {
    const V3 u = v3(x[0], x[1], x[2]), o = v3(x[3], x[4], x[5]);
    const float th2 = dot(o, o), th = sqrtf(th2);
    float A, B, C;
    if (th < 1e-4f) {
        A = 1.f - th2 / 6.f;
        B = 0.5f - th2 / 24.f;
        C = 1.f / 6.f - th2 / 120.f;
    } else {
        A = sinf(th) / th;
        B = (1.f - cosf(th)) / th2;
        C = (th - sinf(th)) / (th2 * th);
    }
    const M33 O{{0.f, -o.z, o.y}, {o.z, 0.f, -o.x}, {-o.y, o.x, 0.f}};
    const M33 O2 = mat_mul(O, O);
    auto comb = [](float a, const M33& X, float b, const M33& Y) {
        return M33{{(X.r0.x * a + Y.r0.x * b) + 1.f, X.r0.y * a + Y.r0.y * b, X.r0.z * a + Y.r0.z * b},
                   {X.r1.x * a + Y.r1.x * b, (X.r1.y * a + Y.r1.y * b) + 1.f, X.r1.z * a + Y.r1.z * b},
                   {X.r2.x * a + Y.r2.x * b, X.r2.y * a + Y.r2.y * b, (X.r2.z * a + Y.r2.z * b) + 1.f}};
    };
    Se3 e;
    e.R = comb(A, O, B, O2);          // I + A O + B O^2
    e.t = mul(comb(B, O, C, O2), u);  // V u,  V = I + B O + C O^2
    return e;
}

// |log(R, t)| -- only the norm enters the step-size test (TSDF.cpp:292-296)
__host__ __device__ inline float se3_log_norm(const M33& R, const V3& t) {
    const float tr = R.r0.x + R.r1.y + R.r2.z;
    const float c = fminf(1.f, fmaxf(-1.f, (tr - 1.f) * 0.5f));
    const float th = acosf(c);
    const V3 a = v3(R.r2.y - R.r1.z, R.r0.z - R.r2.x, R.r1.x - R.r0.y);  // 2 sin(th) * axis
    V3 o;
    if (th < 1e-4f) o = a * (0.5f * (1.f + th * th / 6.f));
    else o = a * (th / (2.f * sinf(th)));
    const float th2 = dot(o, o);
    // V^-1 = I - O / 2 + D O^2,  D = (1 - (th / 2) cot(th / 2)) / th^2
    float D;
    if (th2 < 1e-8f) D = 1.f / 12.f;
    else {
        const float thn = sqrtf(th2), hf = 0.5f * thn;
        D = (1.f - hf * cosf(hf) / sinf(hf)) / th2;
    }
    const M33 O{{0.f, -o.z, o.y}, {o.z, 0.f, -o.x}, {-o.y, o.x, 0.f}};
    const V3 Ot = mul(O, t), OOt = mul(O, Ot);
    const V3 u = t + Ot * (-0.5f) + OOt * D;
    return sqrtf(dot(u, u) + th2);
}

// Solve M x = b (6 x 6), float.  The reference calls cv::solve(DECOMP_LU): LU with partial pivoting.
// M = A + mu I with A = sum w g g^T (w >= 0) and mu > 0 is symmetric positive definite, for which
// Gaussian elimination WITHOUT pivoting is backward stable too -- and, fully unrolled, it runs in
// registers, whereas pivot swaps need dynamically indexed arrays that the compiler puts in scratch
// memory (a memory round trip per element: 22 us for this 6 x 6 system).  The two differ at the
// rounding level only.  Returns false and x = 0 for a (numerically) singular system, as cv::solve.
__host__ __device__ inline bool solve6(float M[6][6], float rhs[6], float x[6]) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        ok = ok && fabsf(M[c][c]) >= 1.1920929e-06f;  // FLT_EPSILON * 10, threshold of cv::hal::LU32f
        const float d = -1.f / M[c][c];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const float alpha = M[r][c] * d;
#pragma unroll
            for (int k = c + 1; k < 6; ++k) M[r][k] += alpha * M[c][k];
            rhs[r] += alpha * rhs[c];
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        float acc = rhs[r];
#pragma unroll
        for (int k = r + 1; k < 6; ++k) acc -= M[r][k] * x[k];
        x[r] = acc / M[r][r];
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = 0.f;
    }
    return ok;
}

// ---- the Levenberg-Marquardt step: one launch per iteration -------------------------------------
//
// A launch is [prologue | per-pixel body].  The prologue is the scalar LM logic (what used to be two
// one-workgroup kernels between the per-pixel ones): EVERY workgroup of a model runs it on its own
// LDS copy of the state -- same inputs, same code, same result everywhere -- so nobody waits for a
// grid-wide reduction other than through the kernel boundary: the partial sums the previous launch
// left are added (fixed order, double), the pending trial step is judged, the next one is solved
// for; workgroup 0 stores the state.  The state and the partial sums are double-buffered by launch
// parity (a fast workgroup's body must not overwrite what a slow one's prologue still reads).
// The body then evaluates the new trial pose: the error of the step under the current weights, the
// integration weights and their maximum there, and -- speculatively, they are the same gathers --
// the Hessian sums the NEXT iteration needs if this step is accepted.  An accepted step with an
// unchanged weight maximum (the normaliser of the weights) adopts them; otherwise the next launch
// re-makes them at the accepted pose (body 1) and the iteration after it costs one launch more.

constexpr int kBodyNone = 0, kBodyAccum = 1, kBodyTrial = 2;

// The scalar logic runs on a copy of the state in LDS: on the global-memory struct every access of
// the single working lane was a dependent ~0.5 us round trip; the state is copied cooperatively.
constexpr int kStateWords = sizeof(emf_track_state_t) / 4;
constexpr int kLogTrialWord = offsetof(emf_track_state_t, logTrial) / 4;
template <bool kSkipLogTrial = false>
__device__ __forceinline__ void state_copy(unsigned* dst, const unsigned* src, int tid, int nthreads) {
    for (int i = tid; i < kStateWords; i += nthreads)
        if (!kSkipLogTrial || i != kLogTrialWord) dst[i] = src[i];
}

// TSDF::reduceHessians' results (TSDF.cpp:264-279) from the 28 sums; the wave's lanes take an element each
__device__ __forceinline__ void adopt_sums(emf_track_state_t& st, const float* sums, int lane) {
    if (lane < 36) {
        const int j = lane / 6, k = lane - 6 * j, a = min(j, k), b = max(j, k);
        st.A[lane] = sums[6 * a - a * (a - 1) / 2 + (b - a)];  // (the upper triangle row by row)
    }
    if (lane < 6) st.b[lane] = sums[21 + lane];
    st.err = sums[27];
    st.needAccum = st.haveSpec = 0;
    st.checkB = 1;
}

// One wave: everything between two per-pixel passes.  Sets st.body (what this launch does per pixel) and st.pending
// (what the next launch will find in the partial sums).  Every lane runs the scalar logic on the same LDS values -- the
// instructions of one lane -- and the copies of arrays take a lane per element.
// st.logCur / st.logTrial: |log| of the current and of the trial pose (the step-size test needs the one of the
// pose the verdict leaves current); the trial's is made beside the per-pixel pass of the launch that made the pose.
__device__ void lm_advance(emf_track_state_t& st, const double* sums, const TrackFrame& f, int lane) {
    if (f.launch == 0) st.iterTarget = st.iterations + f.iterations;
    st.body = kBodyNone;
    if (st.converged) {
        st.pending = 0;
        return;
    }
    if (st.pending == kBodyAccum) {
        // (the pass has written the current pose's weight image with that pose's maximum -- or, the first pass of a stage
        // that looks the integration weights up itself, with the weight cap for the maximum it could not know yet: the
        // factor goes the way of an accepted step's, below)
        float k = 1.f;
        const uint32_t mxBits = __float_as_uint(static_cast<float>(sums[kCols - 1]));
        if (f.rescale && mxBits != st.maxIwBits) {
            const float mxGuess = __uint_as_float(st.maxIwBits), mxTrue = __uint_as_float(mxBits);
            k = static_cast<double>(mxTrue) > 2.220446049250313e-16
                    ? static_cast<float>(static_cast<double>(mxGuess) / static_cast<double>(mxTrue)) : 0.f;
            st.maxIwBits = mxBits;
        }
        if (lane < 28) st.spec[lane] = k == 1.f ? static_cast<float>(sums[lane]) : static_cast<float>(sums[lane]) * k;
        st.needAccum = 0;
        st.haveSpec = 1;
        st.wFac = k;
    } else if (st.pending == kBodyTrial) {
        // ---- computePoseUpdate, second half (TSDF.cpp:315-337) ----
        const float errNew = static_cast<float>(sums[28]);
        st.errNew = errNew;
        st.maxIwTrialBits = __float_as_uint(static_cast<float>(sums[29]));
        // gain = 0.5 * -x^T (mu * -x - b)  (TSDF.cpp:319)
        float gain = 0.f;
        for (int j = 0; j < 6; ++j) gain += -st.x[j] * (st.mu * -st.x[j] - st.b[j]);
        gain = 0.5f * gain;
        const float rho = (st.err - errNew) / gain;
        st.rho = rho;
        st.iterations += 1;
        if (rho > 0) {  // accept (TSDF.cpp:322-327)
            if (lane < 9) st.R[lane] = st.Rtrial[lane];
            if (lane < 3) st.t[lane] = st.ttrial[lane];
            st.logCur = st.logTrial;
            const float c = 2.f * rho - 1.f;
            const float rhoFac = 1.f - c * c * c;
            st.mu *= fmaxf(1.f / 3.f, rhoFac);
            st.nu = f.prm.nuInit;
            st.evaluateGradient = 1;
            st.accepted += 1;
            st.iwSel ^= 1;  // the trial pose's integration weights become the current ones
            const float mxCur = __uint_as_float(st.maxIwBits), mxNew = __uint_as_float(st.maxIwTrialBits);
            if (st.maxIwTrialBits == st.maxIwBits) {  // the body's weights were normalised correctly
                if (lane < 28) st.spec[lane] = static_cast<float>(sums[lane]);
                st.haveSpec = 1;
                st.wSel ^= 1;
                st.wFac = 1.f;
            } else if (f.rescale && static_cast<double>(mxCur) > 2.220446049250313e-16) {
                // ... by the previous pose's maximum: every weight, hence every sum, carries the factor mxNew / mxCur too
                // many.  The weights are products iw * (1 / max) * huber * assoc: taking the factor out afterwards differs
                // from making them anew by the rounding of two multiplications (2e-7 relative; the sums are compared with
                // the oracle's at 2e-5) and saves the launch that would make them anew -- one per accepted step for as
                // long as a model's integration weights have not reached their cap (its first maxTSDFWeight frames).
                const float k = static_cast<double>(mxNew) > 2.220446049250313e-16
                                    ? static_cast<float>(static_cast<double>(mxCur) / static_cast<double>(mxNew)) : 0.f;
                if (lane < 28) st.spec[lane] = static_cast<float>(sums[lane]) * k;
                st.haveSpec = 1;
                st.wSel ^= 1;
                st.wFac = k;
            } else {
                st.needAccum = 1;
            }
            st.maxIwBits = st.maxIwTrialBits;
        } else {  // reject (TSDF.cpp:328-336)
            st.mu *= st.nu;
            st.nu *= f.prm.nuInit;
            st.evaluateGradient = 0;
        }
        st.haveTrial = 0;
    }
    st.pending = 0;
    if (st.iterations >= st.iterTarget) return;
    if (st.needAccum) {
        // a stage's first pass looks the integration weights up itself (no k_track_maxw in front of it) and normalises by
        // the weight cap, which is their maximum once a volume has been seen maxTSDFWeight times
        if (st.firstIteration && f.rescale) st.maxIwBits = __float_as_uint(f.prm.maxWeight);
        st.body = st.pending = kBodyAccum;
        return;
    }
    // the iteration proper starts here: its A, b, err (TSDF.cpp:264-275) are the sums made ahead of it
    if (st.haveSpec) adopt_sums(st, st.spec, lane);
    if (st.checkB) {  // TSDF.cpp:276-278, on freshly reduced sums only
        st.checkB = 0;
        float maxB = 0.f;
        for (int j = 0; j < 6; ++j) maxB = fmaxf(maxB, fabsf(st.b[j]));
        if (maxB < f.prm.eps1) {
            st.converged = 1;
            return;
        }
    }
    // ---- computePoseUpdate, first half (TSDF.cpp:281-313) ----
    if (st.firstIteration) {
        float maxA = st.A[0];
        for (int j = 1; j < 6; ++j) maxA = fmaxf(maxA, st.A[7 * j]);
        st.mu = f.prm.tau * maxA;
        st.firstIteration = 0;
    }
    float M[6][6], rhs[6], x[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k < 6; ++k) M[j][k] = st.A[6 * j + k] + (j == k ? st.mu : 0.f);
        rhs[j] = st.b[j];
    }
    solve6(M, rhs, x);
    float nx = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        st.x[j] = x[j];
        nx += x[j] * x[j];
    }
    nx = sqrtf(nx);
    const M33 R = state_R(st.R);
    const V3 t = v3(st.t[0], st.t[1], st.t[2]);
    if (nx < f.prm.eps2 * (st.logCur + f.prm.eps2)) {  // se3_log_norm(R, t)
        st.converged = 1;
        return;
    }
    float mx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) mx[j] = -x[j];
    const Se3 inc = se3_exp(mx);  // pose_incr = exp(-x); rel_pose_CO = pose_incr * rel_pose_CO
    const M33 Rn = mat_mul(inc.R, R);
    const V3 tn = mul(inc.R, t) + inc.t;
    st.Rtrial[0] = Rn.r0.x; st.Rtrial[1] = Rn.r0.y; st.Rtrial[2] = Rn.r0.z;
    st.Rtrial[3] = Rn.r1.x; st.Rtrial[4] = Rn.r1.y; st.Rtrial[5] = Rn.r1.z;
    st.Rtrial[6] = Rn.r2.x; st.Rtrial[7] = Rn.r2.y; st.Rtrial[8] = Rn.r2.z;
    st.ttrial[0] = tn.x; st.ttrial[1] = tn.y; st.ttrial[2] = tn.z;
    st.haveTrial = 1;
    st.body = st.pending = kBodyTrial;
}

// ---- the per-pixel pass of k_track_step -----------------------------------------------------------
// What a pixel contributes: pose gradient (6), residual, combined weight, the trial step's error term, the clamped
// integration weight.
struct PixelTerms {
    float g[6], r, w, e, iw;
};
struct PixelPass {  // (wave-uniform)
    const float* tsdf;
    const float* weights;
    const float* grads;
    const float* assoc;
    I3 n;
    float voxelSize;
    M33 R;
    V3 t;
    bool trial;
    bool lookupIw;       // the pass looks the clamped integration weights up (a trial pose; a stage's first pass) or reads iwCur
    float scale, huberThresh, maxWeight;
    float wFac;          // factor on wCur (emf_track_state_t::wFac)
    const float* wCur;   // the weight image of the current pose (trial: read for the step's error)
    float* wOut;         // the weight image this pass fills
    const float* iwCur;  // clamped integration weights at the current pose (read when the pass is at that pose)
    float* iwOut;        // ... at the trial pose (filled by a trial pass)
};

// what a pass leaves per pixel: the combined weight, and at a trial pose the clamped integration weight
__device__ __forceinline__ void store_terms(const PixelPass& a, bool valid, size_t pix, const PixelTerms& o) {
    if (!valid) return;
    if (a.lookupIw) a.iwOut[pix] = o.iw;
    a.wOut[pix] = o.w;
}

// One pixel: the reference's operators one by one (computePoseGradients TSDF.cu:603-637, getVolumeVals 662-688,
// TSDF.cpp:212-262, 390-394), but with every load in front of every use: the range tests become selects, a pixel outside
// the volume (or the image) reads voxel 0 / pixel 0 and drops what it read, and the 20 tsdf values of the cell and of
// its forward differences (or the 24 of the gradient volume), the 8 weights and the three per-pixel images are requested
// in one batch -- one round trip instead of the three of "lookup, test, next lookup" (rounds 1-3, in 64 registers).
__device__ __forceinline__ PixelTerms pixel_terms(const PixelPass& a, bool valid, size_t pix, const V3& pc) {
    const bool zpos = valid && pc.z > 0;
    const V3 p = mul(a.R, pc) + a.t;
    const V3 v = to_voxel(p, a.voxelSize, half_extent(a.n));
    const bool in1 = zpos && !outside(v, 1.f, a.n);  // getVolumeVals' range (TSDF.cu:676-683)
    const bool in2 = zpos && !outside(v, 2.f, a.n);  // computePoseGradients' (TSDF.cu:617-624)
    const Cell c = cell_of(in1 ? v : v3(0.f, 0.f, 0.f), a.n);
    const size_t sy = static_cast<size_t>(a.n.x), sz = sy * a.n.y;
    const float* q = a.tsdf + c.base;
    // the cell's corners (z, y, x)
    const float c000 = q[0], c001 = q[1], c010 = q[sy], c011 = q[sy + 1];
    const float c100 = q[sz], c101 = q[sz + 1], c110 = q[sz + sy], c111 = q[sz + sy + 1];
    float gx, gy, gz;  // gradient_at: the blend of the eight corners' gradients
    if (a.grads) {     // (uniform) ... from the materialised volume (TSDF.cu:429-464)
        const float* pg = a.grads + 3 * c.base;
        float gv[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t off = 3 * ((k & 1) + ((k >> 1) & 1) * sy + (k >> 2) * sz);
            gv[k][0] = pg[off];
            gv[k][1] = pg[off + 1];
            gv[k][2] = pg[off + 2];
        }
        gx = blend8(gv[0][0], gv[1][0], gv[2][0], gv[3][0], gv[4][0], gv[5][0], gv[6][0], gv[7][0], c.fx, c.fy, c.fz);
        gy = blend8(gv[0][1], gv[1][1], gv[2][1], gv[3][1], gv[4][1], gv[5][1], gv[6][1], gv[7][1], c.fx, c.fy, c.fz);
        gz = blend8(gv[0][2], gv[1][2], gv[2][2], gv[3][2], gv[4][2], gv[5][2], gv[6][2], gv[7][2], c.fx, c.fy, c.fz);
    } else {           // ... or forward differences taken here: the voxels one further along each axis
        const size_t x2 = in2 ? 2 : 0, y2 = in2 ? 2 * sy : 0, z2 = in2 ? 2 * sz : 0;
        const float x00 = q[x2], x01 = q[sy + x2], x10 = q[sz + x2], x11 = q[sz + sy + x2];
        const float y00 = q[y2], y01 = q[y2 + 1], y10 = q[sz + y2], y11 = q[sz + y2 + 1];
        const float z00 = q[z2], z01 = q[z2 + 1], z10 = q[z2 + sy], z11 = q[z2 + sy + 1];
        gx = blend8(c001 - c000, x00 - c001, c011 - c010, x01 - c011, c101 - c100, x10 - c101, c111 - c110, x11 - c111,
                    c.fx, c.fy, c.fz);
        gy = blend8(c010 - c000, c011 - c001, y00 - c010, y01 - c011, c110 - c100, c111 - c101, y10 - c110, y11 - c111,
                    c.fx, c.fy, c.fz);
        gz = blend8(c100 - c000, c101 - c001, c110 - c010, c111 - c011, z00 - c100, z01 - c101, z10 - c110, z11 - c111,
                    c.fx, c.fy, c.fz);
    }
    const float* qw = a.weights + c.base;
    const float w000 = qw[0], w001 = qw[1], w010 = qw[sy], w011 = qw[sy + 1];
    const float w100 = qw[sz], w101 = qw[sz + 1], w110 = qw[sz + sy], w111 = qw[sz + sy + 1];
    const size_t pp = valid ? pix : 0;
    const float wCur = a.wCur[pp], iwCur = a.iwCur[pp], assoc = a.assoc[pp];
    PixelTerms o;
    // / voxelSize, and the rotational part (TSDF.cu:626-637)
    const V3 gt = v3(gx, gy, gz) / a.voxelSize;
    const M33 S{{0.f, -p.z, p.y}, {p.z, 0.f, -p.x}, {-p.y, p.x, 0.f}};
    const V3 gr = mul(S, gt);
    o.g[0] = in2 ? gt.x : 0.f; o.g[1] = in2 ? gt.y : 0.f; o.g[2] = in2 ? gt.z : 0.f;
    o.g[3] = in2 ? gr.x : 0.f; o.g[4] = in2 ? gr.y : 0.f; o.g[5] = in2 ? gr.z : 0.f;
    const float rIn = blend8(c000, c001, c010, c011, c100, c101, c110, c111, c.fx, c.fy, c.fz);
    o.r = in1 ? rIn : 0.f;
    const float iwIn = blend8(w000, w001, w010, w011, w100, w101, w110, w111, c.fx, c.fy, c.fz);
    o.e = a.trial ? (o.r * o.r) * (wCur * a.wFac) : 0.f;  // (wFac = 1 leaves the weight as it is)
    o.iw = a.lookupIw ? fminf(in1 ? iwIn : 0.f, a.maxWeight) : iwCur;
    const float ab = fabsf(o.r);
    float tw = ab != 0.f ? a.huberThresh / ab : 0.f;
    tw = fminf(tw, 1.0f);
    float w = o.iw * a.scale;
    w = tw * w;
    w = w * assoc;
    o.w = w;
    if (!valid) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o.g[k] = 0.f;
        o.r = o.w = o.e = o.iw = 0.f;
    }
    return o;
}

// A workgroup barrier for LDS contents alone: __syncthreads() also waits for every global load and store the wave has
// in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// progress report to the host (hints only: see emf_hip_trackStep); system scope, so that the stores
// go to the host's memory while the kernel runs
// (one wave: a model that is done also sends its state, before the word that says so -- the host needs no copy command
// and no wait for the stream to read a stage's result)
__device__ __forceinline__ void report(const TrackFrame& f, int m, const emf_track_state_t& st, int lane, bool sendState = true) {
    if (!f.watch) return;
    const uint32_t done = st.converged ? 1u : (st.pending == 0 && st.iterations >= st.iterTarget ? 2u : 0u);
    if (done != 0u && f.finalStates && sendState) {
        const unsigned* const src = reinterpret_cast<const unsigned*>(&st);
        unsigned* const dst = reinterpret_cast<unsigned*>(f.finalStates + m);
        for (int i = lane; i < kStateWords; i += 64) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (system scope: the state is out before the word)
    }
    if (lane == 0) {
        // (the upper half of seq travels with the word: a host that does not wait for a stage's last launches tells their
        // words from the next stage's by it)
        __hip_atomic_store(f.watch + 1 + m, (f.seq & 0xffff0000u) | done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (m == 0) __hip_atomic_store(f.watch, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup per CU (4 waves per SIMD, 128 registers): the prologue is paid once per CU -- two workgroups sharing
// a CU slow each other's scalar part by 40 % -- and the per-pixel pass has the registers to request all of a pixel's
// voxels at once, for two pixels side by side.  A workgroup takes every gridDim.x-th row of kRowPixels pixels, up to
// kMaxRows of them per pass: per wave a row is one slot (two for the three waves that take a second pixel of it); the
// points of all slots are tested first (one batch of loads), then the slots with a live pixel are worked off two at a
// time.  An object covers a few of its rows: its stage used to walk them one after the other, a barrier each.
constexpr int kMaxRows = 4;
__global__ __launch_bounds__(kTrackBlock, 4) void k_track_step(const TrackFrame f) {
    constexpr int kWaves = kTrackBlock / 64;
    __shared__ double sums[kCols];
    __shared__ emf_track_state_t st;
    __shared__ float red[kMaxRows][kWaves][32], redMax[kMaxRows][kWaves];
    const int m = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef EMF_TRACK_TRACE  // timing probe: 10 ns stamps of workgroup (EMF_TRACK_TRACE) of model 0, per launch
    long long* stamps = reinterpret_cast<long long*>(state_buf(f, 0, 1) + 1) + 8 * f.launch;
    const bool tracer = blockIdx.x == EMF_TRACK_TRACE && m == 0 && threadIdx.x == 0 && f.launch < 24;
#define STAMP(i) do { if (tracer) stamps[i] = wall_clock64(); \
        if ((i == 0 || i == 6) && m == 0 && threadIdx.x == 0 && f.launch == 5) /* every workgroup of launch 5 */ \
            (reinterpret_cast<long long*>(state_buf(f, 0, 1) + 1) + 8 * 24)[2 * blockIdx.x + (i ? 1 : 0)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    STAMP(0);
    const emf_track_state_t* in = state_buf(f, m, f.launch & 1);
    // this wave's columns of the previous launch's partial sums: wave, wave + kWaves, ... (a sum each; the last
    // column is the maximum)
    constexpr int kColsPerWave = (kCols + kWaves - 1) / kWaves;
    const float* const prev = scratch_partials(f, m, (f.launch + 1) & 1);
    // The first batch of this wave's partial sums is requested BEFORE the state says whether they are wanted: the state and
    // the sums, both written on other XCDs by the previous launch, are two misses all the way to memory -- side by side
    // instead of in a row.  (The compiler sinks a load behind the branch that alone uses it: the other side "uses" the
    // values too.  Volatile loads are waited for one by one: 4.8 instead of 3.2 us.)
    float v0[kColsPerWave][8];
#pragma unroll
    for (int q = 0; q < kColsPerWave; ++q) {
        const int c = wave + q * kWaves;
        const float* const col = prev + static_cast<size_t>(c < kCols ? c : 0) * f.nblocks;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v0[q][j] = 0.f;
            if (64 * j < f.nblocks) v0[q][j] = col[min(lane + 64 * j, f.nblocks - 1)];  // (uniform: 253 rows are 4 of the 8)
        }
    }
    // (likewise the state's own words: one per lane, in flight with the flags the branch below reads)
    static_assert(kStateWords <= kTrackBlock, "a word per lane");
    const unsigned stWord = threadIdx.x < kStateWords ? reinterpret_cast<const unsigned*>(in)[threadIdx.x] : 0u;
    // nothing left to do for this model in this call (lm_advance would find the same): pass the state on
    if (in->converged || (f.launch > 0 && in->pending == 0 && in->iterations >= in->iterTarget)) {
        if (blockIdx.x == 0) {
            if (threadIdx.x < kStateWords) reinterpret_cast<unsigned*>(state_buf(f, m, (f.launch + 1) & 1))[threadIdx.x] = stWord;
            // (the state went out when the model became done; a call that FINDS it done sends it with its first launch)
            if (wave == 0) report(f, m, *in, lane, f.launch == 0);
        }
        asm volatile("" ::"v"(stWord));
#pragma unroll
        for (int q = 0; q < kColsPerWave; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(v0[q][j]));  // (see above)
        return;
    }
    if (threadIdx.x < kStateWords) reinterpret_cast<unsigned*>(&st)[threadIdx.x] = stWord;
    const emf_model_t& md = f.models[m];
    const I3 n{md.res[0], md.res[1], md.res[2]};
    // the points of the workgroup's first row: fetched under the prologue
    const size_t npx = static_cast<size_t>(f.w) * f.h;
    // A row's kRowExtra pixels beyond the workgroup's lanes are the second pixels of its first kRowExtra / 64 waves.
    // (Measured and dropped: other waves for every row of a workgroup, so that not the same three carry all second slots --
    // 18.1 instead of 17.3 us per launch, for the camera's single row as for four objects' four.)
    const auto extra_of = [&](unsigned) { return wave < kRowExtra / 64 ? wave : -1; };
    // the pixel of slot (row, sec) and its point; the point is loaded whether the pixel exists or not (pixel 0 instead),
    // so that the loads of several slots go out together
    const bool dense = f.points.pitch == static_cast<size_t>(f.w) * 12;  // (no row padding: the pixel index is the address)
    const auto slot_pixel = [&](unsigned row, int sec) {
        return static_cast<size_t>(row) * kRowPixels + (sec ? kTrackBlock + 64 * extra_of(row) + lane : threadIdx.x);
    };
    const auto slot_point = [&](unsigned row, int sec, size_t& pix, V3& pc) {
        pix = slot_pixel(row, sec);
        const bool ok = pix < npx;
        // (32-bit: a 64-bit quotient by a run-time divisor is ~150 instructions, and a wave takes up to eight of these
        // per launch; fill_frame admits fewer than 2^31 pixels)
        const unsigned pp = ok ? static_cast<unsigned>(pix) : 0u;
        const float* p;
        if (dense) {
            p = f.points.data + 3 * static_cast<size_t>(pp);
        } else {
            const unsigned uw = static_cast<unsigned>(f.w);
            const int y = static_cast<int>(pp / uw), x = static_cast<int>(pp - static_cast<unsigned>(y) * uw);
            p = f.points.row(y) + 3 * x;
        }
        pc = v3(p[0], p[1], p[2]);
        return ok;
    };
    const bool second = extra_of(blockIdx.x) >= 0;  // (of the workgroup's first row)
    size_t pixA, pixB = 0;
    V3 pcA, pcB = v3(0.f, 0.f, 0.f);
    const bool okA = slot_point(blockIdx.x, 0, pixA, pcA);
    const bool okB = second && slot_point(blockIdx.x, 1, pixB, pcB);
    // ---- prologue ----
    if (in->pending != 0) {  // (uniform: read from global memory, not from the copy in flight)
        // lane-strided partial sums in double, then a fixed xor tree -- the same order on every run.
        // The rows of a lane are loaded in batches of 8 before any of them is added: a plain
        // `acc += p[i]` loop is not pipelined by the compiler (the double add is a dependency chain)
        // and exposes one memory latency per element.  The maximum column likewise.
        double acc[kColsPerWave];
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < kColsPerWave; ++q) acc[q] = 0.0;
        for (int i0 = lane; i0 < f.nblocks; i0 += 64 * 8) {
            float v[kColsPerWave][8];
#pragma unroll
            for (int q = 0; q < kColsPerWave; ++q) {
                const int c = wave + q * kWaves;
                const float* const col = prev + static_cast<size_t>(c < kCols ? c : 0) * f.nblocks;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 64 * j;
                    v[q][j] = i >= f.nblocks ? 0.f : (i0 == lane ? v0[q][j] : col[i]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 - lane + 64 * j < f.nblocks) {  // (uniform; the rows beyond the last would add +0.0: the same bits)
#pragma unroll
                    for (int q = 0; q < kColsPerWave; ++q) {
                        acc[q] += static_cast<double>(v[q][j]);
                        if (wave + q * kWaves == kCols - 1) mx = fmaxf(mx, v[q][j]);
                    }
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < kColsPerWave; ++q) acc[q] += __shfl_xor(acc[q], o);
        mx = wave_max(mx);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < kColsPerWave; ++q) {
                const int c = wave + q * kWaves;
                if (c < kCols - 1) sums[c] = acc[q];
                else if (c == kCols - 1) sums[c] = static_cast<double>(mx);
            }
        }
    }
#ifdef EMF_TRACK_TRACE  // every wave of the traced workgroup, launch 6: arrival at the barrier behind the column sums
    if (blockIdx.x == EMF_TRACK_TRACE && m == 0 && lane == 0 && f.launch == 6)
        (reinterpret_cast<long long*>(state_buf(f, 0, 1) + 1) + 8 * 24 + 2 * f.nblocks)[wave] = wall_clock64();
#endif
    STAMP(1);
    lds_barrier();  // (not __syncthreads(): that would also wait for the points requested above -- a miss to memory, 1 us)
    STAMP(2);
    if (wave == 0) lm_advance(st, sums, f, lane);
    STAMP(3);
    lds_barrier();
    if (blockIdx.x == 0 && wave == 0) report(f, m, st, lane);
    const int body = __builtin_amdgcn_readfirstlane(st.body);
    if (blockIdx.x == 0) {
        emf_track_state_t* const out = state_buf(f, m, (f.launch + 1) & 1);
        state_copy<true>(reinterpret_cast<unsigned*>(out), reinterpret_cast<const unsigned*>(&st), threadIdx.x, kTrackBlock);
        // the new trial pose's |log| (the step-size test needs it once the pose is the current one): one lane of one
        // workgroup, beside the per-pixel pass -- not two lanes of every workgroup in front of the solve (the barrier
        // behind the column sums waited 1.8 us for them; round 4)
        if (threadIdx.x == kTrackBlock - 1)
            out->logTrial = body == kBodyTrial
                                ? se3_log_norm(state_R(st.Rtrial), v3(st.ttrial[0], st.ttrial[1], st.ttrial[2]))
                                : st.logTrial;
    }
    // (what comes out of the LDS copy of the state is the same in every lane: kept in scalar registers)
    const auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    STAMP(4);
    if (body == kBodyNone) return;
    // ---- body ----
    const bool trial = body == kBodyTrial;
    const float* Rp = trial ? st.Rtrial : st.R;
    const float* tp = trial ? st.ttrial : st.t;
    PixelPass a;
    a.tsdf = md.tsdf;
    a.weights = md.weights;
    a.grads = md.grads;
    a.assoc = md.assoc;
    a.n = n;
    a.voxelSize = md.voxelSize;
    a.R = M33{{uni(Rp[0]), uni(Rp[1]), uni(Rp[2])}, {uni(Rp[3]), uni(Rp[4]), uni(Rp[5])}, {uni(Rp[6]), uni(Rp[7]), uni(Rp[8])}};
    a.t = v3(uni(tp[0]), uni(tp[1]), uni(tp[2]));
    a.trial = trial;
    a.lookupIw = trial || (f.rescale != 0 && __builtin_amdgcn_readfirstlane(st.firstIteration) != 0);
    // cv::cuda::normalize(NORM_INF, alpha = 1): scale = norm > DBL_EPSILON ? 1 / norm : 0
    // (trial: the current pose's maximum, assumed to hold at the trial pose as well -- checked by
    // the next prologue)
    const float mx = __uint_as_float(__builtin_amdgcn_readfirstlane(st.maxIwBits));
    a.scale = uni(static_cast<double>(mx) > 2.220446049250313e-16 ? static_cast<float>(1.0 / static_cast<double>(mx)) : 0.f);
    a.wFac = uni(st.wFac);
    a.huberThresh = f.prm.huberThresh;
    a.maxWeight = f.prm.maxWeight;
    const int iwSel = __builtin_amdgcn_readfirstlane(st.iwSel), wSel = __builtin_amdgcn_readfirstlane(st.wSel);
    a.wCur = scratch_w(f, m, wSel);
    a.wOut = scratch_w(f, m, trial ? 1 - wSel : wSel);
    a.iwCur = scratch_iw(f, m, iwSel);
    a.iwOut = scratch_iw(f, m, trial ? 1 - iwSel : iwSel);
    float* const mine = scratch_partials(f, m, f.launch & 1);
    // A pixel's 28 products As = (g_j * g_k) * w, bs = (r * g_j) * w, r^2 w (computeAb / multSingletonCol; column
    // order: the upper triangle of A row by row (21), b (6), err) and the trial step's error, summed over the wave
    // into the wave's line of its row in red[].  A wave with two pixels in the row adds the second one's products to the
    // first one's lane by lane and sums once (a dead partner would add exact zeros: the sums do not depend on which
    // slots were skipped).
    const auto accumulate = [&](const PixelTerms& o, const PixelTerms* o2, int j) {
        float* const line = red[j][wave];
        {
            float s[16];
            int q = 0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
#pragma unroll
                for (int k = jj; k < 6; ++k)
                    if (6 * jj - jj * (jj - 1) / 2 + (k - jj) < 16) {
                        s[q] = (o.g[jj] * o.g[k]) * o.w;
                        if (o2) s[q] += (o2->g[jj] * o2->g[k]) * o2->w;
                        ++q;
                    }
            wave_sum16(s, lane);
            if (!(lane & 3)) line[lane >> 2] = s[0];
        }
        {
            float s[16];
            int q = 0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
#pragma unroll
                for (int k = jj; k < 6; ++k)
                    if (6 * jj - jj * (jj - 1) / 2 + (k - jj) >= 16) {
                        s[q] = (o.g[jj] * o.g[k]) * o.w;
                        if (o2) s[q] += (o2->g[jj] * o2->g[k]) * o2->w;
                        ++q;
                    }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                s[q] = (o.r * o.g[jj]) * o.w;
                if (o2) s[q] += (o2->r * o2->g[jj]) * o2->w;
                ++q;
            }
            s[q] = (o.r * o.r) * o.w;  // computeError: sqr, multiply, sum (TSDF.cpp:390-394)
            if (o2) s[q] += (o2->r * o2->r) * o2->w;
            ++q;
            s[q] = o.e;
            if (o2) s[q] += o2->e;
            ++q;
            s[13] = s[14] = s[15] = 0.f;
            wave_sum16(s, lane);
            if (!(lane & 3) && 16 + (lane >> 2) < kCols - 1) line[16 + (lane >> 2)] = s[0];
        }
        const float wmx = wave_max(o2 ? fmaxf(fabsf(o.iw), fabsf(o2->iw)) : fabsf(o.iw));
        if (lane == 0) redMax[j][wave] = wmx;
    };
    const auto terms = [&](bool ok, size_t pix, const V3& pc) { return pixel_terms(a, ok, pix, pc); };
    // May the pixel lie in the volume's interpolation range?  The wave-skip below only needs "certainly not": the
    // voxel coordinate through a reciprocal instead of the three divisions of to_voxel (a third of a slot's test), with a
    // margin a hundred times the two forms' difference (< 2.4e-7 relative: 5e-4 voxels in the largest volume).  A wave
    // that passes with no pixel inside after all takes the arithmetic and gets the same zeros out of it.
    const float rcpVoxel = 1.f / a.voxelSize;
    const V3 halfExt = half_extent(n);
    const float hiX = static_cast<float>(n.x) + 0.05f, hiY = static_cast<float>(n.y) + 0.05f, hiZ = static_cast<float>(n.z) + 0.05f;
    const auto alive_at = [&](bool ok, const V3& pc) {
        const V3 p = mul(a.R, pc) + a.t;
        const float vx = p.x * rcpVoxel + halfExt.x, vy = p.y * rcpVoxel + halfExt.y, vz = p.z * rcpVoxel + halfExt.z;
        const bool out = vx < -0.05f || vx + 1.f >= hiX || vy < -0.05f || vy + 1.f >= hiY || vz < -0.05f || vz + 1.f >= hiZ;
        return ok && pc.z > 0 && !out;
    };
    const auto dead_slot = [&](bool ok, size_t pix) {  // a wave without a live pixel: zeros to the images
        if (!ok) return;
        if (a.lookupIw) a.iwOut[pix] = 0.f;
        a.wOut[pix] = 0.f;
    };
    // The image in rows of kRowPixels pixels, one row of partial sums each: the sums depend neither on the grid nor
    // on how the rows are grouped into passes.
    for (unsigned base = blockIdx.x; base < static_cast<unsigned>(f.nblocks); base += gridDim.x * kMaxRows) {
        if (base != blockIdx.x) lds_barrier();  // red[] of the previous pass has been read
        // A pixel whose point is invalid or falls outside the volume's interpolation range contributes exact zeros
        // to everything (value, gradient, weights: TSDF.cu:617-624, 676-683): a wave of such pixels -- most of the
        // image, for an object -- stores its zeros and skips the arithmetic.
        unsigned live = 0u;  // bit 2 j + sec: the wave has a live pixel in slot sec of the pass's j-th row
        int nrows = 1;
        const bool single = static_cast<unsigned>(f.nblocks) <= gridDim.x;  // a row per workgroup: its points are here already
        if (single) {
            if (__ballot(alive_at(okA, pcA)) != 0ull) live |= 1u;
            else dead_slot(okA, pixA);
            if (second) {
                if (__ballot(alive_at(okB, pcB)) != 0ull) live |= 2u;
                else dead_slot(okB, pixB);
            }
        } else {
            // every slot's point first (one batch of loads), then the tests and the dead slots' zeros
            V3 pcs[2 * kMaxRows];
#pragma unroll
            for (int j = 0; j < kMaxRows; ++j)
#pragma unroll
                for (int sec = 0; sec < 2; ++sec) {
                    size_t pix;
                    pcs[2 * j + sec] = v3(0.f, 0.f, 0.f);
                    if (base + j * gridDim.x < static_cast<unsigned>(f.nblocks) && (sec == 0 || extra_of(base + j * gridDim.x) >= 0))
                        slot_point(base + j * gridDim.x, sec, pix, pcs[2 * j + sec]);
                }
#pragma unroll
            for (int j = 0; j < kMaxRows; ++j) {
                const unsigned row = base + j * gridDim.x;
                if (row < static_cast<unsigned>(f.nblocks)) {
                    nrows = j + 1;
#pragma unroll
                    for (int sec = 0; sec < 2; ++sec)
                        if (sec == 0 || extra_of(row) >= 0) {
                            const size_t pix = slot_pixel(row, sec);
                            const bool ok = pix < npx;
                            if (__ballot(alive_at(ok, pcs[2 * j + sec])) != 0ull) live |= 1u << (2 * j + sec);
                            else dead_slot(ok, pix);
                        }
                }
            }
        }
        live = __builtin_amdgcn_readfirstlane(live);
        STAMP(7);
        unsigned touched = 0u;  // rows this wave has a sum for
        while (live != 0u) {  // row by row; a wave's two pixels of a row side by side: their loads in flight together
            const int j = __builtin_ctz(live) >> 1;
            const unsigned both = (live >> (2 * j)) & 3u;
            live &= ~(3u << (2 * j));
            touched |= 1u << j;
            const unsigned row = base + j * gridDim.x;
            size_t pix1 = pixA, pix2 = pixB;
            V3 pc1 = pcA, pc2 = pcB;
            bool ok1 = okA, ok2 = okB;
            if (!single) {  // (from L1: pass 1 has just read them)
                if (both & 1u) ok1 = slot_point(row, 0, pix1, pc1);
                if (both & 2u) ok2 = slot_point(row, 1, pix2, pc2);
            }
            if (both == 3u) {
                const PixelTerms o1 = terms(ok1, pix1, pc1);
                const PixelTerms o2 = terms(ok2, pix2, pc2);
                store_terms(a, ok1, pix1, o1);
                store_terms(a, ok2, pix2, o2);
                STAMP(5);
                accumulate(o1, &o2, j);
            } else {
                const bool sec = both == 2u;
                const PixelTerms o1 = sec ? terms(ok2, pix2, pc2) : terms(ok1, pix1, pc1);
                store_terms(a, sec ? ok2 : ok1, sec ? pix2 : pix1, o1);
                STAMP(5);
                accumulate(o1, nullptr, j);
            }
        }
        for (int j = 0; j < nrows; ++j)
            if (!((touched >> j) & 1u)) {
                if (lane < kCols - 1) red[j][wave][lane] = 0.f;
                if (lane == 0) redMax[j][wave] = 0.f;
            }
        // (a barrier for red[] alone: __syncthreads() also waits for the wave's global stores -- up to sixteen zeros per
        // lane of a wave with dead slots -- to be acknowledged, 3 us that nothing here needs)
        lds_barrier();
#ifdef EMF_TRACK_TRACE
        STAMP(4);  // (probe builds: the stamp behind the state's store is overwritten by the pass's barrier)
#endif
        if (threadIdx.x < static_cast<unsigned>(nrows) * 32u) {
            const int j = threadIdx.x >> 5, col = threadIdx.x & 31;
            const unsigned row = base + j * gridDim.x;
            if (col < kCols - 1) {
                float v = red[j][0][col];
                for (int i = 1; i < kWaves; ++i) v += red[j][i][col];
                // component-major: the next prologue reads each component contiguously
                mine[static_cast<size_t>(col) * f.nblocks + row] = v;
            } else if (col == kCols - 1) {
                float v = redMax[j][0];
                for (int i = 1; i < kWaves; ++i) v = fmaxf(v, redMax[j][i]);
                mine[static_cast<size_t>(kCols - 1) * f.nblocks + row] = v;
            }
        }
    }
    STAMP(6);
#undef STAMP
}

// The stage's Huber and combined weight images at its final pose (emf_hip_trackWeightImages): the body's
// arithmetic for one pixel, nothing summed.
__global__ __launch_bounds__(256) void k_track_weight_images(const TrackFrame f, float* huber, float* track) {
    const int m = blockIdx.y;
    const size_t px = static_cast<size_t>(f.w) * f.h;
    const size_t pix = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (pix >= px) return;
    const emf_track_state_t& st = f.states[m];
    const emf_model_t& md = f.models[m];
    const int y = static_cast<int>(pix / f.w), x = static_cast<int>(pix - static_cast<size_t>(y) * f.w);
    const float* p = f.points.row(y) + 3 * x;
    const float r = lookup1(md.tsdf, state_R(st.R), v3(st.t[0], st.t[1], st.t[2]), v3(p[0], p[1], p[2]),
                            I3{md.res[0], md.res[1], md.res[2]}, md.voxelSize);
    const float a = fabsf(r);
    float tw = a != 0.f ? f.prm.huberThresh / a : 0.f;  // divide(scalar, mat): x / 0 := 0 (Q7)
    tw = fminf(tw, 1.0f);
    if (huber) huber[static_cast<size_t>(m) * px + pix] = tw;
    if (track) {
        const float mx = __uint_as_float(st.maxIwBits);
        const float scale = static_cast<double>(mx) > 2.220446049250313e-16 ? static_cast<float>(1.0 / static_cast<double>(mx)) : 0.f;
        float w = scratch_iw(f, m, st.iwSel)[pix] * scale;
        w = tw * w;
        w = w * md.assoc[pix];
        track[static_cast<size_t>(m) * px + pix] = w;
    }
}

struct PrepareArgs {
    emf_track_state_t* states;
    emf_pose_t poses[EMF_MAX_BATCH];
    int nmodels;
    float nuInit;
};

__global__ void k_track_prepare(const PrepareArgs a) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nmodels) return;
    emf_track_state_t st;
    for (int k = 0; k < 9; ++k) st.R[k] = st.Rtrial[k] = a.poses[m].R[k];
    for (int k = 0; k < 3; ++k) st.t[k] = st.ttrial[k] = a.poses[m].t[k];
    for (int k = 0; k < 36; ++k) st.A[k] = 0.f;
    for (int k = 0; k < 6; ++k) st.b[k] = st.x[k] = 0.f;
    st.mu = 0.f;
    st.nu = a.nuInit;  // TSDF.cpp:188-191
    st.rho = st.err = st.errNew = 0.f;
    st.maxIwBits = st.maxIwTrialBits = 0u;
    st.iwSel = 0;
    st.converged = 0;
    st.firstIteration = 1;
    st.evaluateGradient = 1;
    st.haveTrial = 0;
    st.iterations = 0;
    st.accepted = 0;
    st.wSel = 0;
    st.needAccum = 1;
    st.haveSpec = 0;
    for (int k = 0; k < 28; ++k) st.spec[k] = 0.f;
    st.checkB = 0;
    st.pending = st.body = 0;
    st.iterTarget = 0;
    st.logCur = st.logTrial = se3_log_norm(state_R(st.R), v3(st.t[0], st.t[1], st.t[2]));
    st.wFac = 1.f;
    a.states[m] = st;
}

// level 1: kernel_computePoseGradients as an image (TSDF.cu:603-660)
struct PoseGradArgs {
    const float* tsdf;
    const float* grads;
    Img<const float> points;
    float* out;  // (W*H) x 6
    M33 R;
    V3 t;
    I3 n;
    float voxelSize;
    int w, h;
};

__global__ __launch_bounds__(kTrackBlock) void k_pose_gradients(const PoseGradArgs a) {
    const size_t pix = static_cast<size_t>(blockIdx.x) * kTrackBlock + threadIdx.x;
    if (pix >= static_cast<size_t>(a.w) * a.h) return;
    const int y = static_cast<int>(pix / a.w), x = static_cast<int>(pix - static_cast<size_t>(y) * a.w);
    const float* p = a.points.row(y) + 3 * x;
    float g[6];
    pose_gradient(a.tsdf, a.grads, a.R, a.t, v3(p[0], p[1], p[2]), a.n, a.voxelSize, g);
    float* o = a.out + 6 * pix;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = g[k];  // every pixel written: the setTo(0) is folded in
}

// CUs of the current device (looked up once per device: an immutable property, not state)
int compute_units() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int fill_frame(TrackFrame& f, const emf_model_t* models_dev, emf_track_state_t* states_dev,
               int nmodels, const emf_image_t* points, const emf_track_params_t* prm,
               void* scratch_dev, size_t scratchBytesPerModel, const char* fn) {
    if (!models_dev || !states_dev || !prm || !scratch_dev) return fail(EMF_E_NULL, "%s: NULL argument", fn);
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH)
        return fail(EMF_E_LIMIT, "%s: nmodels = %d, expected 1..%d", fn, nmodels, EMF_MAX_BATCH);
    EMF_TRY(check_image(points, 12, fn));
    f.models = models_dev;
    f.states = states_dev;
    f.nmodels = nmodels;
    f.points = img<const float>(points);
    f.w = points->width;
    f.h = points->height;
    if (static_cast<size_t>(f.w) * f.h >= (static_cast<size_t>(1) << 31))  // (the step kernel indexes pixels in 32 bits)
        return fail(EMF_E_LIMIT, "%s: %d x %d pixels, expected fewer than 2^31", fn, f.w, f.h);
    f.nblocks = static_cast<int>(ceil_div(static_cast<size_t>(f.w) * f.h, kRowPixels));
    f.prm = *prm;
    f.scratch = static_cast<char*>(scratch_dev);
    f.scratchStride = scratchBytesPerModel;
    const char* const rs = std::getenv("EMF_TRACK_RESCALE");  // (A/B and the test of the launch it saves; read per call)
    f.rescale = rs ? std::atoi(rs) != 0 : 1;
    if (scratchBytesPerModel < emf_hip_trackScratchBytes(f.w, f.h) || scratchBytesPerModel % 16)
        return fail(EMF_E_ARG, "%s: scratch of %zu bytes per model, need %zu (multiple of 16)", fn,
                    scratchBytesPerModel, emf_hip_trackScratchBytes(f.w, f.h));
    return EMF_OK;
}

}  // namespace
}  // namespace emf_hip

using namespace emf_hip;

extern "C" {

size_t emf_hip_trackScratchBytes(int width, int height) {
    const size_t px = static_cast<size_t>(width) * height;
    const size_t nblocks = ceil_div(px, kRowPixels);
    size_t bytes = (4 * px + 2 * nblocks * kCols) * sizeof(float) + sizeof(emf_track_state_t);
#ifdef EMF_TRACK_TRACE
    bytes += (24 * 8 + 2 * nblocks + 2 * 16) * sizeof(long long);
#endif
    return (bytes + 255) / 256 * 256;
}

int emf_hip_trackPrepare(emf_track_state_t* states_dev, const emf_pose_t* poseCO_host, int nmodels,
                         float nuInit, emf_stream_t stream) {
    if (!states_dev || !poseCO_host) return fail(EMF_E_NULL, "trackPrepare: NULL argument");
    if (nmodels < 1 || nmodels > EMF_MAX_BATCH)
        return fail(EMF_E_LIMIT, "trackPrepare: nmodels = %d, expected 1..%d", nmodels, EMF_MAX_BATCH);
    PrepareArgs a;
    a.states = states_dev;
    for (int m = 0; m < nmodels; ++m) a.poses[m] = poseCO_host[m];
    a.nmodels = nmodels;
    a.nuInit = nuInit;
    hipLaunchKernelGGL(k_track_prepare, dim3(1), dim3(64), 0, as_stream(stream), a);
    return launch_status("trackPrepare");
}

namespace {
// launch `launch` of a stage (see emf_hip_trackStep)
void enqueue_step(TrackFrame& f, int nmodels, int launch, hipStream_t s) {
    const dim3 px(ceil_div(static_cast<size_t>(f.w) * f.h, kTrackBlock), static_cast<unsigned>(nmodels));
    // the weight maximum is looked up at the first pose of a stage only (device flag); in a later
    // call of the stage the kernel returns at once
    // (only where the first pass cannot fold it in: EMF_TRACK_RESCALE=0)
    if (launch == 0 && !f.rescale) hipLaunchKernelGGL(k_track_maxw, px, dim3(kTrackBlock), 0, s, f);
    // all workgroups of a launch resident at once (two per CU), each taking its share of the blocks
    f.launch = launch;
    // a workgroup per CU, shared out among the models; each takes its share of a model's rows
    const int perModel = std::max(1, std::min(f.nblocks, compute_units() / nmodels));
    hipLaunchKernelGGL(k_track_step, dim3(static_cast<unsigned>(perModel), static_cast<unsigned>(nmodels)),
                       dim3(kTrackBlock), 0, s, f);
}
}  // namespace

int emf_hip_trackIterate(const emf_model_t* models_dev, emf_track_state_t* states_dev, int nmodels,
                         const emf_image_t* points, const emf_track_params_t* params,
                         void* scratch_dev, size_t scratchBytesPerModel, int iterations,
                         emf_stream_t stream) {
    TrackFrame f;
    EMF_TRY(fill_frame(f, models_dev, states_dev, nmodels, points, params, scratch_dev,
                       scratchBytesPerModel, "trackIterate"));
    if (iterations < 0) return fail(EMF_E_ARG, "trackIterate: iterations = %d", iterations);
    if (iterations == 0) return EMF_OK;
    // sums at the first pose + one launch per iteration + the last step's verdict + one spare for a
    // speculation miss; an even number, so that the state ends in the caller's array
    const int launches = (iterations + 3 + 1) & ~1;
    f.iterations = iterations;
    f.watch = nullptr;
    f.finalStates = nullptr;
    f.seq = 0;
    for (int i = 0; i < launches; ++i) enqueue_step(f, nmodels, i, as_stream(stream));
    return launch_status("trackIterate");
}

int emf_hip_trackStep(const emf_model_t* models_dev, emf_track_state_t* states_dev, int nmodels,
                      const emf_image_t* points, const emf_track_params_t* params,
                      void* scratch_dev, size_t scratchBytesPerModel, int launch, int iterations,
                      uint32_t* watch, uint32_t seq, emf_track_state_t* finalStates, emf_stream_t stream) {
    TrackFrame f;
    EMF_TRY(fill_frame(f, models_dev, states_dev, nmodels, points, params, scratch_dev,
                       scratchBytesPerModel, "trackStep"));
    if (launch < 0 || iterations < 0) return fail(EMF_E_ARG, "trackStep: launch = %d, iterations = %d", launch, iterations);
    f.iterations = iterations;
    f.watch = watch;
    f.finalStates = finalStates;
    f.seq = seq;
    enqueue_step(f, nmodels, launch, as_stream(stream));
    return launch_status("trackStep");
}

int emf_hip_trackWeightImages(const emf_model_t* models_dev, const emf_track_state_t* states_dev, int nmodels,
                              const emf_image_t* points, const emf_track_params_t* params,
                              const void* scratch_dev, size_t scratchBytesPerModel, float* huber_dev,
                              float* track_dev, emf_stream_t stream) {
    TrackFrame f;
    EMF_TRY(fill_frame(f, models_dev, const_cast<emf_track_state_t*>(states_dev), nmodels, points, params,
                       const_cast<void*>(scratch_dev), scratchBytesPerModel, "trackWeightImages"));
    if (!huber_dev && !track_dev) return EMF_OK;
    f.launch = f.iterations = 0;
    f.watch = nullptr;
    f.finalStates = nullptr;
    f.seq = 0;
    const size_t px = static_cast<size_t>(f.w) * f.h;
    hipLaunchKernelGGL(k_track_weight_images, dim3(static_cast<unsigned>(ceil_div(px, 256)), static_cast<unsigned>(nmodels)),
                       dim3(256), 0, as_stream(stream), f, huber_dev, track_dev);
    return launch_status("trackWeightImages");
}

int emf_hip_computePoseGradients(const float* tsdf, const float* grads, const emf_image_t* points,
                                 const float R_CO[9], const float t_CO[3], const int32_t res[3],
                                 float voxelSize, float* grads6, emf_stream_t stream) {
    EMF_REQUIRE_PTR(tsdf);
    EMF_REQUIRE_PTR(grads6);
    EMF_REQUIRE_PTR(R_CO);
    EMF_REQUIRE_PTR(t_CO);
    EMF_TRY(check_image(points, 12, "computePoseGradients: points"));
    EMF_TRY(check_res(res));
    if (!(voxelSize > 0.f)) return fail(EMF_E_ARG, "computePoseGradients: voxelSize %g", voxelSize);
    PoseGradArgs a;
    a.tsdf = tsdf;
    a.grads = grads;
    a.points = img<const float>(points);
    a.out = grads6;
    a.R = m33_from(R_CO);
    a.t = v3_from(t_CO);
    a.n = i3_from(res);
    a.voxelSize = voxelSize;
    a.w = points->width;
    a.h = points->height;
    hipLaunchKernelGGL(k_pose_gradients,
                       dim3(static_cast<unsigned>(ceil_div(static_cast<size_t>(a.w) * a.h, kTrackBlock))),
                       dim3(kTrackBlock), 0, as_stream(stream), a);
    return launch_status("computePoseGradients");
}

}  // extern "C"

static_assert(sizeof(emf_track_state_t) == 496, "emf_track_state_t layout is mirrored in _lib.py");
static_assert(sizeof(emf_model_t) == 168, "emf_model_t layout is mirrored in _lib.py");
