// lk_cuda.hip — the reference's GPU tracker rule on gfx950: FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163) = forward + backward
// cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), maxLevel 3, 30 iterations[, useInitialFlow]) + |p - p_rev| <= 1.0 + InBorder, used where the reference uses it: the
// temporal and right-image tracking of TrackImageNaive and the right image of TrackSemanticImage (instance_feature.cpp:191-310, background_tracker.cpp:437-456,797-798).
// A different tracker from the CPU rule of lk.hip, not a second implementation of it: float patches sampled bilinearly (texture semantics: 8-bit fractions,
// normalised 8-bit reads, clamp addressing), Scharr derivatives on the fly, no minimum-eigenvalue test, D < FLT_EPSILON only, stop at |dx|, |dy| < 0.01, its own
// pyramid (cuda::pyrDown rounds half to even), status cleared at level 0 only, and a level that bails out early leaves the iterate of the coarser level in place.
// The two platform-dependent details of the CUDA original that cannot be recovered are fixed by declaration: the texture unit's interpolation
// arithmetic (programming-guide formula, float, left to right) and nvcc's floating-point contraction (none here: -ffp-contract=off).
//
// Mapping: the original's own — one 256-thread workgroup per point (16 x 16 threads, 2 x 2 window pixels each), because the float sums must be formed in the
// original's order to be comparable at all: per-thread partial sums in patch order, then cudev's blockReduce<256> = shuffle-down trees over 32-lane groups and one
// over the eight group results (a wave64 runs two such groups side by side: __shfl_down(.., width 32)).  Forward levels 3..0, backward levels 3..0, the distance
// and border tests in the same launch.
#include <cfloat>
#include "dv_internal.h"

#define LKC_WIN 21
#define LKC_HALF 10
namespace {

// One texture fetch.  The texels come out of an LDS tile staged with the clamp addressing and the 8-bit normalisation (u8 / 255.0f) already applied (tile entry (r, c) =
// image pixel (clamp(y0 + r), clamp(x0 + c)) as a float: the division is done once per staged texel instead of once per read);
// a fetch whose 2 x 2 footprint leaves the tile (float rounding at a tile edge: never seen, kept for safety) reads the image itself.  The arithmetic is the same in
// both paths.
struct LkcTile { const float* t; int x0, y0, cols, rows; };
__device__ __forceinline__ float lkc_tex(const DvLevel& L, const LkcTile& T, float x, float y) {
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const int i = (int)fx, j = (int)fy;
    const float a = floorf((xb - fx) * 256.f + 0.5f) * (1.f / 256.f), b = floorf((yb - fy) * 256.f + 0.5f) * (1.f / 256.f);
    float t00, t10, t01, t11;
    const int li = i - T.x0, lj = j - T.y0;
    if ((unsigned)li < (unsigned)(T.cols - 1) && (unsigned)lj < (unsigned)(T.rows - 1)) {
        const float* r0 = T.t + lj * T.cols + li;
        t00 = r0[0]; t10 = r0[1]; t01 = r0[T.cols]; t11 = r0[T.cols + 1];
    } else {
        const int x0 = min(max(i, 0), L.w - 1), x1 = min(max(i + 1, 0), L.w - 1), y0 = min(max(j, 0), L.h - 1), y1 = min(max(j + 1, 0), L.h - 1);
        const uint8_t* r0 = L.p + (ptrdiff_t)y0 * L.pitch; const uint8_t* r1 = L.p + (ptrdiff_t)y1 * L.pitch;
        t00 = (float)r0[x0] / 255.0f; t10 = (float)r0[x1] / 255.0f; t01 = (float)r1[x0] / 255.0f; t11 = (float)r1[x1] / 255.0f;
    }
    float v = (1.f - a) * (1.f - b) * t00;
    v = v + a * (1.f - b) * t10;
    v = v + (1.f - a) * b * t01;
    v = v + a * b * t11;
    return v;
}
// stage rows [y0, y0 + rows) x columns [x0, x0 + cols) of level L, clamped, into LDS (all 256 threads; the caller synchronises)
__device__ __forceinline__ void lkc_stage(float* lds, const DvLevel& L, int x0, int y0, int cols, int rows) {
    for (int e = threadIdx.x; e < cols * rows; e += 256) {
        const int r = e / cols, c = e - r * cols;
        const int sx = min(max(x0 + c, 0), L.w - 1), sy = min(max(y0 + r, 0), L.h - 1);
        lds[e] = (float)L.p[(ptrdiff_t)sy * L.pitch + sx] / 255.0f;
    }
}
#define LKC_IT 26          // I tile: window 21 + Scharr halo 2 + bilinear 1 + one spare column / row on each side
#define LKC_JT 34          // J tile: window 21 + bilinear 1 + 2 x (5 px of drift + 1 spare)
#define LKC_JM 6

// cudev::blockReduce<256> of N values at once, the result in every thread.  The summation tree is the original's, term for term: shuffle-down by 16, 8, 4, 2, 1 inside each
// 32-lane group (lane i adds lane i + d; only the lanes whose operands are in range feed lane 0), then the eight group sums as ((s0+s4)+(s2+s6))+((s1+s5)+(s3+s7)) =
// the shuffle-down tree over 8.  What changed is the data path: the lane shifts stay in the VALU (v_permlane16_swap for 16, DPP row_shl for 8..1) instead of
// ds_bpermute round trips, the N sums share ONE barrier (every thread reads the eight group sums back and finishes the tree itself), and the LDS slots alternate
// between two buffers so that the next call's writes cannot overtake this call's reads.
template <int CTRL> __device__ __forceinline__ float lkc_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
typedef unsigned lkc_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lkc_down16(float v) {      // lanes 0..15 of each 32-lane half receive lanes 16..31
    const lkc_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[1]);
}
struct LkcRed { float* buf; int phase; };       // buf: [2][3][8] floats
template <int N> __device__ __forceinline__ void lkc_reduce(float (&v)[N], LkcRed& R) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] + lkc_down16(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] + lkc_dpp<0x108>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] + lkc_dpp<0x104>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] + lkc_dpp<0x102>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] + lkc_dpp<0x101>(v[k]);
    float* s = R.buf + R.phase * 24; R.phase ^= 1;
    if ((tid & 31) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) s[k * 8 + (tid >> 5)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float4 lo = *reinterpret_cast<const float4*>(s + k * 8), hi = *reinterpret_cast<const float4*>(s + k * 8 + 4);
        v[k] = ((lo.x + hi.x) + (lo.z + hi.z)) + ((lo.y + hi.y) + (lo.w + hi.w));
    }
}

struct LkcPt { float nx, ny; bool st; };

// pyrlk::sparseKernel<1, 2, 2, false, uchar> for the workgroup's point at one level; all control flow is workgroup-uniform
__device__ __forceinline__ void lkc_level(const DvLevel& I, const DvLevel& J, float px, float py, LkcPt& o, int level, int iters, LkcRed& R, float* sI, float* sJ) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int rows = I.h, cols = I.w;
    float prx = px * (1.0f / (float)(1 << level)), pry = py * (1.0f / (float)(1 << level));
    if (prx < 0 || prx >= cols || pry < 0 || pry >= rows) { if (level == 0) o.st = false; return; }
    prx -= (float)LKC_HALF; pry -= (float)LKC_HALF;
    LkcTile TI{ sI, (int)floorf(prx) - 2, (int)floorf(pry) - 2, LKC_IT, LKC_IT };
    __syncthreads();                                    // the previous level's readers of the tiles are done
    lkc_stage(sI, I, TI.x0, TI.y0, LKC_IT, LKC_IT);
    __syncthreads();
    float Ip[2][2], Dx[2][2], Dy[2][2];
    float a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yb = ty + 16 * i, xb = tx + 16 * j;
            Ip[i][j] = 0.f; Dx[i][j] = 0.f; Dy[i][j] = 0.f;
            if (yb < LKC_WIN && xb < LKC_WIN) {
                const float x = prx + xb + 0.5f, y = pry + yb + 0.5f;
                Ip[i][j] = lkc_tex(I, TI, x, y);
                const float dIdx = 3.0f * lkc_tex(I, TI, x + 1, y - 1) + 10.0f * lkc_tex(I, TI, x + 1, y) + 3.0f * lkc_tex(I, TI, x + 1, y + 1) -
                                   (3.0f * lkc_tex(I, TI, x - 1, y - 1) + 10.0f * lkc_tex(I, TI, x - 1, y) + 3.0f * lkc_tex(I, TI, x - 1, y + 1));
                const float dIdy = 3.0f * lkc_tex(I, TI, x - 1, y + 1) + 10.0f * lkc_tex(I, TI, x, y + 1) + 3.0f * lkc_tex(I, TI, x + 1, y + 1) -
                                   (3.0f * lkc_tex(I, TI, x - 1, y - 1) + 10.0f * lkc_tex(I, TI, x, y - 1) + 3.0f * lkc_tex(I, TI, x + 1, y - 1));
                Dx[i][j] = dIdx; Dy[i][j] = dIdy;
                a11 += dIdx * dIdx; a12 += dIdx * dIdy; a22 += dIdy * dIdy;
            }
        }
    float A[3] = { a11, a12, a22 };
    lkc_reduce<3>(A, R);
    float A11 = A[0], A12 = A[1], A22 = A[2];
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) { if (level == 0) o.st = false; return; }
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    float qx = o.nx * 2.f, qy = o.ny * 2.f;
    qx -= (float)LKC_HALF; qy -= (float)LKC_HALF;
    LkcTile TJ{ sJ, 0, 0, LKC_JT, LKC_JT }; bool have_j = false;
    for (int k = 0; k < iters; ++k) {
        if (qx < -(float)LKC_HALF || qx >= cols || qy < -(float)LKC_HALF || qy >= rows) { if (level == 0) o.st = false; return; }
        {   // the J tile follows the iterate: re-centred when it has drifted more than 5 px from the tile's centre (workgroup-uniform)
            const int jx = (int)floorf(qx), jy = (int)floorf(qy);
            if (!have_j || abs(jx - (TJ.x0 + LKC_JM)) > LKC_JM - 1 || abs(jy - (TJ.y0 + LKC_JM)) > LKC_JM - 1) {
                __syncthreads();
                TJ.x0 = jx - LKC_JM; TJ.y0 = jy - LKC_JM;
                lkc_stage(sJ, J, TJ.x0, TJ.y0, LKC_JT, LKC_JT);
                __syncthreads();
                have_j = true;
            }
        }
        float b1 = 0, b2 = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = ty + 16 * i, x = tx + 16 * j;
                if (y < LKC_WIN && x < LKC_WIN) {
                    const float Jv = lkc_tex(J, TJ, qx + x + 0.5f, qy + y + 0.5f);
                    const float diff = (Jv - Ip[i][j]) * 32.0f;
                    b1 += diff * Dx[i][j];
                    b2 += diff * Dy[i][j];
                }
            }
        float bb[2] = { b1, b2 };
        lkc_reduce<2>(bb, R);
        b1 = bb[0]; b2 = bb[1];
        const float dx = A12 * b2 - A22 * b1, dy = A12 * b1 - A11 * b2;
        qx += dx; qy += dy;
        if (fabsf(dx) < 0.01f && fabsf(dy) < 0.01f) break;
    }
    o.nx = qx + (float)LKC_HALF; o.ny = qy + (float)LKC_HALF;
}

__device__ __forceinline__ bool lkc_in_border(float x, float y, int rows, int cols) {
    const int ix = __float2int_rn(x), iy = __float2int_rn(y);
    return 1 <= ix && ix < cols - 1 && 1 <= iy && iy < rows - 1;
}

// one direction: PyrLKOpticalFlowBase::sparse (nextPts = init * (1 / 2^maxLevel / 2); status = 1; levels maxLevel .. 0)
__device__ __forceinline__ LkcPt lkc_sparse(const DvPyr& A, const DvPyr& B, float px, float py, float ix, float iy, int max_level, int iters, LkcRed& R, float* sI, float* sJ) {
    const float scale = (float)(1.0 / (1 << max_level) / 2.0);
    LkcPt o; o.nx = ix * scale; o.ny = iy * scale; o.st = true;
    for (int level = max_level; level >= 0; --level) { const DvLevel I = A.L[level], J = B.L[level]; lkc_level(I, J, px, py, o, level, iters, R, sI, sJ); }
    return o;
}

__global__ __launch_bounds__(256) void lk_cuda_track_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, const int* __restrict__ n_dev, int n_host, int flow_back, float dist_thresh,
                                                            int max_level, int iters, float2* __restrict__ pts_b, uint8_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) float sred[48];
    __shared__ float sI[LKC_IT * LKC_IT], sJ[LKC_JT * LKC_JT];
    LkcRed R{ sred, 0 };
    const int p = blockIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    if (p >= n) return;
    const float2 prev = pts_a[p];
    const LkcPt f = lkc_sparse(A, B, prev.x, prev.y, prev.x, prev.y, max_level, iters, R, sI, sJ);
    bool st = f.st;
    if (flow_back) {          // lkOpticalFlowBack->calc(img_next, img_prev, d_nextPts, d_reverse_pts = d_prevPts, ...): the previous points are the initial flow
        const LkcPt r = lkc_sparse(B, A, f.nx, f.ny, prev.x, prev.y, max_level, iters, R, sI, sJ);
        const float dx = prev.x - r.nx, dy = prev.y - r.ny;
        st = st && r.st && sqrtf(dx * dx + dy * dy) <= dist_thresh;
    }
    if (st && !lkc_in_border(f.nx, f.ny, B.L[0].h, B.L[0].w)) st = false;
    if (threadIdx.x == 0) { pts_b[p] = make_float2(f.nx, f.ny); status[p] = st ? 1 : 0; }
}
// single direction (parity tests of SparsePyrLKOpticalFlow::calc itself)
__global__ __launch_bounds__(256) void lk_cuda_generic_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, int n, int max_level, int iters, int use_initial,
                                                              float2* __restrict__ pts_b, uint8_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) float sred[48];
    __shared__ float sI[LKC_IT * LKC_IT], sJ[LKC_JT * LKC_JT];
    LkcRed R{ sred, 0 };
    const int p = blockIdx.x;
    if (p >= n) return;
    const float2 prev = pts_a[p];
    const float2 init = use_initial ? pts_b[p] : prev;
    const LkcPt f = lkc_sparse(A, B, prev.x, prev.y, init.x, init.y, max_level, iters, R, sI, sJ);
    __syncthreads();
    if (threadIdx.x == 0) { pts_b[p] = make_float2(f.nx, f.ny); status[p] = f.st ? 1 : 0; }
}

}  // namespace

void dv_launch_lk_cuda_track(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max, int flow_back, float dist_thresh, float2* pts_b, uint8_t* status, hipStream_t s) {
    if (n_max <= 0) return;
    const int ml = (A.levels < B.levels ? A.levels : B.levels) - 1;
    hipLaunchKernelGGL(lk_cuda_track_kernel, dim3(n_max), dim3(256), 0, s, A, B, pts_a, n_dev, n_max, flow_back, dist_thresh, ml < 3 ? ml : 3, 30, pts_b, status);
}
void dv_launch_lk_cuda_generic(const DvPyr& A, const DvPyr& B, const float2* pts_a, int n, int max_level, int iters, int use_initial, float2* pts_b, uint8_t* status, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lk_cuda_generic_kernel, dim3(n), dim3(256), 0, s, A, B, pts_a, n, max_level, iters, use_initial, pts_b, status);
}
