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

__device__ __forceinline__ float lkc_tex(const DvLevel& L, float x, float y) {
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const int i = (int)fx, j = (int)fy;
    const float a = floorf((xb - fx) * 256.f + 0.5f) * (1.f / 256.f), b = floorf((yb - fy) * 256.f + 0.5f) * (1.f / 256.f);
    const int x0 = min(max(i, 0), L.w - 1), x1 = min(max(i + 1, 0), L.w - 1), y0 = min(max(j, 0), L.h - 1), y1 = min(max(j + 1, 0), L.h - 1);
    const uint8_t* r0 = L.p + (ptrdiff_t)y0 * L.pitch; const uint8_t* r1 = L.p + (ptrdiff_t)y1 * L.pitch;
    const float t00 = (float)r0[x0] / 255.0f, t10 = (float)r0[x1] / 255.0f, t01 = (float)r1[x0] / 255.0f, t11 = (float)r1[x1] / 255.0f;
    float v = (1.f - a) * (1.f - b) * t00;
    v = v + a * (1.f - b) * t10;
    v = v + (1.f - a) * b * t01;
    v = v + a * b * t11;
    return v;
}

// cudev::blockReduce<256>: result in every thread
__device__ __forceinline__ float lkc_reduce(float v, float* s8, float* s1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v = v + __shfl_down(v, d, 32);
    __syncthreads();                                  // the previous sum's result has been read by everybody
    if ((tid & 31) == 0) s8[tid >> 5] = v;
    __syncthreads();
    if (tid < 8) {
        float w = s8[tid];
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) w = w + __shfl_down(w, d, 8);
        if (tid == 0) *s1 = w;
    }
    __syncthreads();
    return *s1;
}

struct LkcPt { float nx, ny; bool st; };

// pyrlk::sparseKernel<1, 2, 2, false, uchar> for the workgroup's point at one level; all control flow is workgroup-uniform
__device__ __forceinline__ void lkc_level(const DvLevel& I, const DvLevel& J, float px, float py, LkcPt& o, int level, int iters, float* s8, float* s1) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int rows = I.h, cols = I.w;
    float prx = px * (1.0f / (float)(1 << level)), pry = py * (1.0f / (float)(1 << level));
    if (prx < 0 || prx >= cols || pry < 0 || pry >= rows) { if (level == 0) o.st = false; return; }
    prx -= (float)LKC_HALF; pry -= (float)LKC_HALF;
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
                Ip[i][j] = lkc_tex(I, x, y);
                const float dIdx = 3.0f * lkc_tex(I, x + 1, y - 1) + 10.0f * lkc_tex(I, x + 1, y) + 3.0f * lkc_tex(I, x + 1, y + 1) -
                                   (3.0f * lkc_tex(I, x - 1, y - 1) + 10.0f * lkc_tex(I, x - 1, y) + 3.0f * lkc_tex(I, x - 1, y + 1));
                const float dIdy = 3.0f * lkc_tex(I, x - 1, y + 1) + 10.0f * lkc_tex(I, x, y + 1) + 3.0f * lkc_tex(I, x + 1, y + 1) -
                                   (3.0f * lkc_tex(I, x - 1, y - 1) + 10.0f * lkc_tex(I, x, y - 1) + 3.0f * lkc_tex(I, x + 1, y - 1));
                Dx[i][j] = dIdx; Dy[i][j] = dIdy;
                a11 += dIdx * dIdx; a12 += dIdx * dIdy; a22 += dIdy * dIdy;
            }
        }
    float A11 = lkc_reduce(a11, s8, s1), A12 = lkc_reduce(a12, s8, s1), A22 = lkc_reduce(a22, s8, s1);
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) { if (level == 0) o.st = false; return; }
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    float qx = o.nx * 2.f, qy = o.ny * 2.f;
    qx -= (float)LKC_HALF; qy -= (float)LKC_HALF;
    for (int k = 0; k < iters; ++k) {
        if (qx < -(float)LKC_HALF || qx >= cols || qy < -(float)LKC_HALF || qy >= rows) { if (level == 0) o.st = false; return; }
        float b1 = 0, b2 = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = ty + 16 * i, x = tx + 16 * j;
                if (y < LKC_WIN && x < LKC_WIN) {
                    const float Jv = lkc_tex(J, qx + x + 0.5f, qy + y + 0.5f);
                    const float diff = (Jv - Ip[i][j]) * 32.0f;
                    b1 += diff * Dx[i][j];
                    b2 += diff * Dy[i][j];
                }
            }
        b1 = lkc_reduce(b1, s8, s1); b2 = lkc_reduce(b2, s8, s1);
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
__device__ __forceinline__ LkcPt lkc_sparse(const DvPyr& A, const DvPyr& B, float px, float py, float ix, float iy, int max_level, int iters, float* s8, float* s1) {
    const float scale = (float)(1.0 / (1 << max_level) / 2.0);
    LkcPt o; o.nx = ix * scale; o.ny = iy * scale; o.st = true;
    for (int level = max_level; level >= 0; --level) { const DvLevel I = A.L[level], J = B.L[level]; lkc_level(I, J, px, py, o, level, iters, s8, s1); }
    return o;
}

__global__ __launch_bounds__(256) void lk_cuda_track_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, const int* __restrict__ n_dev, int n_host, int flow_back, float dist_thresh,
                                                            int max_level, int iters, float2* __restrict__ pts_b, uint8_t* __restrict__ status) {
    __shared__ float s8[8]; __shared__ float s1;
    const int p = blockIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    if (p >= n) return;
    const float2 prev = pts_a[p];
    const LkcPt f = lkc_sparse(A, B, prev.x, prev.y, prev.x, prev.y, max_level, iters, s8, &s1);
    bool st = f.st;
    if (flow_back) {          // lkOpticalFlowBack->calc(img_next, img_prev, d_nextPts, d_reverse_pts = d_prevPts, ...): the previous points are the initial flow
        const LkcPt r = lkc_sparse(B, A, f.nx, f.ny, prev.x, prev.y, max_level, iters, s8, &s1);
        const float dx = prev.x - r.nx, dy = prev.y - r.ny;
        st = st && r.st && sqrtf(dx * dx + dy * dy) <= dist_thresh;
    }
    if (st && !lkc_in_border(f.nx, f.ny, B.L[0].h, B.L[0].w)) st = false;
    if (threadIdx.x == 0) { pts_b[p] = make_float2(f.nx, f.ny); status[p] = st ? 1 : 0; }
}
// single direction (parity tests of SparsePyrLKOpticalFlow::calc itself)
__global__ __launch_bounds__(256) void lk_cuda_generic_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, int n, int max_level, int iters, int use_initial,
                                                              float2* __restrict__ pts_b, uint8_t* __restrict__ status) {
    __shared__ float s8[8]; __shared__ float s1;
    const int p = blockIdx.x;
    if (p >= n) return;
    const float2 prev = pts_a[p];
    const float2 init = use_initial ? pts_b[p] : prev;
    const LkcPt f = lkc_sparse(A, B, prev.x, prev.y, init.x, init.y, max_level, iters, s8, &s1);
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
