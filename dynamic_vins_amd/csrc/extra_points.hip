// extra_points.hip — the object "extra point" pipeline of dynamic mode on gfx950 (second half), ONE launch per frame for all visible objects:
//   InstFeat::DetectExtraPoints            front_end/instance_feature.cpp:413-461   strided disparity sampling inside the ROI mask -> camera-frame 3-D points
//   InstsFeatManager::ProcessExtraPoints   front_end/dynamic_tracker.cpp:268-338    pcl::RadiusOutlierRemoval(0.5 m, 10 neighbours) + pcl::EuclideanClusterExtraction(1 m,
//                                                                                   10..25000 points), cluster_indices[0] replaces extra_points3d
// One 1024-thread workgroup per object, everything in LDS (<= 3200 points: the sampling step max(sqrt(0.8 rows cols / 1000), 2) bounds the grid):
//   1. sampling: one thread per grid node, the reference's tests in its order (mask, disparity <= 0, NaN, 0.1 < depth <= 100) and its float arithmetic
//      (-ffp-contract=off); survivors are compacted with ballot + prefix sums so that the output order IS the reference's row-major scan;
//   2. radius filter: a point survives iff at least 11 points (itself included) lie within d^2 <= 0.25 — what the k-nearest form of
//      RadiusOutlierRemoval::applyFilterIndices decides on a dense cloud; distances as flann::L2_Simple<float> ((dx dx + dy dy) + dz dz); order kept;
//      fewer than 5 survivors: the object gets no extra points (dynamic_tracker.cpp:287-289);
//   3. Euclidean clustering: PCL grows regions over a strict radius search (d^2 < 1) from seeds in index order — the clusters are the connected components
//      of that graph, numbered by lowest member.  Here: min-label propagation with pointer jumping until nothing changes (a few sweeps; every sweep is an
//      all-pairs pass over LDS), component sizes by LDS atomics, the largest component with 10 <= size <= 25000 (equal sizes: the lowest label = the cluster
//      PCL finds first), members written in ascending index order (PCL sorts the indices) as doubles (PclToEigen widens the floats).
// Integer / index work and float comparisons only: bit-exact against the CPU restatement the tests hold (tests/test_extra_points.py).
// The mask read is the detection's mask as uploaded — NOT the 5x5-eroded one: the reference's extra-point thread races with the in-place erosion on the
// tracking thread (dynamic_tracker.cpp:378 / :425); the sampling is the first thing that thread does, so the un-eroded mask is the canonical reading.
#include <algorithm>
#include <cmath>
#include "dv_internal.h"
#include "dev_once.h"

#define XP_THREADS 1024

namespace {

__device__ __forceinline__ float xp_d2(const float4 a, const float4 b) {      // flann::L2_Simple<float>, dimension 3
    float r = 0.f, d;
    d = a.x - b.x; r += d * d;
    d = a.y - b.y; r += d * d;
    d = a.z - b.z; r += d * d;
    return r;
}

// order-preserving position of a flagged element inside the block: exclusive prefix over the block's flags; *total = number of flags.  Two barriers.
__device__ __forceinline__ int xp_block_prefix(bool flag, int* s_wsum, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                      // s_wsum of the previous call has been consumed
    if (lane == 0) s_wsum[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < XP_THREADS / 64; ++w) { const int v = s_wsum[w]; if (w < wv) before += v; sum += v; }
    *total = sum;
    return before + in_wave;
}

__global__ __launch_bounds__(XP_THREADS) void extra_points_kernel(const DvExtraJob* __restrict__ jobs, DvExtraArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xp_smem[];
    float4* A = reinterpret_cast<float4*>(xp_smem);                   // sampled points
    float4* B = A + DV_XP_CAP;                                         // after the radius filter
    int* label = reinterpret_cast<int*>(B + DV_XP_CAP);
    int* csize = label + DV_XP_CAP;
    __shared__ int s_wsum[XP_THREADS / 64];
    __shared__ int s_changed;
    __shared__ unsigned long long s_best;
    const DvExtraJob j = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    // ---- 1. DetectExtraPoints ----
    const int step = j.step;
    const int ni = (j.rows + step - 1) / step, nj = (j.cols + step - 1) / step, S = ni * nj;
    int n = 0;
    for (int base = 0; base < S; base += XP_THREADS) {
        const int k = base + tid;
        bool ok = false; float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < S) {
            const int gi = k / nj, gj = k - gi * nj, i = gi * step, jx = gj * step;
            if (j.mask[(size_t)i * j.mask_pitch + jx] != 0) {                                 // mask_cv.at<uchar>(i, j) <= 0.5 -> skip
                const int r = (int)((float)i + (float)j.box_y), c = (int)((float)jx + (float)j.box_x);      // int + Rect2f::tl() (float), truncated
                if (r >= 0 && r < a.disp_h && c >= 0 && c < a.disp_w) {
                    const float disparity = a.disp[(size_t)r * a.disp_pitch + c];
                    if (!(disparity <= 0.f) && disparity == disparity) {
                        const float depth = a.fx0 * a.baseline / disparity;
                        if (!((double)depth <= 0.1 || (double)depth > 100.0)) {
                            p.x = ((float)c - a.cx0) * depth / a.fx0;
                            p.y = ((float)r - a.cy0) * depth / a.fy0;
                            p.z = depth;
                            ok = true;
                        }
                    }
                }
            }
        }
        int total;
        const int pos = n + xp_block_prefix(ok, s_wsum, &total);
        if (ok && pos < DV_XP_CAP) A[pos] = p;
        n += total;
    }
    if (n > DV_XP_CAP) { if (tid == 0 && a.err_flag) atomicOr(a.err_flag, 8); n = DV_XP_CAP; }
    __syncthreads();
    if (a.stage == 1) {          // operator form, DetectExtraPoints alone
        for (int i = tid; i < n; i += XP_THREADS) { const float4 p = A[i]; double* o = j.out + 3 * (size_t)i; o[0] = (double)p.x; o[1] = (double)p.y; o[2] = (double)p.z; }
        if (tid == 0) *j.n_out = n;
        return;
    }
    // ---- 2. RadiusOutlierRemoval(0.5, 10) ----
    int m = 0;
    for (int base = 0; base < n; base += XP_THREADS) {
        const int i = base + tid;
        bool keep = false; float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            p = A[i];
            int cnt = 0;
            for (int q = 0; q < n; ++q) cnt += (xp_d2(p, A[q]) <= 0.25f) ? 1 : 0;           // (all lanes read the same A[q]: an LDS broadcast)
            keep = cnt >= 11;                                                                // k = 11 nearest exist and the farthest is not beyond the radius
        }
        int total;
        const int pos = m + xp_block_prefix(keep, s_wsum, &total);
        if (keep) B[pos] = p;
        m += total;
    }
    __syncthreads();
    int result = 0;
    if (m >= 5) {
        // ---- 3. connected components of {d^2 < 1} by min-label propagation ----
        for (int i = tid; i < m; i += XP_THREADS) { label[i] = i; csize[i] = 0; }
        if (tid == 0) s_best = 0ull;
        __syncthreads();
        while (true) {
            if (tid == 0) s_changed = 0;
            __syncthreads();
            for (int i = tid; i < m; i += XP_THREADS) {
                const float4 p = B[i];
                const int mine = label[i];
                int lmin = mine;
                for (int q = 0; q < m; ++q) if (xp_d2(p, B[q]) < 1.0f) lmin = min(lmin, label[q]);      // labels only ever decrease, and only to labels of the same component
                if (lmin < mine) { atomicMin(&label[i], lmin); s_changed = 1; }
            }
            __syncthreads();
            for (int hop = 0; hop < 4; ++hop) {                                                         // pointer jumping: label <- label[label]
                for (int i = tid; i < m; i += XP_THREADS) { const int l = label[i], ll = label[l]; if (ll < l) label[i] = ll; }
                __syncthreads();
            }
            if (!s_changed) break;
            __syncthreads();
        }
        // fixed point: label[i] = lowest index of i's component
        for (int i = tid; i < m; i += XP_THREADS) atomicAdd(&csize[label[i]], 1);
        __syncthreads();
        for (int i = tid; i < m; i += XP_THREADS) {
            const int sz = csize[i];
            if (label[i] == i && sz >= 10 && sz <= 25000) atomicMax(&s_best, ((unsigned long long)sz << 32) | (unsigned)(0x7fffffff - i));      // largest; among equals the first found
        }
        __syncthreads();
        const unsigned long long best = s_best;
        if (best != 0ull) {
            const int root = 0x7fffffff - (int)(unsigned)(best & 0xffffffffull);
            for (int base = 0; base < m; base += XP_THREADS) {
                const int i = base + tid;
                const bool in = i < m && label[i] == root;
                int total;
                const int pos = result + xp_block_prefix(in, s_wsum, &total);
                if (in) { const float4 p = B[i]; double* o = j.out + 3 * (size_t)pos; o[0] = (double)p.x; o[1] = (double)p.y; o[2] = (double)p.z; }
                result += total;
            }
        }
    }
    if (tid == 0) *j.n_out = result;
}

size_t xp_smem_bytes() { return (size_t)DV_XP_CAP * (2 * sizeof(float4) + 2 * sizeof(int)); }

}  // namespace

int dv_launch_extra_points(const DvExtraJob* jobs_dev, int n_jobs, const DvExtraArgs& a, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    static DevOnce once;
    if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(extra_points_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)xp_smem_bytes()) != hipSuccess; })) return -1;
    hipLaunchKernelGGL(extra_points_kernel, dim3(n_jobs), dim3(XP_THREADS), xp_smem_bytes(), s, jobs_dev, a);
    return 0;
}

// InstFeat::DetectExtraPoints' sampling step (instance_feature.cpp:421-422): double arithmetic, truncated
int dv_extra_points_step(int rows, int cols) {
    const float N_max = 1000.f;
    return (int)std::max(std::sqrt(0.8 * rows * cols / N_max), 2.);
}
