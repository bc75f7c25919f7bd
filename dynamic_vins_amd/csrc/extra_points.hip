// extra_points.hip — the object "extra point" pipeline of dynamic mode on gfx950 (second half), ONE launch per frame for all visible objects:
//   InstFeat::DetectExtraPoints            front_end/instance_feature.cpp:413-461   strided disparity sampling inside the ROI mask -> camera-frame 3-D points
//   InstsFeatManager::ProcessExtraPoints   front_end/dynamic_tracker.cpp:268-338    pcl::RadiusOutlierRemoval(0.5 m, 10 neighbours) + pcl::EuclideanClusterExtraction(1 m,
//                                                                                   10..25000 points), cluster_indices[0] replaces extra_points3d
// Per object (<= 3200 points: the sampling step max(sqrt(0.8 rows cols / 1000), 2) bounds the grid):
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

// The pipeline is five kernels on the object tracker's side stream, all objects of the frame in each launch (blockIdx.y = object where a stage is spread over
// several workgroups): one CU per object was compute-bound — n^2 = 8.4 M pair tests per pass at n = 2900, five to eight passes —, so the two all-pairs stages
// (radius filter, label sweeps) run as ceil(n / 256) workgroups per object, each with the object's whole cloud in LDS and 256 of its points to answer for.
//   xp_detect   (1 workgroup / object)   sampling + order-preserving compaction                       -> pts_a, n_a
//   xp_filter   (n / 256 workgroups)     neighbour counts                                             -> keep flags
//   xp_compact  (1 workgroup / object)   kept points in order, labels = own index                     -> pts_b, n_b, label[0]
//   xp_sweep x XP_SWEEPS (n / 256 wgs)   label[i] <- min label over {d^2 < 1}, with pointer jumping on the way in; a sweep that changes nothing ends the chain
//                                        (the later launches return at once)
//   xp_finish   (1 workgroup / object)   component sizes, the winner, its members in index order -> out (pinned host memory), n_out
#define XP_SWEEPS 16
struct XpScratch { float4* a; float4* b; uint8_t* keep; int* lab; int* ctl; };      // per object: a[CAP] | b[CAP] | keep[CAP] | lab[2][CAP] | ctl: n_a, n_b, changed[XP_SWEEPS + 1]
__device__ __forceinline__ XpScratch xp_scratch(uint8_t* pool, int obj) {
    uint8_t* p = pool + (size_t)obj * DV_XP_SCRATCH_BYTES;
    XpScratch s; s.a = (float4*)p; s.b = s.a + DV_XP_CAP; s.keep = (uint8_t*)(s.b + DV_XP_CAP); s.lab = (int*)(s.keep + DV_XP_CAP); s.ctl = s.lab + 2 * DV_XP_CAP;
    return s;
}

__global__ __launch_bounds__(XP_THREADS) void xp_detect_kernel(const DvExtraJob* __restrict__ jobs, DvExtraArgs a) {
    __shared__ int s_wsum[XP_THREADS / 64];
    const DvExtraJob j = jobs[blockIdx.x];
    const XpScratch sc = xp_scratch(a.pool, blockIdx.x);
    const int tid = threadIdx.x;
    if (tid < XP_SWEEPS + 3) sc.ctl[tid] = 0;
    // ---- DetectExtraPoints ----
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
        if (ok && pos < DV_XP_CAP) sc.a[pos] = p;
        n += total;
    }
    if (n > DV_XP_CAP) { if (tid == 0 && a.err_flag) atomicOr(a.err_flag, 8); n = DV_XP_CAP; }
    __syncthreads();
    if (tid == 0) sc.ctl[0] = n;
    if (a.stage == 1) {          // operator form, DetectExtraPoints alone (this block's own global writes are visible to it after the barrier)
        __threadfence_block();
        for (int i = tid; i < n; i += XP_THREADS) { const float4 p = sc.a[i]; double* o = j.out + 3 * (size_t)i; o[0] = (double)p.x; o[1] = (double)p.y; o[2] = (double)p.z; }
        if (tid == 0) *j.n_out = n;
    }
}

// RadiusOutlierRemoval(0.5, 10): keep[i] = at least 11 points (itself included) with d^2 <= 0.25
__global__ __launch_bounds__(256) void xp_filter_kernel(DvExtraArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xp_smem[];
    float4* A = reinterpret_cast<float4*>(xp_smem);
    const XpScratch sc = xp_scratch(a.pool, blockIdx.y);
    const int n = sc.ctl[0], i0 = blockIdx.x * 256;
    if (a.stage == 1 || i0 >= n) return;
    for (int q = threadIdx.x; q < n; q += 256) A[q] = sc.a[q];
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= n) return;
    const float4 p = A[i];
    int cnt = 0;
#pragma unroll 4
    for (int q = 0; q < n; ++q) cnt += (xp_d2(p, A[q]) <= 0.25f) ? 1 : 0;                   // (all lanes read the same A[q]: an LDS broadcast)
    sc.keep[i] = cnt >= 11 ? 1 : 0;                                                          // k = 11 nearest exist and the farthest is not beyond the radius
}

__global__ __launch_bounds__(XP_THREADS) void xp_compact_kernel(DvExtraArgs a) {
    __shared__ int s_wsum[XP_THREADS / 64];
    const XpScratch sc = xp_scratch(a.pool, blockIdx.x);
    if (a.stage == 1) return;
    const int n = sc.ctl[0], tid = threadIdx.x;
    int m = 0;
    for (int base = 0; base < n; base += XP_THREADS) {
        const int i = base + tid;
        const bool keep = i < n && sc.keep[i] != 0;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (keep) p = sc.a[i];
        int total;
        const int pos = m + xp_block_prefix(keep, s_wsum, &total);
        if (keep) { sc.b[pos] = p; sc.lab[pos] = pos; }
        m += total;
    }
    if (tid == 0) sc.ctl[1] = m >= 5 ? m : 0;                                               // fewer than 5 survivors: no extra points (dynamic_tracker.cpp:287-289)
}

// one Jacobi sweep of the min-label propagation over {d^2 < 1}: reads label set (t & 1) — each label first replaced by its label's label, twice (pointer jumping) —,
// writes set ((t + 1) & 1) for all points, raises changed[t] if any label moved.  Labels only decrease and stay inside their component; the fixed point is the
// component's lowest index.  A sweep behind a sweep that changed nothing returns at once (both sets are equal then).
__global__ __launch_bounds__(256) void xp_sweep_kernel(DvExtraArgs a, int t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xp_smem[];
    float4* B = reinterpret_cast<float4*>(xp_smem);                      // x, y, z and — in w — the point's label: one 16-byte LDS broadcast per pair
    const XpScratch sc = xp_scratch(a.pool, blockIdx.y);
    const int m = sc.ctl[1], i0 = blockIdx.x * 256;
    if (a.stage == 1 || i0 >= m) return;
    if (t > 0 && sc.ctl[2 + t - 1] == 0) return;
    const int* in = sc.lab + (size_t)(t & 1) * DV_XP_CAP; int* out = sc.lab + (size_t)((t + 1) & 1) * DV_XP_CAP;
    for (int q = threadIdx.x; q < m; q += 256) { float4 v = sc.b[q]; int l = in[q]; l = in[l]; l = in[l]; v.w = __int_as_float(l); B[q] = v; }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= m) return;
    const float4 p = B[i];
    int lmin = __float_as_int(p.w);
#pragma unroll 4
    for (int q = 0; q < m; ++q) { const float4 v = B[q]; const int l = __float_as_int(v.w); lmin = (xp_d2(p, v) < 1.0f && l < lmin) ? l : lmin; }
    out[i] = lmin;
    if (lmin != in[i]) sc.ctl[2 + t] = 1;
}

__global__ __launch_bounds__(XP_THREADS) void xp_finish_kernel(const DvExtraJob* __restrict__ jobs, DvExtraArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xp_smem[];
    int* csize = reinterpret_cast<int*>(xp_smem);
    __shared__ int s_wsum[XP_THREADS / 64];
    __shared__ unsigned long long s_best;
    const DvExtraJob j = jobs[blockIdx.x];
    const XpScratch sc = xp_scratch(a.pool, blockIdx.x);
    if (a.stage == 1) return;
    const int m = sc.ctl[1], tid = threadIdx.x;
    int result = 0;
    if (m > 0) {
        // the sweeps ended when one changed nothing; if even the last one still did, the labels are not the components' yet: flagged, never silent
        int last = 0; for (int t = 0; t < XP_SWEEPS; ++t) if (sc.ctl[2 + t]) last = t + 1;
        if (last >= XP_SWEEPS && tid == 0 && a.err_flag) atomicOr(a.err_flag, 16);
        const int* label = sc.lab + (size_t)(min(last + 1, XP_SWEEPS) & 1) * DV_XP_CAP;      // the set the last executed sweep wrote (equal to the other one once converged)
        for (int i = tid; i < m; i += XP_THREADS) csize[i] = 0;
        if (tid == 0) s_best = 0ull;
        __syncthreads();
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
                if (in) { const float4 p = sc.b[i]; double* o = j.out + 3 * (size_t)pos; o[0] = (double)p.x; o[1] = (double)p.y; o[2] = (double)p.z; }
                result += total;
            }
        }
    }
    if (tid == 0) *j.n_out = result;
}

}  // namespace

size_t dv_extra_points_scratch_bytes(int n_jobs) { return (size_t)std::max(n_jobs, 1) * DV_XP_SCRATCH_BYTES; }
int dv_launch_extra_points(const DvExtraJob* jobs_dev, int n_jobs, const DvExtraArgs& a, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    static DevOnce once;
    if (once.run([] {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(xp_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DV_XP_CAP * sizeof(float4))) != hipSuccess) return 1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(xp_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DV_XP_CAP * sizeof(float4))) != hipSuccess) return 1;
            return 0; })) return -1;
    const int nb = (DV_XP_CAP + 255) / 256;
    hipLaunchKernelGGL(xp_detect_kernel, dim3(n_jobs), dim3(XP_THREADS), 0, s, jobs_dev, a);
    if (a.stage == 1) return 0;
    hipLaunchKernelGGL(xp_filter_kernel, dim3(nb, n_jobs), dim3(256), DV_XP_CAP * sizeof(float4), s, a);
    hipLaunchKernelGGL(xp_compact_kernel, dim3(n_jobs), dim3(XP_THREADS), 0, s, a);
    for (int t = 0; t < XP_SWEEPS; ++t) hipLaunchKernelGGL(xp_sweep_kernel, dim3(nb, n_jobs), dim3(256), DV_XP_CAP * sizeof(float4), s, a, t);
    hipLaunchKernelGGL(xp_finish_kernel, dim3(n_jobs), dim3(XP_THREADS), DV_XP_CAP * sizeof(int), s, jobs_dev, a);
    return 0;
}

// InstFeat::DetectExtraPoints' sampling step (instance_feature.cpp:421-422): double arithmetic, truncated
int dv_extra_points_step(int rows, int cols) {
    const float N_max = 1000.f;
    return (int)std::max(std::sqrt(0.8 * rows * cols / N_max), 2.);
}
