// copy.hip — the per-frame staging copies as KERNELS on the stream that needs them, not as hipMemcpyAsync.
//
// Every upload of the per-frame path (the window solve's packed tables, the object / line solves' problems, the instance tracker's job arena and mask staging, the
// job tables of the shared launches) starts in a pinned host buffer the library owns, and every per-frame download ends in one.  hipMemcpyAsync moves those through
// the SDMA copy engines: ~18 us per call for ~100 KB, and — what matters for a real-time estimator — the runtime brings its copy queues up LAZILY, the first time
// two host threads' copies collide: one 6 - 7 ms frame somewhere in the first seconds of a dynamic sequence (T2 tracker thread beside T3 estimator thread, the
// reference's system/main.cpp:178-330 layout).  Pinned host memory is mapped into the device's address space: a kernel on the consumer's own stream
// reads it over PCIe directly (16 B per lane, four loads in flight per lane), stream order does the rest, and no copy engine exists in the per-frame path any more.
// Copies from / to CALLER memory (frames, masks handed over as DV_MEM_HOST) stay hipMemcpy: that memory may be pageable.
#include "dv_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {
constexpr int CP_THREADS = 256, CP_UNROLL = 4;

__global__ __launch_bounds__(CP_THREADS) void dv_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, size_t bytes) {
    const size_t base = ((size_t)blockIdx.x * CP_THREADS * CP_UNROLL) + threadIdx.x;
    uint4 v[CP_UNROLL];
#pragma unroll
    for (int u = 0; u < CP_UNROLL; ++u) { const size_t i = base + (size_t)u * CP_THREADS; if (i < n16) v[u] = src[i]; }
#pragma unroll
    for (int u = 0; u < CP_UNROLL; ++u) { const size_t i = base + (size_t)u * CP_THREADS; if (i < n16) dst[i] = v[u]; }
    if (blockIdx.x == 0) {          // the tail that is not a multiple of 16 bytes
        const size_t t = n16 * 16 + threadIdx.x;
        if (t < bytes) reinterpret_cast<uint8_t*>(dst)[t] = reinterpret_cast<const uint8_t*>(src)[t];
    }
}
__global__ __launch_bounds__(CP_THREADS) void dv_copy_bytes_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t bytes) {
    for (size_t i = (size_t)blockIdx.x * CP_THREADS + threadIdx.x; i < bytes; i += (size_t)gridDim.x * CP_THREADS) dst[i] = src[i];
}
}  // namespace

// dst / src: device memory or pinned host memory (hipHostMalloc), either direction.  Stream-ordered like hipMemcpyAsync; the caller keeps the source alive and
// unchanged until the stream has passed the copy — the rule the hipMemcpyAsync calls it replaces already imposed.
// DVINS_COPY_ENGINE=1 (environment, read once) keeps the copy engines in the path: the A/B switch of the measurement in DESIGN.md 5.
hipError_t dv_copy_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    static const bool engine = [] { const char* e = std::getenv("DVINS_COPY_ENGINE"); return e && e[0] == '1'; }();
    if (engine) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s);
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const size_t per = (size_t)CP_THREADS * CP_UNROLL;
        const unsigned grid = (unsigned)std::max<size_t>((n16 + per - 1) / per, 1);
        hipLaunchKernelGGL(dv_copy_kernel, dim3(grid), dim3(CP_THREADS), 0, s, (uint4*)dst, (const uint4*)src, n16, bytes);
    } else {
        const unsigned grid = (unsigned)std::min<size_t>((bytes + CP_THREADS - 1) / CP_THREADS, 1024);
        hipLaunchKernelGGL(dv_copy_bytes_kernel, dim3(grid), dim3(CP_THREADS), 0, s, (uint8_t*)dst, (const uint8_t*)src, bytes);
    }
    return hipGetLastError();
}

int dv_copy_prepare() { hipFuncAttributes fa; return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(dv_copy_kernel)) == hipSuccess ? 0 : -1; }

// FeatureTrack's static-instance unmasking (system/main.cpp:217-245): the ROI mask lives in pinned host memory and is read in place
__global__ __launch_bounds__(256) void dv_unmask_kernel(uint8_t* __restrict__ inv_mask, int pitch, int W, int H, int x0, int y0, int w, int h, const uint8_t* __restrict__ roi) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= w || r >= h) return;
    const int x = x0 + c, y = y0 + r;
    if (x < W && y < H && roi[(size_t)r * w + c] >= 1) inv_mask[(size_t)y * pitch + x] = 255;
}
void dv_launch_unmask(uint8_t* inv_mask, int pitch, int W, int H, int x0, int y0, int w, int h, const uint8_t* roi_mask, hipStream_t s) {
    if (w > 0 && h > 0) hipLaunchKernelGGL(dv_unmask_kernel, dim3((w + 255) / 256, h), dim3(256), 0, s, inv_mask, pitch, W, H, x0, y0, w, h, roi_mask);
}

// A queue's first dispatch of a kernel that needs scratch (private-segment) memory makes the runtime allocate that memory for the queue — a device allocation plus a queue
// reconfiguration, 1 - 2 ms when the device is busy or its memory is held by other processes — and a later kernel with a LARGER per-lane need repeats it.  be_solve (28 B per
// lane), the gauge kernels (80 B) and the batched solve (132 B) would pay it in the middle of a sequence: at the first window solve (seen as a 3.7 ms frame from a cold process
// beside a busy session, tests/test_frame_gaps.py).  One trivial kernel with ~150 B per lane per stream at create time pays it there.
__global__ void dv_scratch_warm_kernel(int* out, int n) {
    int a[36];          // 144 B per lane (+ what the compiler adds): at least what any kernel of the per-frame path needs (batched solve 132 B); NOT much more — the runtime treats very
                        // large scratch requests as use-once and frees them behind the dispatch
    for (int i = 0; i < 36; ++i) a[i] = i * n + (int)threadIdx.x;
    int acc = 0;
    for (int i = 0; i < 36; ++i) acc += a[(i * 7 + n + (int)threadIdx.x) % 36];      // dynamic indexing keeps the array in scratch
    if (n == -12345) *out = acc;
}
int dv_warm_stream(hipStream_t s) {
    hipLaunchKernelGGL(dv_scratch_warm_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr, 3);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
