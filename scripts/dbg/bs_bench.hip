// micro-benchmark (debug): what one pivot of the back substitution costs on one wave of gfx950
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/dbg/bs_bench.hip -o /tmp/bs_bench && /tmp/bs_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double lane_bcast(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }
template <int SEG, int ROWS>
__device__ __forceinline__ void bs_chunk(const double* Lm, int kt, int lane, double& x0, double& x1, double& x2) {
    double cr[ROWS][3];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        const int k = kt - u;
        const double* row = Lm + tri(k, 0) + lane;
        cr[u][0] = row[0];
        if (SEG >= 1) cr[u][1] = row[64];
        if (SEG >= 2) cr[u][2] = row[128];
        cr[u][SEG] = lane < k - 64 * SEG ? cr[u][SEG] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        const int k = kt - u;
        const double xs = SEG == 0 ? x0 : (SEG == 1 ? x1 : x2);
        const double xk = lane_bcast(xs, k - 64 * SEG);
        x0 = __builtin_fma(-cr[u][0], xk, x0);
        if (SEG >= 1) x1 = __builtin_fma(-cr[u][1], xk, x1);
        if (SEG >= 2) x2 = __builtin_fma(-cr[u][2], xk, x2);
    }
}
template <int SEG>
__device__ __forceinline__ void ld8(const double* Lm, int kt, int lane, double (&cr)[8][3]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kt - u;
        const double* row = Lm + tri(k, 0) + lane;
        cr[u][0] = row[0];
        if (SEG >= 1) cr[u][1] = row[64];
        if (SEG >= 2) cr[u][2] = row[128];
    }
}
template <int SEG>
__device__ __forceinline__ void ap8(const double (&cr)[8][3], int kt, int lane, double& x0, double& x1, double& x2) {
    double m[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) m[u] = lane < kt - u - 64 * SEG ? cr[u][SEG] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const double xs = SEG == 0 ? x0 : (SEG == 1 ? x1 : x2);
        const double xk = lane_bcast(xs, kt - u - 64 * SEG);
        x0 = __builtin_fma(-(SEG == 0 ? m[u] : cr[u][0]), xk, x0);
        if (SEG >= 1) x1 = __builtin_fma(-(SEG == 1 ? m[u] : cr[u][1]), xk, x1);
        if (SEG >= 2) x2 = __builtin_fma(-m[u], xk, x2);
    }
}
template <int SEG>
__device__ __forceinline__ int seg_pipe(const double* Lm, int k, int lane, double& x0, double& x1, double& x2) {
    const int kend = 64 * SEG + 7;
    double ca[8][3], cb[8][3];
    ld8<SEG>(Lm, k, lane, ca);
    for (; k >= kend; k -= 16) {
        if (k - 8 >= kend) ld8<SEG>(Lm, k - 8, lane, cb);
        __builtin_amdgcn_sched_barrier(0); ap8<SEG>(ca, k, lane, x0, x1, x2); __builtin_amdgcn_sched_barrier(0);
        if (k - 8 >= kend) {
            if (k - 16 >= kend) ld8<SEG>(Lm, k - 16, lane, ca);
            __builtin_amdgcn_sched_barrier(0); ap8<SEG>(cb, k - 8, lane, x0, x1, x2); __builtin_amdgcn_sched_barrier(0);
        }
    }
    return 64 * SEG - 1;
}
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int nn, int busy_waves) {
    extern __shared__ double Lm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < nn * (nn + 1) / 2; i += blockDim.x) Lm[i] = 1e-3 * ((i * 7) % 13 - 6);
    __syncthreads();
    double x0 = 1.0 + lane * 1e-3, x1 = 2.0 + lane * 1e-3, x2 = 3.0 + lane * 1e-3;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    if (tid < 64) {
        // (1) the bare hand-off chain: fma -> readlane x2 -> fma
        t0 = clock64();
        double x = x0;
#pragma unroll 1
        for (int i = 0; i < 100; ++i) {
#pragma unroll
            for (int j = 0; j < 10; ++j) { const double xk = lane_bcast(x, (i + j) & 63); x = __builtin_fma(-1e-3, xk, x); }
        }
        t1 = clock64();
        x0 += x * 1e-9;
        // (2) the kernel's loop
        int kk = nn - 1;
        for (; kk >= 0 && ((kk + 1) & 7); --kk) { const int seg = kk >> 6; if (seg == 0) bs_chunk<0, 1>(Lm, kk, lane, x0, x1, x2); else if (seg == 1) bs_chunk<1, 1>(Lm, kk, lane, x0, x1, x2); else bs_chunk<2, 1>(Lm, kk, lane, x0, x1, x2); }
        t2 = clock64();
        for (; kk >= 7; kk -= 8) { const int seg = kk >> 6; if (seg == 0) bs_chunk<0, 8>(Lm, kk, lane, x0, x1, x2); else if (seg == 1) bs_chunk<1, 8>(Lm, kk, lane, x0, x1, x2); else bs_chunk<2, 8>(Lm, kk, lane, x0, x1, x2); }
        t3 = clock64();
        // (3) loads only
        double acc = 0;
        for (kk = 159; kk >= 7; kk -= 8) { double cr[8][3]; const int seg = kk >> 6; if (seg == 0) ld8<0>(Lm, kk, lane, cr); else if (seg == 1) ld8<1>(Lm, kk, lane, cr); else ld8<2>(Lm, kk, lane, cr);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cr[u][0] + (seg >= 1 ? cr[u][1] : 0.0) + (seg >= 2 ? cr[u][2] : 0.0); }
        t4 = clock64();
        x0 += acc * 1e-12;
        // (4) pipelined per segment
        kk = nn - 1 - ((nn) & 7);
        if (kk >= 135) kk = seg_pipe<2>(Lm, kk, lane, x0, x1, x2);
        if (kk >= 71) kk = seg_pipe<1>(Lm, kk, lane, x0, x1, x2);
        if (kk >= 7) kk = seg_pipe<0>(Lm, kk, lane, x0, x1, x2);
        t5 = clock64();
    }
    __syncthreads();
    out[tid] = x0 + x1 + x2;
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 64);
    const int nn = 165;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    for (int threads : {64, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), nn * (nn + 1) / 2 * 8, 0, out, cyc, nn, 0);
        long long h[5]; hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
        printf("threads %4d: hand-off chain %.1f cycles/pivot; head (%d single rows) %lld cycles; 8-row chunks: %lld cycles = %.1f per pivot\n", threads, h[0] / 1000.0, (nn & 7), h[1], h[2], h[2] / (double)(nn - (nn & 7)));
        printf("              loads + masks + sum only: %lld cycles; pipelined per segment: %lld cycles\n", h[3], h[4]);
    }
    return 0;
}
