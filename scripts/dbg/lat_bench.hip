// micro-benchmark (debug): dependent-issue latency of fp64 VALU ops on one wave of gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, double x0, int nw) {
    double x = x0 + threadIdx.x * 1e-9, y = 1.0000001;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) x = __builtin_fma(x, y, 1e-9);
    }
    long long t1 = clock64();
    double r = x;
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) r = __builtin_amdgcn_rcp(r) + 1.5;
    }
    long long t2 = clock64();
    double a = x, b = x + 1, c = x + 2, d = x + 3;
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) { a = __builtin_fma(a, y, 1e-9); b = __builtin_fma(b, y, 1e-9); c = __builtin_fma(c, y, 1e-9); d = __builtin_fma(d, y, 1e-9); }
    }
    long long t3 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + r + a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 64);
    for (int threads : {64, 256, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0, 0);
        long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
        printf("threads %4d: dependent fma %.1f cycles/op, dependent rcp+add %.1f cycles/pair, 4 independent fma chains %.1f cycles per 4 ops\n", threads, h[0] / 1000.0, h[1] / 1000.0, h[2] / 1000.0);
    }
    return 0;
}
