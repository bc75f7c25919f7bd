// debug: shader clock of a lightly loaded gfx950 — s_memtime (core clock) against s_memrealtime (100 MHz) around a dependent chain; one wave, then 256 x 4 waves
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* ts, int n) {
    double x = threadIdx.x * 1e-3, y = 1.0 - x;
    int v = threadIdx.x;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) { x = __builtin_fma(x, y, 0.5); }
    long long c1 = clock64(), w1 = wall_clock64();
    for (int i = 0; i < n; ++i) { v = v * 3 + 1; asm volatile("" : "+v"(v)); }
    long long c2 = clock64(), w2 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts[0] = w1 - w0; ts[1] = c1 - c0; ts[2] = w2 - w1; ts[3] = c2 - c1; }
    out[threadIdx.x] = x + v;
}
int main() {
    double* out; long long* ts; hipMalloc(&out, 8192); hipMalloc(&ts, 64);
    for (int blocks : { 1, 256, 1024 }) for (int rep = 0; rep < 3; ++rep) {
        long long h[4];
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, ts, 4096);
        hipMemcpy(h, ts, 32, hipMemcpyDeviceToHost);
        printf("blocks %4d: fma chain %.1f ns/op, %.2f core clocks/op -> %.0f MHz; int chain %.1f ns/op, %.2f clocks/op\n", blocks, h[0] * 10.0 / 4096, (double)h[1] / 4096, h[1] / (h[0] * 10e-3), h[2] * 10.0 / 4096, (double)h[3] / 4096);
    }
    return 0;
}
