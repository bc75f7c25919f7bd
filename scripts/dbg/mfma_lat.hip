// debug: latency of dependent v_mfma_f64_16x16x4_f64 chains and of LDS-fed ones on gfx950.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lat scripts/dbg/mfma_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* ts, int n, int mode) {
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1e-3 * i;
    __syncthreads();
    d4 acc = { 0, 0, 0, 0 }, acc2 = { 0, 0, 0, 0 };
    double x = threadIdx.x * 1e-3, y = 1.0 - x;
    long long t0 = wall_clock64();
    if (mode == 0) for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    else if (mode == 1) for (int i = 0; i < n; i += 2) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, acc2, 0, 0, 0); }
    else if (mode == 2) for (int i = 0; i < n; ++i) { const double a = lds[(threadIdx.x & 63) + 108 * (i & 63)], b = lds[(threadIdx.x & 63) + 54 + 108 * (i & 63)]; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }
    else for (int i = 0; i < n; ++i) { x = __builtin_fma(x, y, 0.5); }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ts[0] = t1 - t0; }
    out[threadIdx.x] = acc[0] + acc[1] + acc2[2] + x;
}
int main() {
    double* out; long long* ts; hipMalloc(&out, 8192); hipMalloc(&ts, 64);
    for (int mode = 0; mode < 4; ++mode) for (int threads : { 64, 256, 512 }) {
        long long h = 0;
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, ts, 256, mode); hipMemcpy(&h, ts, 8, hipMemcpyDeviceToHost); }
        printf("mode %d threads %d: %.1f ns per op\n", mode, threads, h * 10.0 / 256);
    }
    return 0;
}
