// debug: time per dependent launch of a chain of tiny kernels — plain stream launches against the same chain replayed as a hipGraph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(double* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] * 1.0000001 + k; }
__global__ void wide(double* p, int k) { const int i = blockIdx.x * 256 + threadIdx.x; p[i] = p[i] * 1.0000001 + k; }
int main() {
    double* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 300;
    for (int variant = 0; variant < 2; ++variant) {
        auto body = [&](hipStream_t st) { for (int k = 0; k < N; ++k) { if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, k); else hipLaunchKernelGGL(wide, dim3(272), dim3(256), 0, st, d, k); } };
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            body(s); hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("variant %d stream: %.2f us per launch\n", variant, us / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); body(s); hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("variant %d graph : %.2f us per launch\n", variant, us / N);
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
