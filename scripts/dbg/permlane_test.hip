#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    unsigned v = threadIdx.x;                 // lane id
    u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    unsigned a = r[0], b = r[1];
    u2 s = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    u2 t = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    out[threadIdx.x] = s[0]; out[64 + threadIdx.x] = s[1]; out[128 + threadIdx.x] = t[0]; out[192 + threadIdx.x] = t[1];
    out[256 + threadIdx.x] = a; out[320 + threadIdx.x] = b;
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"s0", "s1", "t0", "t1", "a", "b"};
    for (int q = 0; q < 6; ++q) { printf("%s:", names[q]); for (int i = 0; i < 64; i += 4) printf(" %u", h[64 * q + i]); printf("\n"); }
    return 0;
}
