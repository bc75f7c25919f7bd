// debug: which XCDs / CUs a stream created with hipExtStreamCreateWithCUMask lands on (bit layout of the mask on gfx950), and what a masked stream costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void probe(unsigned* out) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));       // HW_REG_XCC_ID, bits [3:0]
    unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));      // HW_REG_HW_ID
    for (volatile int i = 0; i < 20000; ++i) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned* d; hipMalloc(&d, 8192 * 8);
    auto run = [&](const char* name, std::vector<uint32_t> mask) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); return; }
        hipMemsetAsync(d, 0xff, 8192 * 8, s);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(8192); hipMemcpy(h.data(), d, 8192 * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> cus;
        for (int b = 0; b < 4096; ++b) { unsigned hw = h[2 * b + 1]; cus[h[2 * b]].insert((hw >> 8) & 0xf | ((hw >> 12) & 0x1) << 4 | ((hw >> 13) & 0x7) << 5); }      // cu_id [11:8], sh_id [12], se_id [15:13]
        printf("%s:", name); for (auto& kv : cus) printf(" xcc%u:%zu", kv.first, kv.second.size()); printf("\n");
        hipStreamDestroy(s);
    };
    run("all 256", std::vector<uint32_t>(8, 0xffffffffu));
    run("bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0-63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
    run("bits 64-127", {0, 0, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("every 8th bit (0,8,..)", std::vector<uint32_t>(8, 0x01010101u));
    run("bits = 0,1 mod 8", std::vector<uint32_t>(8, 0x03030303u));
    run("bits = 0..3 mod 8", std::vector<uint32_t>(8, 0x0f0f0f0fu));
    return 0;
}
