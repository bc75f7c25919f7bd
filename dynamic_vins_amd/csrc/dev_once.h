// dev_once.h — "once per HIP device" guard.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only,
// and one process may hold ctxs on several GPUs (dv_config.device) driven from several host threads: the opt-in is therefore
// tracked per device ordinal under a mutex (a process-wide `static bool` set it on the first device only).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

struct DevOnce {
    std::mutex mu; bool done[64] = { false };
    template <class F> int run(F&& f) {          // f() returns 0 on success; runs at most once per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
        std::lock_guard<std::mutex> lk(mu);
        if (done[dev]) return 0;
        if (f()) return -1;
        done[dev] = true;
        return 0;
    }
};
