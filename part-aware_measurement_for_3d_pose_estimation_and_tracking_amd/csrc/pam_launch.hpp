// Launch helpers shared by the convolution kernels of libpam_hip.so: every kernel of the HRNet conv stack takes ONE argument struct, so a
// launch is (function, grid, block, LDS bytes, struct).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <mutex>
#include <unordered_map>

template <typename A>
static inline void pam_launch(void (*kernel)(A), dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a) {
    hipLaunchKernelGGL(kernel, grid, block, lds, s, a);
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a DEVICE's copy of a function: set it once per (function, device) -- a process that
// drives a second GPU would otherwise launch with > 64 KB of LDS without the attribute and fail.  Returns false on a HIP error.
static inline bool pam_max_dynamic_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;      // function -> bit per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> g(mu);
    unsigned long long& m = done[func];
    if (dev < 64 && ((m >> dev) & 1)) return true;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (dev < 64) m |= 1ull << dev;
    return true;
}
