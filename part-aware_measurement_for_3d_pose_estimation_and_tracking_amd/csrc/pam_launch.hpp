// Launch wrapper shared by the convolution kernels of libpam_hip.so: every kernel of the HRNet conv stack takes ONE argument struct, so a
// launch is (function, grid, block, LDS bytes, struct bytes).  Outside a recording the wrapper is hipLaunchKernelGGL; while the calling
// thread records a plan (pam_plan_begin .. pam_plan_end, csrc/pam_plan.hip) the launch is stored instead of issued.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool pam_plan_recording();
void pam_plan_add_launch(const void* func, dim3 grid, dim3 block, size_t lds, const void* arg, size_t arg_bytes);

template <typename A>
static inline void pam_launch(void (*kernel)(A), dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a) {
    if (pam_plan_recording()) pam_plan_add_launch((const void*)kernel, grid, block, lds, &a, sizeof(A));
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, a);
}
