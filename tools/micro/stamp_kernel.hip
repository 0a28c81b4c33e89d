// Development tool (tools/fwd_stamps.py): a one-wave kernel that stores the constant-rate clock (s_memrealtime, 100 MHz) into a slot of a
// buffer of its own.  Captured into the forward's hipGraph at chosen points of its streams it gives an UN-profiled timeline (rocprofv3's
// kernel trace slows and serialises the submission of a replay).
#include <hip/hip_runtime.h>
__global__ void k_stamp(unsigned long long* buf, int idx) { if (threadIdx.x == 0) buf[idx] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int stamp_launch(void* stream, void* buf, int idx) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)buf, idx);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
