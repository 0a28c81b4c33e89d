// libpam_hip.so, scheduling part of a1 (round 5): device-side ordering of the HRNet forward's branch streams.
// An HR module ends with a full exchange: every output's sum reads every branch (hrnet.py:64-100; stands inside the absent HRNet backend
// behind /root/reference/src/ivclabpose.py:210).  As stream events inside a captured hipGraph that join costs 15-21 us of idle chip per
// module on ROCm 7.2 (two cross-queue hops through the caller's stream: pairwise event waits between the branch streams crash the
// capture) -- ten times per forward.  Here the branch streams stay independent chains of the graph and meet through a counter in device
// memory: a branch's chain ends with k_flag_signal (or, where the stream goes on with a sum, with the gate's own arrival) (the preceding kernels of its stream have completed and released their
// stores at the kernel boundary; one agent-scope atomic add), the first launch in front of a sum is k_flag_gate (ONE wave polls the
// counter with agent-scope relaxed loads + s_sleep until every branch has signalled; the sum that follows on the same stream starts with
// the usual kernel-boundary acquire).  The poll is BOUNDED: past max_us (2 s in the executor: far beyond any stall of a healthy run,
// e.g. the context switches of two processes sharing a device) the gate sets *err and lets the stream go on -- a mapping of two chains
// onto one in-order hardware queue, or a profiler that serialises kernels, cannot hang the device; the host checks err after the first
// replay of every capture and falls back to stream events (pam/hrnet.py).  A time-out also stores 1 to *host_err when the caller gave
// one (a word of pinned host memory: the host reads it before every later replay at no cost to the device), so a time-out in ANY replay
// is reported -- e.g. two flagged replays in flight at the same time can block each other's queues, which no check of one replay sees.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"

__global__ __launch_bounds__(64) void k_flag_signal(int* counter) {
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// several counters in one launch (round 6: one counter per OUTPUT of a module -- a sum waits for the branches' blocks and the chains it
// reads, not for every branch's whole tail): lane k adds to counters[k] where bit k of mask is set; ONE release for all of them
__global__ __launch_bounds__(64) void k_flag_signal_mask(int* counters, unsigned mask) {
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x < 32 && ((mask >> threadIdx.x) & 1u))
        __hip_atomic_fetch_add(counters + threadIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void k_flag_gate(int* counter, int target, int* err, const unsigned* max_us_p, int arrive, int* host_err, int* dev_void) {
    if (threadIdx.x == 0) {
        if (arrive) {                                     // arrive-and-wait: this stream's own contribution, then everybody else's
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned max_us = *max_us_p;                                      // a device word: the host can change the bound of captured gates
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(16);
            // a gate of this forward has already given up: the forward's result is void anyway, do not add another time-out to it
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)max_us * 100ull) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the consumers of this forward's keypoints on the DEVICE (the frame kernel: pam_set_input_guard) must not use them either
                if (dev_void) __hip_atomic_store(dev_void, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (host_err) __hip_atomic_store(host_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

extern "C" int pam_flag_signal(void* stream, int32_t* dev_counter) {
    if (!dev_counter) return PAM_E_ARG;
    hipLaunchKernelGGL(k_flag_signal, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_counter);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
extern "C" int pam_flag_signal_mask(void* stream, int32_t* dev_counters, uint32_t mask) {
    if (!dev_counters || !mask) return PAM_E_ARG;
    hipLaunchKernelGGL(k_flag_signal_mask, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_counters, mask);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
extern "C" int pam_flag_gate(void* stream, int32_t* dev_counter, int target, int32_t* dev_err, const uint32_t* dev_max_us, int arrive, int32_t* host_err,
                             int32_t* dev_void) {
    if (!dev_counter || !dev_err || target < 1 || !dev_max_us) return PAM_E_ARG;
    hipLaunchKernelGGL(k_flag_gate, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_counter, target, dev_err, (const unsigned*)dev_max_us, arrive ? 1 : 0, host_err,
                       dev_void);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
