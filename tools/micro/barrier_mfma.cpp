// Microbenchmark (development tool): cost of one s_barrier per block of NM independent MFMAs, 8 waves per workgroup, one workgroup per CU.
// build: hipcc -O3 --offload-arch=gfx950 barrier_mfma.cpp -o /tmp/barrier_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NM, int MODE>   // MODE 0: no barrier, 1: s_barrier per block, 2: barrier + 8 ds_read_b128 per block, 3: ds_reads only
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    f32x4 acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane * 2 + i); }
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((int*)lds)[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 2) asm volatile("s_barrier" ::: "memory");
        if (MODE >= 2) {
            typedef __attribute__((ext_vector_type(8))) short s8;
            s8 r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = *(const s8*)(lds + ((threadIdx.x * 96 + q * 4096 + it * 16) & 65520));
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = (__bf16)(float)(r[q][0] & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NM, int MODE> void run(const char* name, int threads = 512) {
    float* out; unsigned long long* cyc; const int nb = 256, iters = 20000;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, MODE>), dim3(nb), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, MODE>), dim3(nb), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb); hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= nb;
    const double flop = (double)nb * (threads / 64) * NM * 16384.0 * iters;
    printf("%-30s NM=%2d thr=%3d: %7.1f ticks/block  wall %7.3f ms  tick rate %.2f GHz  %.0f TFLOP/s\n", name, NM, threads, m / iters, ms, m / (ms * 1e6), flop / (ms * 1e9));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<18, 0>("no barrier"); run<18, 1>("s_barrier per block"); run<18, 0>("no barrier", 256); run<18, 1>("s_barrier per block", 256);
    run<18, 0>("no barrier", 1024); run<18, 1>("s_barrier per block", 1024);
    run<36, 1>("s_barrier per block"); run<8, 1>("s_barrier per block");
    return 0;
}
