// libpam_hip.so, fuse-layer part of a1 (round 5): one output of an HR module's fuse layer in ONE launch
//     out_i = ReLU( x_i + sum_{j < i} down_ij + sum_{j > i} nearest_up_{2^(j-i)}( W_ij x_j + b_ij ) )
// (hrnet.py:64-100; stands inside the absent HRNet backend behind /root/reference/src/ivclabpose.py:210).
// down_ij = the result of the strided-convolution chain from the finer branch j (already at out_i's resolution: a "plain" term);
// the coarser branches j > i enter through a 1x1 convolution at THEIR resolution, up-sampled by pixel replication.
//
// Until round 5 the 1x1 products were launches of their own (18 per forward on the generic gather kernel: 2 % of the MFMA roof, 12 % of
// the HBM roof, each on some branch's tail in front of the module's join) whose results went to HBM and came back into k_upsample_add.
// k_fuse_sum: PERSISTENT workgroups (one per CU at most; eight worker waves + one loader wave), the 1x1 weights of ALL sources resident in LDS for the whole launch (host-packed
// MFMA A fragments, 1 KiB each, fetched once by LDS-DMA: 64 KB for output 0 of a four-branch module, 144 KB for output 2).  A workgroup
// walks tiles of out_i that are a whole number of pixels of the COARSEST source (TA x TB of them, 1 x 3 by default = 8 x 24 pixels of
// output 0): the source pixels under tile t + 1 arrive by LDS-DMA (second buffer) while tile t is processed; the products of tile t run
// out of LDS (accumulators start from the bias, one bf16 rounding: exactly the arithmetic of the convolution launches they replace) into a
// few KB of term tiles; then the tile's base + plain terms (requested before the products) + the replicated LDS terms -> ReLU -> out,
// 16 bytes per lane, in k_upsample_add's summation order.  Nothing but base, plain terms, sources and out touches HBM.
// Bit-identical to pam_conv2d_nhwc_bf16 (1x1) + pam_upsample_add_nhwc_bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"
#include "pam_launch.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int FS_MAXI = 5;                 // 16-byte items of the output tile per thread at most

struct FSArgs {
    const uint16_t* base; uint16_t* out;
    const uint16_t* plain[2]; int plain_cs[2]; int nplain;
    const uint16_t* src[3]; const char* wimg[3]; const float* ubias[3]; int shift[3]; int src_c[3]; int nup;
    int N, H, W, relu, TA, TB, tiles_y, tiles_x, ntiles, OTH, OTW, smax;
    int w_off[3], w_bytes[3];               // LDS: the sources' weight images
    int bias_off;                           // float32 [nup][C]
    int src_off[3], src_bytes;              // a source-pixel buffer: [source][tile row][tile column][Cs]; two of them from src_base on
    int src_base, term_off[3];              // term tiles [source][pixel][C]
    float inv_otw;
};

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }   // exact for x < 2^16

template <int C>
__global__ __launch_bounds__(576) void k_fuse_sum(FSArgs a) {
    constexpr int C8 = C / 8, NT = C / 16;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_img = a.tiles_y * a.tiles_x;

    // ---- once per launch: every source's weight fragments and biases -> LDS
    for (int s = 0; s < a.nup; ++s) {
        {
            const char* wsrc = a.wimg[s] + lane * 16;
            for (int p = wave; p * 1024 < a.w_bytes[s]; p += 9)
                __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + p * 1024), (lds_void*)(smem + a.w_off[s] + p * 1024), 16, 0, 0);
            float* bs = (float*)(smem + a.bias_off) + s * C;
            if (tid < C) bs[tid] = a.ubias[s] ? a.ubias[s][tid] : 0.0f;            // C <= 192 < 512
        }
    }
    // source pixels under tile T -> buffer b, row by row (a tile row of a source is contiguous in the image), pieces of 1 KiB: all of them
    // by the LOADER wave (wave 8), which does nothing else -- the eight worker waves never wait for a DMA or for their own stores
    auto src_dma = [&](int T, int b) {
        const int n = T / per_img, trem = T - n * per_img, ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        char* dst0 = smem + a.src_base + b * a.src_bytes;
        for (int s = 0; s < a.nup; ++s) {
            {
                const int sh = a.shift[s], d = a.smax - sh, th = a.TA << d, tw = a.TB << d, Cs = a.src_c[s];
                const int Hs = a.H >> sh, Ws = a.W >> sh, sy0 = (ty * a.TA) << d, sx0 = (tx * a.TB) << d;
                const int rowb = min(tw, Ws - sx0) * Cs * 2, pitch = tw * Cs * 2, npc = (rowb + 1023) >> 10;
                for (int r = 0; r < th && sy0 + r < Hs; ++r)
                    for (int k = 0; k < npc; ++k)
                        if (k * 1024 + lane * 16 < rowb)
                            __builtin_amdgcn_global_load_lds((glb_void*)((const char*)a.src[s] + ((((size_t)n * Hs + sy0 + r) * Ws + sx0) * Cs) * 2 + k * 1024 + lane * 16),
                                                             (lds_void*)(dst0 + a.src_off[s] + r * pitch + k * 1024), 16, 0, 0);
            }
        }
    };
    // the thread's items of a tile (the same positions in every tile): (tile row << 16) | (tile column << 8) | 8-channel group
    const int nitem = a.OTH * a.OTW * C8;
    int trc[FS_MAXI];
#pragma unroll
    for (int k = 0; k < FS_MAXI; ++k) {
        const int e = tid + 512 * k;
        const int pix = e / C8, c8 = e - pix * C8;
        const int r = fdiv_small(pix, a.inv_otw), c = pix - r * a.OTW;
        trc[k] = (tid < 512 && e < nitem) ? (r << 16) | (c << 8) | c8 : -1;
    }
    // jobs of a tile = (source, 16-pixel tile, 16-channel tile), coarsest (deepest K) source first; wave w takes jobs w, w + 8, ...
    int job0[4];
    {
        int acc_j = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            job0[q] = acc_j;
            if (q < a.nup) {
                const int d = a.smax - a.shift[a.nup - 1 - q];
                acc_j += ((((a.TA << d) * (a.TB << d)) + 15) >> 4) * NT;
            }
        }
        job0[3] = acc_j;
    }

    int T = (int)blockIdx.x;
    if (wave == 8) {
        // ---- loader: tile t + 1's source pixels go into the other buffer while the workers are on tile t (its last readers, the
        // products of tile t - 1, finished before the barrier at the top of tile t)
        if (T < a.ntiles) src_dma(T, 0);
        for (int it = 0; T < a.ntiles; T += (int)gridDim.x, ++it) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");             // top: tile t's pixels (and, first, my share of the weights) have landed
            if (T + (int)gridDim.x < a.ntiles) src_dma(T + (int)gridDim.x, (it + 1) & 1);
            asm volatile("s_barrier" ::: "memory");                                      // mid
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    // my share of the weights has landed
    for (int it = 0; T < a.ntiles; T += (int)gridDim.x, ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // top: this tile's source pixels are in LDS; the term tiles are free
        const int n = T / per_img, trem = T - n * per_img, ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int oy0 = ty * a.OTH, ox0 = tx * a.OTW;
        // ---- base and plain terms of the thread's items: requested now, summed behind the products
        bf16x8 vb[FS_MAXI], vp[2][FS_MAXI];
        int eo[FS_MAXI];                                  // pixel index of the item in the output tensor (-1: none); the host checks N H W < 2^31
#pragma unroll
        for (int k = 0; k < FS_MAXI; ++k) {
            const int r = trc[k] >> 16, c = (trc[k] >> 8) & 255;
            const int oy = oy0 + r, ox = ox0 + c;
            const bool ok = trc[k] >= 0 && oy < a.H && ox < a.W;
            const int p = (n * a.H + oy) * a.W + ox;
            eo[k] = ok ? p : -1;
            const long pz = ok ? p : 0;
            const int c8 = ok ? trc[k] & 255 : 0;
            vb[k] = *(const bf16x8*)(a.base + pz * C + c8 * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                vp[t][k] = t < a.nplain ? *(const bf16x8*)(a.plain[t] + pz * a.plain_cs[t] + c8 * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        // ---- the 1x1 products of the source pixels under the tile, out of LDS -> term tiles
        const char* sbuf = smem + a.src_base + (it & 1) * a.src_bytes;
        for (int job = wave; job < job0[3]; job += 8) {
            const int q = job >= job0[2] ? 2 : (job >= job0[1] ? 1 : 0);
            const int s = a.nup - 1 - q, jj = job - job0[q];
            const int mt = jj / NT, j = jj - mt * NT;
            const int d = a.smax - a.shift[s], npx = (a.TA << d) * (a.TB << d);
            const int Cs = a.src_c[s], KS = Cs >> 5;
            const int p = mt * 16 + l15, pq = min(p, npx - 1);
            const char* xp = sbuf + a.src_off[s] + pq * Cs * 2 + g * 16;
            const char* wp = smem + a.w_off[s] + (j * KS * 64 + lane) * 16;
            f32x4 acc = *(const f32x4*)(smem + a.bias_off + (s * C + 16 * j + 4 * g) * 4);
            for (int k0 = 0; k0 < KS; k0 += 3) {                                          // three k-steps' fragments read together, then multiplied
                bf16x8 fa[3], fb[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int ks = min(k0 + u, KS - 1);
                    fa[u] = *(const bf16x8*)(wp + ks * 1024); fb[u] = *(const bf16x8*)(xp + ks * 64);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (k0 + u < KS)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[u]), __builtin_bit_cast(bf16x8_t, fb[u]), acc, 0, 0, 0);
            }
            if (p < npx)
                *(u32x2*)(smem + a.term_off[s] + (p * C + 16 * j + 4 * g) * 2) = (u32x2){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                    // the term tiles are complete

        // ---- out = [ReLU](base + plain terms + replicated products), k_upsample_add's order: base, then the terms in branch order
#pragma unroll
        for (int k = 0; k < FS_MAXI; ++k) {
            if (eo[k] < 0) continue;
            const int r = trc[k] >> 16, c = (trc[k] >> 8) & 255, c8 = trc[k] & 255;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f((uint16_t)vb[k][e]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (t < a.nplain) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f((uint16_t)vp[t][k][e]);
                }
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (s < a.nup) {
                    const int sh = a.shift[s], tw = a.TB << (a.smax - sh);
                    const bf16x8 q = *(const bf16x8*)(smem + a.term_off[s] + (((r >> sh) * tw + (c >> sh)) * C + c8 * 8) * 2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f((uint16_t)q[e]);
                }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(a.relu ? fmaxf(v[e], 0.0f) : v[e]);
            *(bf16x8*)(a.out + (long)eo[k] * C + c8 * 8) = o;
        }
    }
}

template <int C>
int launch_fs(hipStream_t s, const FSArgs& a, size_t lds, int grid) {
    if (!pam_max_dynamic_lds((const void*)k_fuse_sum<C>, 160 * 1024)) return PAM_E_HIP;
    pam_launch(k_fuse_sum<C>, dim3(grid), dim3(576), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

}  // namespace

// up_wimg[s]: the 1x1 weights W (C x src_channels[s]) as MFMA A fragments, bf16 [C / 16][src_channels / 32][64 lanes][8]: lane l of fragment
// (j, ks) holds W[16 j + (l & 15)][32 ks + 8 (l >> 4) .. + 7].
extern "C" int pam_fuse_sum_nhwc_bf16(void* stream, const void* base, int n_plain, const void* const* plain, const int32_t* plain_cstrides,
                                      int n_up, const void* const* up_src, const int32_t* up_shifts, const int32_t* up_channels,
                                      const void* const* up_wimg, const float* const* up_bias, void* out, int N, int H, int W, int C,
                                      int relu, int tile_a, int tile_b, int max_workgroups) {
    if (!base || !out || N < 1 || H < 1 || W < 1 || (C != 48 && C != 96 && C != 192) || n_plain < 0 || n_plain > 2 || n_up < 1 || n_up > 3) return PAM_E_ARG;
    if ((size_t)N * H * W >= (1ull << 31)) return PAM_E_ARG;
    FSArgs a;
    a.base = (const uint16_t*)base; a.out = (uint16_t*)out; a.nplain = n_plain; a.nup = n_up;
    for (int t = 0; t < 2; ++t) {
        a.plain[t] = t < n_plain ? (const uint16_t*)plain[t] : nullptr;
        a.plain_cs[t] = (t < n_plain && plain_cstrides && plain_cstrides[t] > 0) ? plain_cstrides[t] : C;
        if (t < n_plain && (!a.plain[t] || a.plain_cs[t] < C || a.plain_cs[t] % 8 != 0)) return PAM_E_ARG;
    }
    int prev = 0;
    for (int s = 0; s < 3; ++s) {
        const bool on = s < n_up;
        a.src[s] = on ? (const uint16_t*)up_src[s] : nullptr; a.wimg[s] = on ? (const char*)up_wimg[s] : nullptr;
        a.ubias[s] = (on && up_bias) ? up_bias[s] : nullptr; a.shift[s] = on ? up_shifts[s] : 0; a.src_c[s] = on ? up_channels[s] : 0;
        a.w_off[s] = a.w_bytes[s] = a.src_off[s] = a.term_off[s] = 0;
        if (!on) continue;
        // sources in ascending shift (= branch) order; the map must be a whole number of source pixels (PyTorch's up-sampling needs that too)
        if (!a.src[s] || !a.wimg[s] || a.shift[s] <= prev || a.shift[s] > 5 || a.src_c[s] < 32 || a.src_c[s] % 32 != 0) return PAM_E_ARG;
        if (H % (1 << a.shift[s]) != 0 || W % (1 << a.shift[s]) != 0) return PAM_E_ARG;
        prev = a.shift[s];
    }
    a.smax = a.shift[n_up - 1];
    a.N = N; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    // LDS plan for a tile of TA x TB pixels of the coarsest source; false = does not fit
    const int c8 = C / 8;
    size_t lds = 0;
    auto plan = [&](int TA, int TB) {
        if ((long)(TA << a.smax) * (TB << a.smax) * c8 > 512L * FS_MAXI || (TA << a.smax) > 32767 || (TB << a.smax) > 255) return false;
        size_t o = 0;
        for (int s = 0; s < n_up; ++s) { a.w_off[s] = (int)o; a.w_bytes[s] = C * a.src_c[s] * 2; o += (size_t)a.w_bytes[s]; }
        a.bias_off = (int)o; o += (size_t)n_up * C * 4; o = (o + 1023) / 1024 * 1024;
        size_t sb = 0;
        for (int s = 0; s < n_up; ++s) { const int d = a.smax - a.shift[s]; a.src_off[s] = (int)sb; sb += (size_t)(TA << d) * (TB << d) * a.src_c[s] * 2; sb = (sb + 1023) / 1024 * 1024; }
        a.src_bytes = (int)sb; a.src_base = (int)o; o += 2 * sb;
        for (int s = 0; s < n_up; ++s) { const int d = a.smax - a.shift[s]; a.term_off[s] = (int)o; o += (size_t)(TA << d) * (TB << d) * C * 2; o = (o + 15) / 16 * 16; }
        lds = o;
        return o <= 160 * 1024;
    };
    int TA = tile_a, TB = tile_b;
    if (TA <= 0 || TB <= 0) {
        // the largest candidate that fits (FS_MAXI items per thread, LDS) and divides the coarsest map: an iteration costs two barriers and a
        // round of latencies whatever its size; a 16 x 24 tile of output 0 is 2 304 items
        const int Hc = H >> a.smax, Wc = W >> a.smax;
        const int cand[8][2] = {{8, 12}, {4, 6}, {4, 3}, {2, 3}, {1, 3}, {1, 2}, {1, 1}, {0, 0}};
        TA = 0;
        for (int pass = 0; pass < 2 && TA == 0; ++pass)
            for (int k = 0; cand[k][0] && TA == 0; ++k)
                if ((pass == 1 || (Hc % cand[k][0] == 0 && Wc % cand[k][1] == 0)) && plan(cand[k][0], cand[k][1])) { TA = cand[k][0]; TB = cand[k][1]; }
        if (TA == 0) return PAM_E_ARG;
    }
    if (!plan(TA, TB)) return PAM_E_ARG;
    a.TA = TA; a.TB = TB; a.OTH = TA << a.smax; a.OTW = TB << a.smax;
    a.tiles_y = (H + a.OTH - 1) / a.OTH; a.tiles_x = (W + a.OTW - 1) / a.OTW;
    const long long nt = (long long)N * a.tiles_y * a.tiles_x;
    if (nt >= (1ll << 30)) return PAM_E_ARG;
    a.ntiles = (int)nt;
    a.inv_otw = 1.0f / (float)a.OTW;
    int ncu = 256;
    {
        static thread_local int cached_dev = -1, cached_cu = 256;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            if (dev != cached_dev) {
                hipDeviceProp_t pr;
                if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) { cached_cu = pr.multiProcessorCount; cached_dev = dev; }
            }
            ncu = cached_cu;
        }
    }
    // every workgroup walks the same number of tiles: 360 tiles -> 180 workgroups x 2 (not 256 of which 104 walk two); the CUs left over
    // take the other outputs' sums, which run on the branch streams at the same time
    int grid = max_workgroups > 0 && max_workgroups < ncu ? max_workgroups : ncu;
    const int per_wg = (a.ntiles + grid - 1) / grid;
    grid = (a.ntiles + per_wg - 1) / per_wg;
    switch (C) {
        case 48: return launch_fs<48>((hipStream_t)stream, a, lds, grid);
        case 96: return launch_fs<96>((hipStream_t)stream, a, lds, grid);
        default: return launch_fs<192>((hipStream_t)stream, a, lds, grid);
    }
}
