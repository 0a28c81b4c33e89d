// libpam_hip.so, fuse-layer part of a1 (round 5): one output of an HR module's fuse layer in ONE launch
//     out_i = ReLU( x_i + sum_{j < i} down_ij + sum_{j > i} nearest_up_{2^(j-i)}( W_ij x_j + b_ij ) )
// (hrnet.py:64-100; stands inside the absent HRNet backend behind /root/reference/src/ivclabpose.py:210).
// down_ij = the result of the strided-convolution chain from the finer branch j (already at out_i's resolution: a "plain" term);
// the coarser branches j > i enter through a 1x1 convolution at THEIR resolution, up-sampled by pixel replication.
//
// Until round 5 the 1x1 products were launches of their own (18 per forward on the generic gather kernel: 2 % of the MFMA roof, 12 % of
// the HBM roof, 7 us each, every one of them on some branch's tail) whose results went to HBM and came back into k_upsample_add.
// Here a workgroup owns a tile of out_i that is a whole number of pixels of the COARSEST source (TA x TB of them): it computes the 1x1
// products of the source pixels under its tile itself -- B fragments = the lane's 16 bytes of a source pixel straight from global
// memory, A fragments = host-packed 1 KiB weight fragments straight from L2, accumulators start from the bias, one bf16 rounding, exactly
// the arithmetic of the convolution launches they replace -- into a few KB of LDS, and then streams base + plain terms + the replicated
// LDS terms -> ReLU -> out, 16 bytes per lane, in k_upsample_add's summation order.  128 registers: two workgroups per CU, one streams
// while the other multiplies.
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

constexpr int FS_MAXI = 5;                 // 16-byte items of the output tile per thread at most
constexpr int FS_KB = 6;                   // k-steps whose fragments are requested together

struct FSArgs {
    const uint16_t* base; uint16_t* out;
    const uint16_t* plain[2]; int plain_cs[2]; int nplain;
    const uint16_t* src[3]; const uint16_t* wimg[3]; const float* ubias[3]; int shift[3]; int src_c[3]; int nup;
    int N, H, W, relu, TA, TB, tiles_y, tiles_x, OTH, OTW, smax;
    int lds_off[3];                         // byte offset of source s's term tile in LDS
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
__global__ __launch_bounds__(512, 4) void k_fuse_sum(FSArgs a) {
    constexpr int C8 = C / 8, NT = C / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = (int)blockIdx.x / per_img, trem = (int)blockIdx.x - n * per_img, ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int oy0 = ty * a.OTH, ox0 = tx * a.OTW;

    // ---- the 1x1 products of the source pixels under the tile -> LDS.  job = (source, 16-pixel tile, 16-channel tile), coarsest (deepest K)
    // source first; wave w takes jobs w, w + 8, ...
    int job0[4];                                          // first job of source nup - 1, nup - 2, ...; job0[nup] = number of jobs
    {
        int acc_j = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            job0[q] = acc_j;
            if (q < a.nup) {
                const int s = a.nup - 1 - q, d = a.smax - a.shift[s];
                acc_j += ((((a.TA << d) * (a.TB << d)) + 15) >> 4) * NT;
            }
        }
        job0[3] = acc_j;
    }
    for (int job = wave; job < job0[3]; job += 8) {
        const int q = job >= job0[2] ? 2 : (job >= job0[1] ? 1 : 0);
        const int s = a.nup - 1 - q, jj = job - job0[q];
        const int mt = jj / NT, j = jj - mt * NT;
        const int sh = a.shift[s], d = a.smax - sh, th = a.TA << d, tw = a.TB << d, npx = th * tw;
        const int Cs = a.src_c[s], KS = Cs >> 5, Hs = a.H >> sh, Ws = a.W >> sh;
        const int p = mt * 16 + l15;
        const int pr = p / tw, pc = p - pr * tw;
        const int sy = (oy0 >> sh) + pr, sx = (ox0 >> sh) + pc;
        const bool okp = p < npx && sy < Hs && sx < Ws;
        const uint16_t* xp = a.src[s] + (okp ? (((long)n * Hs + sy) * Ws + sx) * Cs : (long)n * Hs * Ws * Cs) + 8 * g;
        const uint16_t* wp = a.wimg[s] + ((long)j * KS * 64 + lane) * 8;
        f32x4 acc = a.ubias[s] ? *(const f32x4*)(a.ubias[s] + 16 * j + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < KS; k0 += FS_KB) {
            bf16x8 fa[FS_KB], fb[FS_KB];
#pragma unroll
            for (int u = 0; u < FS_KB; ++u)
                if (k0 + u < KS) {
                    fa[u] = *(const bf16x8*)(wp + (long)(k0 + u) * 512);
                    fb[u] = *(const bf16x8*)(xp + (k0 + u) * 32);
                }
#pragma unroll
            for (int u = 0; u < FS_KB; ++u)
                if (k0 + u < KS)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[u]), __builtin_bit_cast(bf16x8_t, fb[u]), acc, 0, 0, 0);
        }
        if (p < npx)
            *(u32x2*)(smem + a.lds_off[s] + (p * C + 16 * j + 4 * g) * 2) = (u32x2){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
    }
    __syncthreads();

    // ---- out = [ReLU](base + plain terms + replicated products), k_upsample_add's order: base, then the terms in branch order.  All of a
    // thread's loads are requested before the first sum (two workgroups share a CU: one streams while the other multiplies)
    const int nitem = a.OTH * a.OTW * C8;
    bf16x8 vb[FS_MAXI], vp[2][FS_MAXI];
    int eo[FS_MAXI];                                      // pixel index of the item in the output tensor (-1: outside the image); the host checks N H W < 2^31
    int trc[FS_MAXI];                                     // (tile row << 16) | (tile column << 8) | 8-channel group
#pragma unroll
    for (int k = 0; k < FS_MAXI; ++k) {
        const int e = tid + 512 * k;
        const int pix = e / C8, c8 = e - pix * C8;
        const int r = fdiv_small(pix, a.inv_otw), c = pix - r * a.OTW;
        const int oy = oy0 + r, ox = ox0 + c;
        const bool ok = e < nitem && oy < a.H && ox < a.W;
        const int p = (n * a.H + oy) * a.W + ox;
        eo[k] = ok ? p : -1;
        trc[k] = (r << 16) | (c << 8) | c8;
        const long pz = ok ? p : 0;
        vb[k] = *(const bf16x8*)(a.base + pz * C + c8 * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            vp[t][k] = t < a.nplain ? *(const bf16x8*)(a.plain[t] + pz * a.plain_cs[t] + c8 * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
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
                const bf16x8 q = *(const bf16x8*)(smem + a.lds_off[s] + (((r >> sh) * tw + (c >> sh)) * C + c8 * 8) * 2);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bf2f((uint16_t)q[e]);
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(a.relu ? fmaxf(v[e], 0.0f) : v[e]);
        *(bf16x8*)(a.out + (long)eo[k] * C + c8 * 8) = o;
    }
}

template <int C>
int launch_fs(hipStream_t s, const FSArgs& a, size_t lds) {
    if (lds > 64 * 1024 && !pam_max_dynamic_lds((const void*)k_fuse_sum<C>, (int)lds)) return PAM_E_HIP;
    pam_launch(k_fuse_sum<C>, dim3(a.N * a.tiles_y * a.tiles_x), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

}  // namespace

// up_wimg[s]: the 1x1 weights W (C x src_channels[s]) as MFMA A fragments, bf16 [C / 16][src_channels / 32][64 lanes][8]: lane l of fragment
// (j, ks) holds W[16 j + (l & 15)][32 ks + 8 (l >> 4) .. + 7].
extern "C" int pam_fuse_sum_nhwc_bf16(void* stream, const void* base, int n_plain, const void* const* plain, const int32_t* plain_cstrides,
                                      int n_up, const void* const* up_src, const int32_t* up_shifts, const int32_t* up_channels,
                                      const void* const* up_wimg, const float* const* up_bias, void* out, int N, int H, int W, int C,
                                      int relu, int tile_a, int tile_b) {
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
        a.src[s] = on ? (const uint16_t*)up_src[s] : nullptr; a.wimg[s] = on ? (const uint16_t*)up_wimg[s] : nullptr;
        a.ubias[s] = (on && up_bias) ? up_bias[s] : nullptr; a.shift[s] = on ? up_shifts[s] : 0; a.src_c[s] = on ? up_channels[s] : 0;
        if (!on) continue;
        // sources in ascending shift (= branch) order; the map must be a whole number of source pixels (PyTorch's up-sampling needs that too)
        if (!a.src[s] || !a.wimg[s] || a.shift[s] <= prev || a.shift[s] > 5 || a.src_c[s] < 32 || a.src_c[s] % 32 != 0) return PAM_E_ARG;
        if (H % (1 << a.shift[s]) != 0 || W % (1 << a.shift[s]) != 0) return PAM_E_ARG;
        prev = a.shift[s];
    }
    a.smax = a.shift[n_up - 1];
    a.N = N; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    // tile = TA x TB pixels of the coarsest source: as large as FS_MAXI items per thread allow, at most 4 x 3 (more workgroups, and
    // a 12 x 9 coarsest map divides by it); callers may state it
    const int c8 = C / 8;
    int TA = tile_a, TB = tile_b;
    if (TA <= 0 || TB <= 0) {
        // the largest candidate that fits FS_MAXI items per thread, divides the coarsest map and still gives every CU a workgroup;
        // failing that, the one with the most workgroups (the smallest that fits)
        const int cand[8][2] = {{8, 12}, {8, 6}, {4, 6}, {4, 3}, {2, 3}, {1, 3}, {1, 1}, {0, 0}};
        const int Hc = H >> a.smax, Wc = W >> a.smax;
        TA = 0;
        for (int pass = 0; pass < 3 && TA == 0; ++pass)
            for (int k = 0; cand[k][0] && TA == 0; ++k) {
                const int ta = cand[k][0], tb = cand[k][1];
                if ((long)(ta << a.smax) * (tb << a.smax) * c8 > 512L * FS_MAXI) continue;
                const bool divides = Hc % ta == 0 && Wc % tb == 0;
                const long wgs = (long)N * ((Hc + ta - 1) / ta) * ((Wc + tb - 1) / tb);
                if ((pass == 0 && divides && wgs >= 256) || (pass == 1 && divides && (cand[k + 1][0] == 0 || k >= 4)) || pass == 2) { TA = ta; TB = tb; }
            }
        if (TA == 0) return PAM_E_ARG;
    }
    a.TA = TA; a.TB = TB; a.OTH = TA << a.smax; a.OTW = TB << a.smax;
    if ((long)a.OTH * a.OTW * c8 > 512L * FS_MAXI || a.OTH > 65535 || a.OTW > 255) return PAM_E_ARG;
    a.tiles_y = (H + a.OTH - 1) / a.OTH; a.tiles_x = (W + a.OTW - 1) / a.OTW;
    if ((long long)N * a.tiles_y * a.tiles_x >= (1ll << 30)) return PAM_E_ARG;
    a.inv_otw = 1.0f / (float)a.OTW;
    size_t lds = 0;
    for (int s = 0; s < 3; ++s) {
        a.lds_off[s] = (int)lds;
        if (s < n_up) { const int d = a.smax - a.shift[s]; lds += (size_t)(TA << d) * (TB << d) * C * 2; lds = (lds + 15) / 16 * 16; }
    }
    if (lds > 150 * 1024) return PAM_E_ARG;
    switch (C) {
        case 48: return launch_fs<48>((hipStream_t)stream, a, lds);
        case 96: return launch_fs<96>((hipStream_t)stream, a, lds);
        default: return launch_fs<192>((hipStream_t)stream, a, lds);
    }
}
