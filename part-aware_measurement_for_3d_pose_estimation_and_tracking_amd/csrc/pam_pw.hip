// libpam_hip.so, conv part of a1: the pointwise tail of an HRNet layer1 Bottleneck as ONE kernel.
//
// HRNet-W48's layer1 is four Bottlenecks at 96 x 72 (conv1 1x1 -> 64, conv2 3x3 64 -> 64, conv3 1x1 64 -> 256, + residual, ReLU).  Its
// 256-channel tensors make it HBM / load-path bound (5.5 % of the network's FLOPs, 14 % of the 20-crop forward when every
// convolution is its own launch: the 256-channel block output is written, read back as the residual of the next block AND read again
// by that block's conv1).  k_pw2 chains everything that is pointwise around that tensor:
//
//     X  = ReLU(W3 . y2 [+ Wd . x0] + b3 [+ R])        conv3 (+ the first block's 1x1 downsample as a second K range) + residual
//     y1 = ReLU(W1 . X + b1)                           conv1 of the NEXT block (optional)
//
// per pixel: 128 B (y2) [+ 512 B residual] in, 512 B (X) + 128 B (y1) out; X is read exactly once (as the next residual).
// The 3x3 convolution between two such launches stays on k_conv3x3s<64>.
//
// Decomposition: purely per pixel, so the unit of work is a WAVE tile of 16 * MT pixels; a 512-thread workgroup (8 independent waves,
// one workgroup per CU, persistent) keeps both weight matrices in LDS (host-packed images, 32 KB per 64-deep K chunk) and every wave
// streams its tiles on its own: no barrier after the weight copy.  Weights are the MFMA A operand (v_mfma_f32_16x16x32_bf16), pixels the
// B operand; the rows of a 64-channel slab are permuted on the host so that a lane ends with 16 CONTIGUOUS channels of its pixel
// (32-byte residual loads / stores, as in k_conv3x3), and those 16 bf16 values ARE the lane's B fragments of the second product:
// k-step h of slab sl, lane group g, element e = channel 64 sl + 16 g + 8 h + e -- W1's image stores its K dimension in that order,
// so X never passes through LDS or another lane (the accumulator-as-operand idiom of the CDNA guide, in 16x16x32 form).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"
#include "pam_launch.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;
#define OOB_OFFSET 0x80000000u

__device__ __forceinline__ uint32_t pw_pack(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float pw_lo(uint32_t d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float pw_hi(uint32_t d) { return __builtin_bit_cast(float, d & 0xffff0000u); }
__device__ __forceinline__ uint32_t pw_relu2(uint32_t d) {          // ReLU on a bf16 pair: one packed int16 max (bf16 is sign-magnitude)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, d), (s16x2){0, 0}));
}

struct PwArgs {
    const uint16_t* a0; const uint16_t* a1; const uint16_t* res; const uint16_t* w3; const float* b3;
    const uint16_t* w1; const float* b1; uint16_t* outx; uint16_t* outy; int M;
};

// S = number of 64-channel K sources of the first product (1, or 2 = y2 and the block input x0 with the downsample weights),
// RES = a 256-channel residual is added, HAS2 = the second product (the next block's conv1) is computed, MT = 16-pixel tiles per wave tile
template <int S, bool RES, bool HAS2, int MT, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_pw2(PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* W3s = smem;                                   // [S chunks][256 rows][128 B]
    char* W1s = smem + S * 32768;                       // [4 chunks][64 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, px = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwt = (a.M + 16 * MT - 1) / (16 * MT);
    const int stride = (int)gridDim.x * NW;
    int wt = (int)blockIdx.x * NW + wave;

    const auto rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.a0, 0, (int)((size_t)a.M * 128), 0x00020000);
    const auto rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(S == 2 ? a.a1 : a.a0), 0, (int)((size_t)a.M * 128), 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.res : a.outx), 0, (int)((size_t)a.M * 512), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.outx, 0, (int)((size_t)a.M * 512), 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS2 ? a.outy : a.outx), 0, (int)((size_t)a.M * (HAS2 ? 128 : 512)), 0x00020000);

    u32x4 yb[MT][2 * S];                                // B fragments of the first product: k-step 2 c + h = channels 32 h + 8 g .. + 7 of source c
    u32x4 rc[MT][2];                                    // residual of the slab being computed: channels 64 sl + 16 g .. + 15
    auto pix_off = [&](int wt_, int mt, unsigned bytes_per_pixel) -> unsigned {
        const int m = wt_ * 16 * MT + mt * 16 + px;
        return m < a.M ? (unsigned)m * bytes_per_pixel : OOB_OFFSET;
    };
    auto load_y = [&](int wt_, u32x4 (&y)[MT][2 * S]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned o = pix_off(wt_, mt, 128);
            const unsigned oo = o == OOB_OFFSET ? o : o + g * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                y[mt][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a0, oo, h * 64, 0);
                if constexpr (S == 2) y[mt][2 + h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a1, oo, h * 64, 0);
            }
        }
    };
    auto load_res = [&](int wt_, int sl, u32x4 (&r)[MT][2]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned o = pix_off(wt_, mt, 512);
            const unsigned oo = o == OOB_OFFSET ? o : o + sl * 128 + g * 32;
            r[mt][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, oo, 0, 0);
            r[mt][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, oo, 16, 0);
        }
    };

    // the first tile's operands go in flight before the weight copy (which then runs under their latency)
    if (wt < nwt) {
        load_y(wt, yb);
        if constexpr (RES) load_res(wt, 0, rc);
    }
    {   // weight images -> LDS: linear 16-byte copy of the host-packed (already swizzled) images
        constexpr int N3 = S * 32768 / 16, N1 = HAS2 ? 32768 / 16 : 0;
        for (int q = tid; q < N3; q += 64 * NW) *(u32x4*)(W3s + (size_t)q * 16) = *(const u32x4*)((const char*)a.w3 + (size_t)q * 16);
        for (int q = tid; q < N1; q += 64 * NW) *(u32x4*)(W1s + (size_t)q * 16) = *(const u32x4*)((const char*)a.w1 + (size_t)q * 16);
    }
    __syncthreads();

    // LDS byte offset of this lane's A fragment inside a 16-row tile of a chunk image: row px, logical piece 4 h + g at physical
    // position (4 h + g) ^ ((row >> 1) & 7); h = 1 toggles bit 6
    const unsigned s3 = (unsigned)px >> 1;
    const unsigned fo0 = (unsigned)px * 128 + (((unsigned)g ^ (s3 & 3)) << 4) + ((s3 >> 2) << 6);

    f32x4 bias1[4];
    if constexpr (HAS2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias1[j] = *(const f32x4*)(a.b1 + 16 * g + 4 * j);
    }

    for (; wt < nwt; wt += stride) {
        const int wn = wt + stride;
        const bool has_next = wn < nwt;
        u32x4 yn[MT][2 * S];
        f32x4 acc1[MT][4];
        if constexpr (HAS2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[mt][j] = bias1[j];
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            // operands of what comes next go in flight now: the next slab's residual (or the next tile's first), the next tile's y2
            u32x4 rn[MT][2];
            if constexpr (RES) {
                if (sl < 3) load_res(wt, sl + 1, rn);
                else if (has_next) load_res(wn, 0, rn);
            }
            if (sl == 1 && has_next) load_y(wn, yn);
            // the weight fragments are loop-invariant LDS reads: without this the compiler hoists ALL of them (64 fragments = 256 VGPRs)
            // out of the tile loop and spills; an opaque copy of the lane's fragment offset per slab keeps each read where it is used
            unsigned f0 = fo0;
            asm volatile("" : "+v"(f0));
            const unsigned f1 = f0 ^ 64u;
            // ---- first product, slab sl: 64 output channels x 16 MT pixels, K = 64 S
            f32x4 acc3[MT][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *(const f32x4*)(a.b3 + 64 * sl + 16 * g + 4 * j);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc3[mt][j] = b4;
            }
#pragma unroll
            for (int c = 0; c < S; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8_t wf = *(const bf16x8_t*)(W3s + c * 32768 + (sl * 4 + j) * 2048 + (h ? f1 : f0));
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc3[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, yb[mt][2 * c + h]), acc3[mt][j], 0, 0, 0);
                    }
            // ---- epilogue of the slab: + residual, ReLU, one bf16 rounding; the 16 channels of this lane = 2 x 16 bytes of X and the two
            // B fragments of the second product
            u32x4 xf[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint32_t o[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v0 = acc3[mt][j][0], v1 = acc3[mt][j][1], v2 = acc3[mt][j][2], v3 = acc3[mt][j][3];
                    if constexpr (RES) {
                        const uint32_t d0 = rc[mt][j >> 1][(2 * j) & 3], d1 = rc[mt][j >> 1][(2 * j + 1) & 3];
                        v0 += pw_lo(d0); v1 += pw_hi(d0); v2 += pw_lo(d1); v3 += pw_hi(d1);
                    }
                    o[2 * j] = pw_relu2(pw_pack(v0, v1)); o[2 * j + 1] = pw_relu2(pw_pack(v2, v3));
                }
                xf[mt][0] = (u32x4){o[0], o[1], o[2], o[3]}; xf[mt][1] = (u32x4){o[4], o[5], o[6], o[7]};
                const unsigned po = pix_off(wt, mt, 512);
                const unsigned oo = po == OOB_OFFSET ? po : po + sl * 128 + g * 32;
                __builtin_amdgcn_raw_buffer_store_b128(xf[mt][0], rs_x, oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(xf[mt][1], rs_x, oo, 16, 0);
            }
            // ---- second product, K range of this slab: k-step h = channels 64 sl + 16 g + 8 h .. + 7 (W1's image is packed in that order)
            if constexpr (HAS2) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8_t wf = *(const bf16x8_t*)(W1s + sl * 8192 + j * 2048 + (h ? f1 : f0));
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc1[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, xf[mt][h]), acc1[mt][j], 0, 0, 0);
                    }
            }
            if constexpr (RES) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { rc[mt][0] = rn[mt][0]; rc[mt][1] = rn[mt][1]; }
            }
        }
        if constexpr (HAS2) {                           // y1 = ReLU(acc1): this lane's 16 contiguous channels of each pixel
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint32_t o[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[2 * j] = pw_relu2(pw_pack(acc1[mt][j][0], acc1[mt][j][1]));
                    o[2 * j + 1] = pw_relu2(pw_pack(acc1[mt][j][2], acc1[mt][j][3]));
                }
                const unsigned po = pix_off(wt, mt, 128);
                const unsigned oo = po == OOB_OFFSET ? po : po + g * 32;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[0], o[1], o[2], o[3]}, rs_y, oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[4], o[5], o[6], o[7]}, rs_y, oo, 16, 0);
            }
        }
        if (has_next) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k = 0; k < 2 * S; ++k) yb[mt][k] = yn[mt][k];
        }
    }
}

template <int S, bool RES, bool HAS2, int MT, int NW>
static int launch_pw2(hipStream_t s, const PwArgs& a, int max_wg) {
    constexpr size_t lds = (size_t)S * 32768 + (HAS2 ? 32768 : 0);
    if (!pam_max_dynamic_lds((const void*)k_pw2<S, RES, HAS2, MT, NW>, (int)lds)) return PAM_E_HIP;
    const int nwt = (a.M + 16 * MT - 1) / (16 * MT);
    int grid = (nwt + NW - 1) / NW;
    if (grid > max_wg) grid = max_wg;                   // persistent: one workgroup per CU, every wave walks wave tiles wt, wt + NW grid, ...
    pam_launch(k_pw2<S, RES, HAS2, MT, NW>, dim3(grid), dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// ---- k_pw1: y = ReLU(W . x + b), 64 -> 64 channels, pointwise (the first Bottleneck's conv1, whose input is the stem's 64-channel output).
// 128 B in, 128 B out per pixel: a pure stream.  The whole weight matrix is 8 A fragments = 32 VGPRs per lane, loaded once (w_img: chunk 0 of
// the w1 layout below WITHOUT the K permutation, i.e. row 16 jt + qq = output channel 16 (qq >> 2) + 4 jt + (qq & 3), natural K order, pieces
// swizzled like every chunk image); every wave streams 16-pixel tiles on its own, four tiles in flight.
struct Pw1Args { const uint16_t* in; const uint16_t* w; const float* b; uint16_t* out; int M; int act; };   // act: 1 ReLU, 2 leaky ReLU (slope 0.1; Darknet)
__global__ __launch_bounds__(256) void k_pw1(Pw1Args a) {
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, px = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.M * 128), 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((size_t)a.M * 128), 0x00020000);
    bf16x8_t wf[2][4];
    {
        const unsigned s3 = (unsigned)px >> 1;
        const unsigned fo0 = (unsigned)px * 128 + (((unsigned)g ^ (s3 & 3)) << 4) + ((s3 >> 2) << 6);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[h][j] = *(const bf16x8_t*)((const char*)a.w + j * 2048 + (h ? (fo0 ^ 64u) : fo0));
    }
    f32x4 bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[j] = *(const f32x4*)(a.b + 16 * g + 4 * j);
    constexpr int U = 4;                                 // 16-pixel tiles per iteration (all their loads issued before the first MFMA)
    const int ntile = (a.M + 15) >> 4, stride = (int)gridDim.x * 4 * U;
    for (int t0 = ((int)blockIdx.x * 4 + wave) * U; t0 < ntile; t0 += stride) {
        u32x4 xb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = (t0 + u) * 16 + px;
            const unsigned o = (t0 + u < ntile && m < a.M) ? (unsigned)m * 128u + g * 16 : OOB_OFFSET;
            xb[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, o, 0, 0);
            xb[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, o, 64, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = bias[j];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][j], __builtin_bit_cast(bf16x8_t, xb[u][h]), acc[j], 0, 0, 0);
            uint32_t o[8];
            if (a.act == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][r] = acc[j][r] > 0.0f ? acc[j][r] : 0.1f * acc[j][r];
                    o[2 * j] = pw_pack(acc[j][0], acc[j][1]); o[2 * j + 1] = pw_pack(acc[j][2], acc[j][3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[2 * j] = pw_relu2(pw_pack(acc[j][0], acc[j][1])); o[2 * j + 1] = pw_relu2(pw_pack(acc[j][2], acc[j][3])); }
            }
            const int m = (t0 + u) * 16 + px;
            const unsigned oo = (t0 + u < ntile && m < a.M) ? (unsigned)m * 128u + g * 32 : OOB_OFFSET;
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[0], o[1], o[2], o[3]}, rs_out, oo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[4], o[5], o[6], o[7]}, rs_out, oo, 16, 0);
        }
    }
}
extern "C" int pam_pointwise64_act_nhwc_bf16(void* stream, const void* in, const void* w_img, const float* bias, void* out, long long n_pixels, int act);
extern "C" int pam_pointwise64_relu_nhwc_bf16(void* stream, const void* in, const void* w_img, const float* bias, void* out, long long n_pixels) {
    return pam_pointwise64_act_nhwc_bf16(stream, in, w_img, bias, out, n_pixels, 1);
}
extern "C" int pam_pointwise64_act_nhwc_bf16(void* stream, const void* in, const void* w_img, const float* bias, void* out, long long n_pixels, int act) {
    if (!in || !w_img || !bias || !out || n_pixels <= 0 || n_pixels * 128 >= (1ll << 31) || (act != 1 && act != 2)) return PAM_E_ARG;
    Pw1Args a; a.in = (const uint16_t*)in; a.w = (const uint16_t*)w_img; a.b = bias; a.out = (uint16_t*)out; a.M = (int)n_pixels; a.act = act;
    const int ntile = (a.M + 15) >> 4;
    int grid = (ntile + 15) / 16;                        // 4 waves x 4 tiles per workgroup and iteration
    if (grid > 2048) grid = 2048;
    pam_launch(k_pw1, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// Weight image layouts (what the host must pack; bf16):
//   w3_img [S][256 rows][64 K] -- chunk c = K source c (0: conv3 over y2, 1: the downsample conv over x0); row 64 sl + 16 jt + qq (qq < 16)
//     holds output channel 64 sl + 16 (qq >> 2) + 4 jt + (qq & 3); the row's 16-byte piece at PHYSICAL position p holds K values
//     8 q .. 8 q + 7 with q = p ^ ((row >> 1) & 7).
//   w1_img [4][64 rows][64 K] -- chunk sl = input channels 64 sl .. 64 sl + 63 of the 256; row 16 jt + qq holds output channel
//     16 (qq >> 2) + 4 jt + (qq & 3); physical piece p holds, with q = p ^ ((row >> 1) & 7), h = q >> 2, g = q & 3, the input channels
//     64 sl + 16 g + 8 h .. + 7 (the order in which a lane of the first product holds its 16 outputs).
extern "C" int pam_bottleneck_tail_nhwc_bf16(void* stream, const void* y2, const void* x0, const void* residual, const void* w3_img,
                                             const float* bias3, const void* w1_img, const float* bias1, void* out_x, void* out_y1,
                                             long long n_pixels, int tile_cfg) {
    if (!y2 || !w3_img || !bias3 || !out_x || n_pixels <= 0 || n_pixels * 512 >= (1ll << 31)) return PAM_E_ARG;
    if ((w1_img != nullptr) != (out_y1 != nullptr) || (w1_img && !bias1) || (x0 && residual)) return PAM_E_ARG;
    PwArgs a;
    a.a0 = (const uint16_t*)y2; a.a1 = (const uint16_t*)x0; a.res = (const uint16_t*)residual; a.w3 = (const uint16_t*)w3_img; a.b3 = bias3;
    a.w1 = (const uint16_t*)w1_img; a.b1 = bias1; a.outx = (uint16_t*)out_x; a.outy = (uint16_t*)out_y1; a.M = (int)n_pixels;
    hipStream_t s = (hipStream_t)stream;
    const int S = x0 ? 2 : 1, has2 = w1_img ? 1 : 0, res = residual ? 1 : 0;
    // tile_cfg = MT + 10 * (1: 16 waves per workgroup instead of 8; 2: two 8-wave workgroups per CU): MT = 16-pixel tiles per wave tile (1, 2 or 3); <= 0 = automatic.
    // Purely a latency-hiding choice (the waves are independent): measured at 20 crops, 96 x 72: MT = 1 34 us, 2 44 us, 3 42 us with 8 waves
    const int cfg = tile_cfg > 0 ? tile_cfg : 1;
    const int mt = cfg % 10, nw16 = cfg / 10 == 1;
    if (mt < 1 || mt > 3 || cfg / 10 > 2 || (cfg / 10 && mt != 1)) return PAM_E_ARG;
    const int max_wg = cfg / 10 == 2 ? 512 : 256;        // 21: two 8-wave workgroups per CU (124 VGPRs, 64 KB of LDS each)
#define PW_CASE(S_, R_, H_) \
    if (S == S_ && res == R_ && has2 == H_) { \
        if (nw16) return launch_pw2<S_, R_ != 0, H_ != 0, 1, 16>(s, a, max_wg); \
        if (mt == 1) return launch_pw2<S_, R_ != 0, H_ != 0, 1, 8>(s, a, max_wg); \
        if (mt == 2) return launch_pw2<S_, R_ != 0, H_ != 0, 2, 8>(s, a, max_wg); \
        return launch_pw2<S_, R_ != 0, H_ != 0, 3, 8>(s, a, max_wg); }
    PW_CASE(1, 1, 1) PW_CASE(1, 1, 0) PW_CASE(2, 0, 1) PW_CASE(1, 0, 1) PW_CASE(1, 0, 0) PW_CASE(2, 0, 0)
#undef PW_CASE
    return PAM_E_ARG;
}
