// libpam_hip.so, layer1 part of a1 (round 4): the 3x3 convolution of a layer1 Bottleneck AND its pointwise tail as ONE kernel
//     y2 = ReLU(conv3x3 (y1; 64 -> 64) + b2)
//     X  = ReLU(W3 . y2 + b3 + R)                      conv3 (64 -> 256) + residual
//     y1'= ReLU(W1 . X + b1)                           conv1 of the NEXT block (optional)
// Call site this stands for: the absent HRNet backend inside HRNetPose.predict, /root/reference/src/ivclabpose.py:210 (SURVEY.md
// section 8, row a1).  As two launches (k_conv3x3s<64>, then k_pw2) a block is 16 + 40-47 us of a 20-crop forward plus a launch
// boundary, serial in front of every branch, and y2 (18 MB) makes a round trip through HBM.  Here y2 never leaves the registers: the
// convolution's accumulators ARE the B fragments of the first pointwise product, as that product's are of the second (k_pw2's idiom).
//
// LDS (160 KB, all of it): [W2 72 KB: 9 taps x 64 rows x 128 B][W3 32 KB][W1 32 KB][y1 tile: 18 x 10 positions x 128 B][biases 1.5 KB].
// Work item = a 16 x 8 tile of the 96 x 72 map (persistent workgroups, one per CU, 8 waves):
//   1. the y1 tile with its one-pixel halo (180 positions, zeros outside the image) is requested into REGISTERS a whole item ahead
//      (3 x 16 B per lane) and written to LDS between two barriers once every wave is done with the previous tile;
//   2. the convolution: wave w owns rows (w & 3) + 8 (w >> 2) and that + 4 of the tile (16 pixels; two runs of 8 consecutive slots
//      40 slots apart: with the piece swizzle every fragment read is conflict-free), 4 N tiles, 18 k-steps in k_conv3x3s's order
//      (channels 0-31 of all nine taps, then channels 32-63);
//   3. the tail exactly as k_pw2 with one 16-pixel tile per wave: per 64-channel slab 8 MFMAs, + residual (requested a slab ahead; all
//      four slabs requested in front of the convolution: 41 instead of 37 us),
//      ReLU, one bf16 rounding, 2 x 16-byte stores, and the slab's 8 MFMAs of the next block's conv1.
// Same operand order per output element as the two kernels it replaces: results are bit-identical to theirs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"
#include "pam_launch.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
#define OOB_OFFSET 0x80000000u

constexpr int TR = 16, TC = 8;                         // tile
constexpr int RP = TC + 2, NPOS = (TR + 2) * RP;       // y1 positions under it: 18 x 10
constexpr int W2B = 9 * 64 * 128, W3B = 256 * 128, W1B = 4 * 64 * 128;
constexpr int W3_OFF = W2B, W1_OFF = W2B + W3B, REG_OFF = W1_OFF + W1B, BIAS_OFF = REG_OFF + NPOS * 128;
constexpr int LDS_BYTES = BIAS_OFF + (64 + 64 + 256) * 4;
static_assert(LDS_BYTES == 160 * 1024, "LDS map");
constexpr int NLOAD = NPOS * 8;                        // 16-byte pieces of a y1 tile: 1440 = 2 x 512 + 416

struct BnArgs {
    const uint16_t* y1; const uint16_t* res; const uint16_t* x0; const char* w2img; const float* b2; const char* w3img; const float* b3;
    const char* w1img; const float* b1; uint16_t* outx; uint16_t* outy;
    int N, H, W, tiles_y, tiles_x, ntiles;
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (RNE), visible to the compiler's hazard padding
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}
__device__ __forceinline__ float lo_f(uint32_t d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float hi_f(uint32_t d) { return __builtin_bit_cast(float, d & 0xffff0000u); }

// HAS2: the next block's conv1 is computed; DOWN: the FIRST block -- no residual, instead the 1x1 downsample convolution over the block input x0 as a
// second K range of the first product (w3img's second chunk; its 32 fragments live in registers: 32 KB
// more do not fit LDS)
template <bool HAS2, bool DOWN>
__global__ __launch_bounds__(512, 1) void k_bneck(BnArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // weight images -> LDS by DMA (already in LDS order): 72 + 32 (+ 32) pieces of 1 KiB
    constexpr int NPIECE = (W2B + W3B + (HAS2 ? W1B : 0)) / 1024;
    {
#pragma unroll
        for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
            const int piece = wave + 8 * i;
            if (piece < NPIECE) {
                const char* src = piece < 72 ? a.w2img + piece * 1024 : (piece < 104 ? a.w3img + (piece - 72) * 1024 : a.w1img + (piece - 104) * 1024);
                __builtin_amdgcn_global_load_lds((glb_void*)(src + lane * 16), (lds_void*)(smem + piece * 1024), 16, 0, 0);
            }
        }
    }
    float* const bias_s = (float*)(smem + BIAS_OFF);     // [b2 64][b1 64][b3 256]
    if (tid < 384) bias_s[tid] = tid < 64 ? a.b2[tid] : (tid < 128 ? (HAS2 ? a.b1[tid - 64] : 0.0f) : a.b3[tid - 128]);

    const size_t npix = (size_t)a.N * a.H * a.W;
    const auto rs_y1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.y1, 0, (int)(npix * 128), 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(DOWN ? (const void*)a.x0 : (const void*)a.res), 0, (int)(npix * (DOWN ? 128 : 512)), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.outx, 0, (int)(npix * 512), 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS2 ? a.outy : a.outx), 0, (int)(npix * (HAS2 ? 128 : 512)), 0x00020000);
    const int per_img = a.tiles_y * a.tiles_x;

    // ---- the y1 tile of an item: 1440 pieces, piece j = position j >> 3, physical 16-byte slot j & 7 (logical piece (j & 7) ^ ((pos >> 1) & 7))
    u32x4 rg[3];
    auto region_issue = [&](int T) {
        const int n = T / per_img, r = T - n * per_img, R0 = (r / a.tiles_x) * TR, C0 = (r % a.tiles_x) * TC;
        unsigned off[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int j = tid + 512 * c, P = j >> 3, ry = (P * 205) >> 11, rx = P - 10 * ry;        // P / 10, exact below 1029
            const int y = R0 - 1 + ry, x = C0 - 1 + rx;
            const bool ok = (j < NLOAD) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            off[c] = ok ? (unsigned)((n * a.H + y) * a.W + x) * 128u + (unsigned)((((j & 7) ^ (P >> 1)) & 7) << 4) : OOB_OFFSET;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) rg[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_y1, off[c], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto region_store = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int j = tid + 512 * c;
            if (j < NLOAD) *(u32x4*)(smem + REG_OFF + j * 16) = rg[c];
        }
    };

    // ---- this lane's pixel of the wave's 16-pixel tile, and its fragment addresses
    const int row = (wave & 3) + 8 * (wave >> 2) + 4 * (p >> 3), col = p & 7;
    unsigned boff[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int slot = (row + ky) * RP + col + kx;
            boff[ky * 3 + kx] = (unsigned)(REG_OFF + slot * 128 + ((g ^ ((slot >> 1) & 7)) << 4));
        }
    const unsigned aoff = (unsigned)(p * 128 + ((g ^ (p >> 1)) << 4));     // row p of a 16-row window, logical piece g (k-step 1: ^ 64)

    // DOWN: the 32 downsample fragments (second chunk of w3img) stay in registers for the whole launch: 128 VGPRs that the kernel has to spare
    bf16x8 wd[DOWN ? 32 : 1];
    if constexpr (DOWN) {
        const char* wdp = a.w3img + W3B;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) wd[sl * 8 + h * 4 + j] = *(const bf16x8*)(wdp + (sl * 4 + j) * 2048 + (h ? (aoff ^ 64u) : aoff));
    }

    int T = blockIdx.x;
    if (T < a.ntiles) region_issue(T);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // the weight DMAs (older than the three tile loads) have landed
    __syncthreads();
    if (T < a.ntiles) region_store();
    __syncthreads();

    for (; T < a.ntiles; T += gridDim.x) {
        const int n = T / per_img, r = T - n * per_img, R0 = (r / a.tiles_x) * TR, C0 = (r % a.tiles_x) * TC;
        const int y = R0 + row, x = C0 + col;
        const bool valid = (y < a.H) & (x < a.W);
        const unsigned m = (unsigned)((n * a.H + y) * a.W + x);
        const unsigned o512 = valid ? m * 512u + (unsigned)g * 32u : OOB_OFFSET, o128 = valid ? m * 128u + (unsigned)g * 32u : OOB_OFFSET;
        const bool has_next = T + (int)gridDim.x < a.ntiles;
        if (has_next) region_issue(T + (int)gridDim.x);
        u32x4 rc[2], rn[2];                               // residual of the slab being computed / the next one (DOWN: x0's two B fragments in rc)
        if constexpr (DOWN) {
            const unsigned ox0 = valid ? m * 128u + (unsigned)g * 16u : OOB_OFFSET;
            rc[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ox0, 0, 0);
            rc[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ox0, 64, 0);
        } else {
            rc[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o512, 0, 0);
            rc[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o512, 16, 0);
        }
        // ---- 3x3 convolution from LDS ------------------------------------------------------------------------------------------
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = *(const f32x4*)(bias_s + 8 * g + 32 * (j >> 1) + 4 * (j & 1));
        bf16x8 af[2][4], bfr[2];
        auto ld = [&](int s, bf16x8* a_, bf16x8& b_) {   // k-step s = (channels 32 (s / 9) .., tap s % 9)
            const int tap = s % 9;
            const unsigned xh = s >= 9 ? 64u : 0u;
            b_ = *(const bf16x8*)(smem + (boff[tap] ^ xh));
#pragma unroll
            for (int j = 0; j < 4; ++j) a_[j] = *(const bf16x8*)(smem + tap * 8192 + j * 2048 + (aoff ^ xh));
        };
        ld(0, af[0], bfr[0]);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < 18) ld(s + 1, af[nxt], bfr[nxt]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[cur][j]), __builtin_bit_cast(bf16x8_t, bfr[cur]), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        u32x4 yb[2];                                    // y2: channels 32 h + 8 g .. + 7 of this lane's pixel = the B fragments of the first product
#pragma unroll
        for (int h = 0; h < 2; ++h)
            yb[h] = (u32x4){relu_bf16x2(pack_bf16x2(acc[2 * h][0], acc[2 * h][1])), relu_bf16x2(pack_bf16x2(acc[2 * h][2], acc[2 * h][3])),
                            relu_bf16x2(pack_bf16x2(acc[2 * h + 1][0], acc[2 * h + 1][1])), relu_bf16x2(pack_bf16x2(acc[2 * h + 1][2], acc[2 * h + 1][3]))};
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading the y1 tile
        if (has_next) region_store();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the next tile is in place
        // ---- the pointwise tail (k_pw2 with one 16-pixel tile per wave) -----------------------------------------------------
        f32x4 acc1[4];
        if constexpr (HAS2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[j] = *(const f32x4*)(bias_s + 64 + 16 * g + 4 * j);
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if constexpr (DOWN) {
            } else if (sl < 3) {
                rn[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o512, (sl + 1) * 128, 0);
                rn[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o512, (sl + 1) * 128 + 16, 0);
            }
            f32x4 acc3[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc3[j] = *(const f32x4*)(bias_s + 128 + 64 * sl + 16 * g + 4 * j);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 wf = *(const bf16x8*)(smem + W3_OFF + (sl * 4 + j) * 2048 + (h ? (aoff ^ 64u) : aoff));
                    acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, yb[h]), acc3[j], 0, 0, 0);
                }
            if constexpr (DOWN) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wd[sl * 8 + h * 4 + j]), __builtin_bit_cast(bf16x8_t, rc[h]), acc3[j], 0, 0, 0);
            }
            uint32_t o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (DOWN) {
                    o[2 * j] = relu_bf16x2(pack_bf16x2(acc3[j][0], acc3[j][1]));
                    o[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc3[j][2], acc3[j][3]));
                } else {
                    const uint32_t d0 = rc[j >> 1][(2 * j) & 3], d1 = rc[j >> 1][(2 * j + 1) & 3];
                    o[2 * j] = relu_bf16x2(pack_bf16x2(acc3[j][0] + lo_f(d0), acc3[j][1] + hi_f(d0)));
                    o[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc3[j][2] + lo_f(d1), acc3[j][3] + hi_f(d1)));
                }
            }
            const u32x4 xf[2] = {(u32x4){o[0], o[1], o[2], o[3]}, (u32x4){o[4], o[5], o[6], o[7]}};
            __builtin_amdgcn_raw_buffer_store_b128(xf[0], rs_x, o512, sl * 128, 0);
            __builtin_amdgcn_raw_buffer_store_b128(xf[1], rs_x, o512, sl * 128 + 16, 0);
            if constexpr (HAS2) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 wf = *(const bf16x8*)(smem + W1_OFF + sl * 8192 + j * 2048 + (h ? (aoff ^ 64u) : aoff));
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf[h]), acc1[j], 0, 0, 0);
                    }
            }
            if constexpr (!DOWN) { rc[0] = rn[0]; rc[1] = rn[1]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS2) {
            uint32_t o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] = relu_bf16x2(pack_bf16x2(acc1[j][0], acc1[j][1]));
                o[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc1[j][2], acc1[j][3]));
            }
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[0], o[1], o[2], o[3]}, rs_y, o128, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[4], o[5], o[6], o[7]}, rs_y, o128, 16, 0);
        }
    }
}

template <bool HAS2, bool DOWN>
int launch_bneck(hipStream_t s, const BnArgs& a) {
    if (!pam_max_dynamic_lds((const void*)k_bneck<HAS2, DOWN>, LDS_BYTES)) return PAM_E_HIP;
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
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    pam_launch(k_bneck<HAS2, DOWN>, dim3(grid), dim3(512), LDS_BYTES, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

}  // namespace

// Images the host packs (bf16; layouts also in include/pam.h): w2img [9 taps][64 rows][64 K] as pam_stem_fused_nhwc_bf16's; w3_img (one K
// source with a residual, two with x0) / w1_img as pam_bottleneck_tail_nhwc_bf16's.
extern "C" int pam_bottleneck_fused_nhwc_bf16(void* stream, const void* y1, const void* x0, const void* residual, const void* w2img,
                                              const float* bias2, const void* w3_img, const float* bias3, const void* w1_img,
                                              const float* bias1, void* out_x, void* out_y1, int N, int H, int W) {
    if (!y1 || (residual == nullptr) == (x0 == nullptr) || !w2img || !bias2 || !w3_img || !bias3 || !out_x || N <= 0 || H <= 0 || W <= 0) return PAM_E_ARG;
    if ((w1_img != nullptr) != (out_y1 != nullptr) || (w1_img && !bias1)) return PAM_E_ARG;
    if ((size_t)N * H * W * 512 >= (1ull << 31)) return PAM_E_ARG;
    BnArgs a;
    a.y1 = (const uint16_t*)y1; a.res = (const uint16_t*)residual; a.x0 = (const uint16_t*)x0; a.w2img = (const char*)w2img; a.b2 = bias2;
    a.w3img = (const char*)w3_img; a.b3 = bias3; a.w1img = (const char*)w1_img; a.b1 = bias1; a.outx = (uint16_t*)out_x; a.outy = (uint16_t*)out_y1;
    a.N = N; a.H = H; a.W = W;
    a.tiles_y = (H + TR - 1) / TR; a.tiles_x = (W + TC - 1) / TC;
    a.ntiles = N * a.tiles_y * a.tiles_x;
    hipStream_t s = (hipStream_t)stream;
    if (x0) return w1_img ? launch_bneck<true, true>(s, a) : launch_bneck<false, true>(s, a);
    return w1_img ? launch_bneck<true, false>(s, a) : launch_bneck<false, false>(s, a);
}
