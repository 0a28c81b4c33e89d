// libpam_hip.so, stem part of a1 (round 4): HRNet's stem and the first pointwise convolution of layer1 as ONE kernel
//     c1 = ReLU(conv3x3 s2 (x;  8 -> 64) + b1)          384 x 288 -> 192 x 144     (x = the cropped person, RGB + 5 zero channels)
//     x0 = ReLU(conv3x3 s2 (c1; 64 -> 64) + b2)         192 x 144 ->  96 x 72
//     y1 = ReLU(conv1x1    (x0; 64 -> 64) + bp)         the first Bottleneck's conv1
// Call site this stands for: the absent HRNet backend inside HRNetPose.predict, /root/reference/src/ivclabpose.py:210 (SURVEY.md
// section 8, row a1).  As three launches (k_conv_stem, k_conv_gs, k_pw1) the chain is 21 + 38 + 10 us of a 20-crop forward plus two
// launch boundaries, all of it serial in front of every branch, and c1 (71 MB at 20 crops) is written once and gathered 2.25 times.
// Here c1 never leaves the CU.
//
// LDS: [W2 72 KB][c1 buffer 0: 17 rows x 20 slots x 128 B = 42.5 KB][c1 buffer 1][biases 768 B] = 157.8 KB.
// Work item = an 8 x 8 tile of x0; persistent workgroups, one per CU, eight waves in two roles (one wave of each per SIMD):
//   producers (waves 0-3): conv1 on the 17 x 17 positions of c1 the tile's windows touch (289 positions = 19 wave tiles of 16; 13 %
//      recomputed in the halo).  As in k_conv_stem one MFMA covers a tap row (3 taps x 8 channels = 24 <= 32), the B fragment is the
//      lane's 16-byte input pixel straight from global memory -- two register sets, item k + 1's pixels requested before item k is
//      computed -- and the 12 weight fragments stay in registers.  The result (zero outside c1: conv2's padding) goes to one of TWO
//      c1 buffers in LDS, 128 B per position.
//   consumers (waves 4-7): conv2 from the other buffer: wave w owns x0 rows 2 w, 2 w + 1 of the tile (16 pixels), 4 N tiles, K = 9
//      taps x 64 channels = 18 k-steps.  c1's columns are stored by parity (even columns, then odd ones, rows 20 slots apart) so that
//      the 16 pixels of a stride-2 window read 2 x 8 CONSECUTIVE slots: with the usual piece swizzle (piece ^ (slot >> 1)) every
//      fragment read is conflict-free.  conv2's weights (72 KB, [tap][64 rows][128 B], pieces swizzled the same way) are resident in
//      LDS, the fragments of the first six k-steps in registers.  The accumulators (+ bias, ReLU, one bf16 rounding) are x0 -- stored
//      -- AND the B fragments of the pointwise product: conv2's weight rows are permuted so that lane group g ends with channels
//      32 h + 8 g .. + 7 (h = 0, 1), which is the natural K order of k_pw1; its 8 weight fragments sit in registers.
//   One barrier per item: the producers have filled buffer k & 1 while the consumers emptied the other one.
// Measured at 20 crops (tools/bench_stem.py, knock-out builds): 28 us against 70 for the three launches; the consumers alone 18, the
// producers alone 22 of which 16 are the input loads (5.6 without them; contiguous addresses instead of the stride-2 pattern: 17).
// Same operand order per output element as the three kernels it replaces: results are bit-identical to theirs.
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

#ifndef STEM_LOOK
#define STEM_LOOK 1
#endif
#ifndef STEM_NRES
#define STEM_NRES 6
#endif
constexpr int TR = 8, TC = 8;                          // x0 tile
constexpr int RH = 2 * TR + 1, RW = 2 * TC + 1;        // c1 positions under it: 17 x 17
constexpr int RP = 20, ODD = 10;                       // slots per c1 row; first slot of the odd columns
constexpr int NPOS = RH * RW, NPT1 = (NPOS + 15) / 16; // 289 positions, 19 wave tiles
constexpr int MAXT1 = (NPT1 + 3) / 4;                  // conv1 wave tiles per producer wave (5, 5, 5, 4)
constexpr int W2B = 9 * 64 * 128;                      // conv2 image
constexpr int REGB = RH * RP * 128;                    // one c1 buffer: 43 520 B
constexpr int LDS_BYTES = W2B + 2 * REGB;              // 73 728 + 2 x 43 520 (+ 768 B of biases behind it)

struct StemFArgs {
    const uint16_t* in; const uint16_t* w1frag; const float* b1; const char* w2img; const float* b2; const char* wp; const float* bp;
    uint16_t* x0; uint16_t* y1;
    int N, H, W, H1, W1, H2, W2, tiles_y, tiles_x, ntiles;
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (RNE) the compiler can see: as an asm
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;            // statement it is not padded against the MFMA that wrote lo / hi
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}

__global__ __launch_bounds__(512, 1) void k_stem_fused(StemFArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // conv2's weight image: 72 pieces of 1 KiB, nine per wave, already in LDS order
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int piece = wave + 8 * i;
        __builtin_amdgcn_global_load_lds((glb_void*)(a.w2img + piece * 1024 + lane * 16), (lds_void*)(smem + piece * 1024), 16, 0, 0);
    }
    // the three bias vectors live in LDS (768 B behind the c1 buffers): a global load inside the item loop would queue behind the next
    // item's input prefetch and stall its consumer for the whole prefetch
    float* const bias_s = (float*)(smem + LDS_BYTES);
    if (tid < 192) bias_s[tid] = tid < 64 ? a.b1[tid] : (tid < 128 ? a.b2[tid - 64] : a.bp[tid - 128]);
    const int per_img = a.tiles_y * a.tiles_x;
    const int mine = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // items of this workgroup: blockIdx.x + k gridDim.x
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // weights and biases are in LDS

    // Iteration k: the producer waves (0-3) compute c1 of item k into buffer k & 1 while the consumer waves (4-7) run conv2 + the
    // pointwise product of item k - 1 out of buffer (k - 1) & 1; one barrier per iteration.  One wave of each kind per SIMD: the
    // consumer's LDS reads and MFMAs beside the producer's global loads, 12-MFMA bursts, conversions and LDS writes.
    // producers = waves 0-3, consumers = waves 4-7 (the other way round: 31.9 -> 29.8 us at 20 crops)
    const int prod = wave < 4, lw_ = wave, cw = wave - 4;
    if (prod) {
        // ================================================ producers: conv1 -> LDS ===============================================
        const int lw = lw_;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * 16), 0x00020000);
        bf16x8 wf1[4][3];                               // A fragments [N tile][tap row]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) wf1[j][ky] = *(const bf16x8*)(a.w1frag + ((size_t)(j * 3 + ky) * 64 + lane) * 8);
        // two sets of input registers: item k + 1's pixels are requested BEFORE item k is computed, so they have a whole iteration to arrive
        // (requested after it, they were needed right behind the barrier: the producers then waited out the full load latency every item)
        bf16x8 xinA[MAXT1][3], xinB[MAXT1][3];
        unsigned okA = 0, okB = 0;                      // bit i: this lane's position of wave tile i lies inside c1
        // conv1's input pixels of item T -> registers (in flight)
        auto issue_loads = [&](int T, bf16x8 (&xin)[MAXT1][3], unsigned& okmask) {
            const int n = T / per_img, r = T - n * per_img, R0 = (r / a.tiles_x) * TR, C0 = (r % a.tiles_x) * TC;
            okmask = 0;
            // all offsets first, then the loads back to back (flags combined with &, offsets computed unconditionally: with short-circuit
            // conditions, or with the address arithmetic between the loads, the compiler serialises them with a vmcnt(0) wait per load)
            unsigned off[MAXT1][3];
#pragma unroll
            for (int i = 0; i < MAXT1; ++i) {
                const int t = lw * MAXT1 + i;            // consecutive wave tiles per wave (interleaved over the waves: +1.7 us)
                const int P = 16 * t + p, ry = (P * 241) >> 12, rx = P - 17 * ry;          // P / 17, exact below 4096
                const int cy = 2 * R0 - 1 + ry, cx = 2 * C0 - 1 + rx;
                const bool ok1 = (P < NPOS) & ((unsigned)cy < (unsigned)a.H1) & ((unsigned)cx < (unsigned)a.W1);
                okmask |= ok1 ? 1u << i : 0u;
                const int ix = 2 * cx + g - 1;
                const bool okx = ok1 & (g < 3) & ((unsigned)ix < (unsigned)a.W);
                const unsigned col = (unsigned)ix * 16u;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = 2 * cy + ky - 1;
                    const unsigned rowb = (unsigned)(n * a.H + iy) * (unsigned)a.W * 16u;
                    off[i][ky] = (okx & ((unsigned)iy < (unsigned)a.H)) ? rowb + col : OOB_OFFSET;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MAXT1; ++i)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    xin[i][ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off[i][ky], 0, 0));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto conv1 = [&](int k, const bf16x8 (&xin)[MAXT1][3], unsigned okmask) {
            char* const region = smem + W2B + (k & 1) * REGB;
#pragma unroll
            for (int i = 0; i < MAXT1; ++i) {
                const int t = lw * MAXT1 + i;            // consecutive wave tiles per wave (interleaved over the waves: +1.7 us)
                if (t < NPT1) {
                    f32x4 acc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = *(const f32x4*)(bias_s + 16 * g + 4 * j);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf1[j][ky]), __builtin_bit_cast(bf16x8_t, xin[i][ky]), acc[j], 0, 0, 0);
                    const int P = 16 * t + p, ry = (P * 241) >> 12, rx = P - 17 * ry;
                    // ReLU on the packed pair (bf16 is sign-magnitude: the same bits as k_conv_stem's fmaxf-then-round for every finite
                    // value), positions outside c1 zeroed by a mask: no branches
                    const uint32_t m = 0u - ((okmask >> i) & 1u);
                    uint32_t d[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        d[2 * j] = relu_bf16x2(pack_bf16x2(acc[j][0], acc[j][1])) & m;
                        d[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc[j][2], acc[j][3])) & m;
                    }
                    if (P < NPOS) {
                        const int slot = ry * RP + ((rx & 1) ? ODD + (rx >> 1) : (rx >> 1)), sw = (slot >> 1) & 7;
                        char* dst = region + slot * 128;
                        *(u32x4*)(dst + (((2 * g) ^ sw) << 4)) = (u32x4){d[0], d[1], d[2], d[3]};
                        *(u32x4*)(dst + (((2 * g + 1) ^ sw) << 4)) = (u32x4){d[4], d[5], d[6], d[7]};
                    }
                }
            }
        };
        // mine + 1 barriers in all.  The steady-state body has NO conditions: with guards around the loads and the convolutions the
        // compiler must assume a set of loads may still be pending where the next ones are set up, and drains the queue there
        const int b0 = (int)blockIdx.x, gs = (int)gridDim.x, last = b0 + (mine - 1) * gs;
        if (mine > 0) issue_loads(b0, xinA, okA);
        int k = 0;
        while (k + 2 <= mine) {
            issue_loads(b0 + (k + 1) * gs, xinB, okB);
            conv1(k, xinA, okA);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_loads(min(b0 + (k + 2) * gs, last), xinA, okA);       // past the end: the last item again (never used)
            conv1(k + 1, xinB, okB);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            k += 2;
        }
        if (k < mine) {
            conv1(k, xinA, okA);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the consumers' last item
        return;
    }

    // ==================================================== consumers: conv2 from LDS, pointwise, stores =============================
    bf16x8 wfp[2][4];                                   // pointwise: A fragments [k-step][N tile] (k_pw1's image)
    {
        const unsigned s3 = (unsigned)p >> 1;
        const unsigned fo0 = (unsigned)p * 128 + (((unsigned)g ^ (s3 & 3)) << 4) + ((s3 >> 2) << 6);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) wfp[h][j] = *(const bf16x8*)(a.wp + j * 2048 + (h ? (fo0 ^ 64u) : fo0));
    }
    // fragment addresses: pixel (2 wave + (p >> 3), p & 7) of the tile
    unsigned boff[9];
    {
        const int base = (2 * (2 * cw + (p >> 3))) * RP + (p & 7);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int slot = base + ky * RP + (kx == 1 ? ODD : (kx == 2 ? 1 : 0));
                boff[ky * 3 + kx] = (unsigned)(W2B + slot * 128 + ((g ^ ((slot >> 1) & 7)) << 4));
            }
    }
    const unsigned aoff = (unsigned)(p * 128 + ((g ^ (p >> 1)) << 4));
    // the weight fragments of the first NRES k-steps stay in registers (the kernel's allocation is set by the producer path: the consumers
    // have them to spare): the K loop is bound by its LDS reads, 4 of every 5 of them weights
    constexpr int NRES = STEM_NRES;
    bf16x8 ares[NRES > 0 ? NRES : 1][4];
#pragma unroll
    for (int s = 0; s < NRES; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) ares[s][j] = *(const bf16x8*)(smem + (s >> 1) * 8192 + j * 2048 + (aoff ^ ((s & 1) ? 64u : 0u)));
    for (int k = 0; k <= mine; ++k) {
        if (k > 0) {
            const int T = (int)blockIdx.x + (k - 1) * (int)gridDim.x;
            const int n = T / per_img, r = T - n * per_img, R0 = (r / a.tiles_x) * TR, C0 = (r % a.tiles_x) * TC;
            const unsigned rb = (unsigned)(((k - 1) & 1) * REGB);
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = *(const f32x4*)(bias_s + 64 + 8 * g + 32 * (j >> 1) + 4 * (j & 1));
            // fragments LOOK k-steps ahead of the MFMAs: a consumer wave is alone with its LDS latency on its SIMD (the producer wave beside
            // it reads almost nothing), one step ahead left the K loop latency-bound
            constexpr int LOOK = STEM_LOOK;
            bf16x8 af[LOOK + 1][4], bfr[LOOK + 1];
            auto ld = [&](int s, bf16x8* a_, bf16x8& b_) {   // k-step s = (tap s >> 1, channels 32 (s & 1) ..)
                const unsigned x = (s & 1) ? 64u : 0u;
                b_ = *(const bf16x8*)(smem + rb + (boff[s >> 1] ^ x));
                if (s >= NRES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a_[j] = *(const bf16x8*)(smem + (s >> 1) * 8192 + j * 2048 + (aoff ^ x));
                }
            };
#pragma unroll
            for (int s = 0; s < LOOK; ++s) ld(s, af[s], bfr[s]);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                const int cur = s % (LOOK + 1), nxt = (s + LOOK) % (LOOK + 1);
                if (s + LOOK < 18) ld(s + LOOK, af[nxt], bfr[nxt]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, s < NRES ? ares[s < NRES ? s : 0][j] : af[cur][j]), __builtin_bit_cast(bf16x8_t, bfr[cur]), acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- x0 out, pointwise product from the accumulators, y1 out
            u32x4 xf[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
                xf[h] = (u32x4){relu_bf16x2(pack_bf16x2(acc[2 * h][0], acc[2 * h][1])), relu_bf16x2(pack_bf16x2(acc[2 * h][2], acc[2 * h][3])),
                                relu_bf16x2(pack_bf16x2(acc[2 * h + 1][0], acc[2 * h + 1][1])), relu_bf16x2(pack_bf16x2(acc[2 * h + 1][2], acc[2 * h + 1][3]))};
            f32x4 accp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) accp[j] = *(const f32x4*)(bias_s + 128 + 16 * g + 4 * j);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    accp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfp[h][j]), __builtin_bit_cast(bf16x8_t, xf[h]), accp[j], 0, 0, 0);
            const int oy = R0 + 2 * cw + (p >> 3), ox = C0 + (p & 7);
            if (oy < a.H2 && ox < a.W2) {
                const size_t pix = ((size_t)n * a.H2 + oy) * a.W2 + ox;
                char* xo = (char*)a.x0 + pix * 128 + g * 16;
                *(u32x4*)xo = xf[0];
                *(u32x4*)(xo + 64) = xf[1];
                uint32_t o[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[2 * j] = relu_bf16x2(pack_bf16x2(accp[j][0], accp[j][1]));
                    o[2 * j + 1] = relu_bf16x2(pack_bf16x2(accp[j][2], accp[j][3]));
                }
                char* yo = (char*)a.y1 + pix * 128 + g * 32;
                *(u32x4*)yo = (u32x4){o[0], o[1], o[2], o[3]};
                *(u32x4*)(yo + 16) = (u32x4){o[4], o[5], o[6], o[7]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

}  // namespace

// Images the host packs (bf16; layouts also in include/pam.h):
//   w1frag [4 N tiles][3 tap rows][64 lanes][8]   k_conv_stem's A fragments (pam_conv2d_nhwc_bf16's w_img of the 8 -> 64 stem layer)
//   w2img  [9 taps][64 rows][64 K]                row 16 j + q of a tap holds output channel 32 (j >> 1) + 8 (q >> 2) + 4 (j & 1) + (q & 3);
//                                                 the row's 16-byte piece at PHYSICAL position pp holds input channels 8 c .. 8 c + 7 with c = pp ^ ((q >> 1) & 7)
//   wp     [64 rows][64 K]                        pam_pointwise64_relu_nhwc_bf16's w_img
extern "C" int pam_stem_fused_nhwc_bf16(void* stream, const void* in, const void* w1frag, const float* bias1, const void* w2img,
                                        const float* bias2, const void* wp_img, const float* biasp, void* out_x0, void* out_y1,
                                        int N, int H, int W) {
    if (!in || !w1frag || !bias1 || !w2img || !bias2 || !wp_img || !biasp || !out_x0 || !out_y1 || N <= 0 || H < 4 || W < 4) return PAM_E_ARG;
    if ((size_t)N * H * W * 16 >= (1ull << 31)) return PAM_E_ARG;
    StemFArgs a;
    a.in = (const uint16_t*)in; a.w1frag = (const uint16_t*)w1frag; a.b1 = bias1; a.w2img = (const char*)w2img; a.b2 = bias2;
    a.wp = (const char*)wp_img; a.bp = biasp; a.x0 = (uint16_t*)out_x0; a.y1 = (uint16_t*)out_y1;
    a.N = N; a.H = H; a.W = W;
    a.H1 = (H - 1) / 2 + 1; a.W1 = (W - 1) / 2 + 1; a.H2 = (a.H1 - 1) / 2 + 1; a.W2 = (a.W1 - 1) / 2 + 1;
    a.tiles_y = (a.H2 + TR - 1) / TR; a.tiles_x = (a.W2 + TC - 1) / TC;
    const long long nt = (long long)N * a.tiles_y * a.tiles_x;
    if (nt >= (1ll << 30)) return PAM_E_ARG;
    a.ntiles = (int)nt;
    if (!pam_max_dynamic_lds((const void*)k_stem_fused, LDS_BYTES + 768)) return PAM_E_HIP;
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
    pam_launch(k_stem_fused, dim3(grid), dim3(512), LDS_BYTES + 768, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
