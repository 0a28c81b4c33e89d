// libpam_hip.so, fused-block part of a1 (round 4): one HRNet BasicBlock of the 48-channel branch
//     out = ReLU( conv3x3( ReLU(conv3x3(x) + b1) ) + b2 + x )            (3x3, stride 1, pad 1, 48 -> 48 -> 48, BN folded)
// per work item with the weights of BOTH convolutions RESIDENT in LDS.  Call site this stands for: the absent HRNet backend inside
// HRNetPose.predict, /root/reference/src/ivclabpose.py:210.
//
// Why a second form of the fused block (k_bblock, csrc/pam_block.hip, streams the weights through a ring with one barrier per two
// k-steps and takes TH full rows per item): at 48 channels the two weight sets are 2 x 42 KB -- they fit LDS beside a 75 KB input
// tile.  With them resident
//   * the K loops need NO barrier and no ring protocol: after the prologue the eight waves of an item run free, two per SIMD, each one's
//     fragment reads / waits / address arithmetic beside its partner's MFMAs (k_bblock's conv phases ran at 60-70 % MFMA issue with
//     every wave meeting at a barrier every 36 MFMAs);
//   * every byte of an item is requested by LDS-DMA in the first microsecond (input tile, bias, both weight sets: no register
//     staging, no ds_write pass), the first convolution starts when the tile and the first four k-steps of W1 have landed;
//   * the tile is 2-D (TR rows x TC columns, e.g. 16 x 36 of a 96 x 72 map), chosen on the host so that the items of a launch fill
//     the chip in whole rounds: 240 items at 20 crops where 8-row full-width tiles gave 180 items on 256 CUs.
// LDS (160 KB, all of it): [X tile: (TR+4) x (TC+4) slots of 96 B][bias 1 KB][W1 42 KB][W2 42 KB].
// Work item:
//   1. X: rows ty0-2 .. ty0+TR+1, columns tx0-2 .. tx0+TC+1 (zeros outside the image: fetched from a page of zeros).
//   2. conv1 on the (TR+2) x (TC+2) positions conv2 needs.  Output slot p <-> window corner at X slot p of the X grid (pitch
//      PWx = TC+4), so the MFMA B fragment of a tap is ONE ds_read_b128 at (p + ky*PWx + kx)*96 + channel bytes: linear in p,
//      conflict-free at a 96-byte pitch.  K = (tap, cin) flattened, 13.5 k-steps of 32 (the half step aliases valid data against
//      zero weights).  A wave keeps 6 x 3 accumulator tiles and walks K once.
//   3. the intermediate (bias, ReLU, bf16; zero outside the image) is packed in registers; barrier; it overwrites X as a grid of
//      pitch PWi = TC+2; barrier.  The residual is re-read from global memory (L2) under conv2's K loop.
//   4. conv2 the same way (5 x 3 tiles per wave), epilogue from the accumulators (+ residual, ReLU, 16 + 8 byte stores).
// Weight rows are permuted so that a lane's 12 output channels are 8 g .. 8 g + 7 and 32 + 4 g .. + 3: one aligned 16-byte and one
// aligned 8-byte access per pixel for the intermediate, the residual and the output.
// Same K order per output element as k_bblock / k_conv3x3<48>: results are bit-identical to theirs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/pam.h"
#include "pam_launch.hpp"

// Diagnostic build only (tools/stamp_block2.py compiles this file with -DPAM_DIAG): per-wave s_memtime stamps into a buffer of their
// own that no kernel reads; the shipped library contains no stamp code.
#ifdef PAM_DIAG
static unsigned long long* g_bb2_stamps = nullptr;
extern "C" int pam_block2_debug_stamps(void* dev_buf) { g_bb2_stamps = (unsigned long long*)dev_buf; return PAM_OK; }
#define BB2_STAMP(k) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB2_STAMP(k) do { } while (0)
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int PA = 96;                     // bytes per activation slot (48 bf16)
constexpr int NST = 14;                    // k-steps per convolution (432 = 13.5 x 32, zero weights in the tail)
constexpr int SUB = 48 * 64;               // one k-step's weight image: [48 rows][64 B], 16-byte pieces swizzled (as k_bblock's)
constexpr int WIMG = NST * SUB;            // 43 008 B per convolution = 42 DMA pieces
constexpr int MW1 = 6, MW2 = 5;            // M tiles (16 slots) per wave, conv1 / conv2
constexpr int TST = 8 * 16 * PA;           // byte distance between a wave's consecutive M tiles (tile = wave + 8 i)
constexpr int WPIECES = 1 + 2 * WIMG / 1024;   // bias piece + both weight images = 85 pieces of 1 KiB
constexpr int XSLOTS_MAX = 800;            // (160 KB - 85 KB) / 96

struct BB2Args {
    const uint16_t* in; const char* wpack; uint16_t* out;
    int N, H, W, TR, TC, tiles_y, tiles_x, nitems, xbytes;
    float inv_pwx, inv_pwi;
#ifdef PAM_DIAG
    unsigned long long* stamps;
#endif
};

__device__ __attribute__((aligned(64))) const uint32_t g_bb2_zero[16] = {0};

__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }   // exact for x < 2^16
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (RNE).  The element-wise cast form compiles to two
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;           // converts + a permute; an asm statement is one instruction too but the compiler
    typedef __attribute__((ext_vector_type(2))) float f32x2;               // does not pad it against the MFMA that wrote lo / hi (csrc/pam_stem.hip met that)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {           // bf16 is sign-magnitude: max(int16, 0) clears the negatives
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}

// issue order of one k-step: the next step's NR fragment reads alternate with the first NR of this step's NM MFMAs
template <int NM, int NR, int... R>
__device__ __forceinline__ void spread(std::integer_sequence<int, R...>) {
    (((void)R, __builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)), ...);
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
}

// One convolution of a wave: MT M tiles x 3 N tiles, K walked once, fragments of k-step st + 1 read under the MFMAs of k-step st.
// wl: this lane's row of k-step 0 in the weight image; xl: this lane's slot of the wave's first M tile; koff[st]: byte offset of the
// lane's 8-channel slice of k-step st (tap shift + channel) in the activation image.  top(st) runs ahead of k-step st's reads (conv1:
// the wait for the rest of W1, and W2's DMAs one per k-step).
template <int MT, typename Top>
__device__ __forceinline__ void conv_pass(f32x4 (&acc)[MW1][3], const char* wl, const char* xl, const unsigned (&koff)[NST], Top top) {
    bf16x8 af[2][3], bf[2][MT];
    auto ld = [&](int st, bf16x8* a_, bf16x8* b_) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a_[j] = *(const bf16x8*)(wl + st * SUB + j * 1024);
#ifdef PAM_KO_PIXREADS                                     // timing knock-out (wrong results): pixel fragments read every third k-step only
        if (st % 3 == 0)
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) b_[i] = *(const bf16x8*)(xl + koff[st] + i * TST);
    };
    ld(0, af[0], bf[0]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        top(st);
        if (st + 1 < NST) ld(st + 1, af[nxt], bf[nxt]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[cur][j]), __builtin_bit_cast(bf16x8_t, bf[cur][i]), acc[i][j], 0, 0, 0);
        spread<3 * MT, MT + 3>(std::make_integer_sequence<int, MT + 3>{});
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(512) void k_bblock2_48(BB2Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware item order: workgroups b, b + 8, ... share an XCD and its L2 -> give each XCD a contiguous run of tiles (neighbouring
    // tiles re-read each other's halo)
    int bx;
    { const int v = blockIdx.x, q = a.nitems >> 3, r = a.nitems & 7, xcd = v & 7, loc = v >> 3; bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = bx / per_img, trem = bx - n * per_img, tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int ty0 = tyi * a.TR, tx0 = txi * a.TC;
    const int PWx = a.TC + 4, PWi = a.TC + 2, XR = a.TR + 4;
    char* Xb = smem;
    char* Bs = smem + a.xbytes;                          // bias piece, then W1, W2: the packed image as it lies in global memory
    char* W1 = Bs + 1024;
    BB2_STAMP(0);

    // ---- every byte of the item by LDS-DMA (up to 1 KiB per wave-instruction), in the order it is needed ---------------------------
    const char* wsrc = a.wpack + lane * 16;
    auto wdma = [&](int p) { __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + p * 1024), (lds_void*)(Bs + p * 1024), 16, 0, 0); };
    for (int p = wave; p < 13; p += 8) wdma(p);          // bias + k-steps 0-3 of W1
    // X row by row (row r of the tile = PWx slots = PWx * 96 contiguous bytes of LDS, and of the image where it is inside): piece k of a
    // row is bytes k KiB .. of it, the same columns in every row -> the per-lane part of the address is computed once per k
    {
        const int rowb = PWx * PA;
        const char* img = (const char*)a.in + (size_t)n * a.H * a.W * PA;
        for (int k = 0; k * 1024 < rowb; ++k) {
            const int ob = k * 1024 + lane * 16;
            const int col = (int)(((unsigned)ob * 43691u) >> 22);         // ob / 96, exact below 2^17
            const int ix = tx0 - 2 + col;
            const bool okx = ob < rowb && (unsigned)ix < (unsigned)a.W;
            const int offx = (tx0 - 2) * PA + ob;                          // byte offset from the start of the image row
            if (ob < rowb) {
                for (int r = wave; r < XR; r += 8) {
                    const int iy = ty0 - 2 + r;
                    const bool ok = okx && (unsigned)iy < (unsigned)a.H;
                    const char* src = ok ? img + (size_t)iy * a.W * PA + offx : (const char*)g_bb2_zero;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + r * rowb + k * 1024), 16, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) wdma(13 + wave + 8 * k);  // the rest of W1 (30 pieces) and the first 2 of W2: 4 per wave, the LAST 4 of every wave

    // per-lane byte offsets of the 14 k-steps in an activation image of pitch PW slots: k0 = 32 st + 8 g = (tap, channel); the upper
    // half of the last k-step (k0 >= 432: zero weights) aliases the lower half's address -- valid, finite data of the same window
    auto mk_koff = [&](int PW, unsigned (&koff)[NST]) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            int k0 = 32 * st + 8 * g;
            if (k0 >= 432) k0 -= 16;
            const int t = k0 / 48, ch = k0 - 48 * t, ky = t / 3, kx = t - 3 * ky;
            koff[st] = (unsigned)((ky * PW + kx) * PA + ch * 2);
        }
    };
    unsigned koff[NST];
    mk_koff(PWx, koff);
    const char* xl = Xb + (wave * 16 + l15) * PA;
    const char* wl = W1 + l15 * 64 + ((g ^ ((0x78 >> ((l15 >> 2) * 2)) & 3)) * 16);     // 0x78 = the piece swizzle sigma = (0, 2, 3, 1), 2 bits each
    const int nt1 = ((a.TR + 2) * PWx + 15) >> 4, nt2 = (a.TR * PWi + 15) >> 4;          // M tiles that carry real slots
    const int mt1 = (nt1 - wave + 7) >> 3, mt2 = (nt2 - wave + 7) >> 3;                    // ... of this wave (tile = wave + 8 i)

    BB2_STAMP(1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // X, bias and k-steps 0-3 of W1 have landed (this wave's share) ...
    asm volatile("s_barrier" ::: "memory");              // ... and everybody's
    BB2_STAMP(2);
    // a lane's 12 output channels: 8 g .. 8 g + 7 (N tiles 0, 1) and 32 + 4 g .. + 3 (N tile 2) -- 16 + 8 aligned bytes of a slot
    const float* bias = (const float*)Bs;

    f32x4 acc[MW1][3];
    // ---- conv1 -------------------------------------------------------------------------------------------------------------------
    {
        const f32x4 b0 = *(const f32x4*)(bias + 8 * g), b1 = *(const f32x4*)(bias + 8 * g + 4), b2 = *(const f32x4*)(bias + 32 + 4 * g);
#pragma unroll
        for (int i = 0; i < MW1; ++i) { acc[i][0] = b0; acc[i][1] = b1; acc[i][2] = b2; }
    }
    // k-step 3: the rest of W1 has landed (nothing younger is in flight yet); k-steps 4-8: one piece of W2 each (40 pieces, 5 per wave),
    // issued beside the MFMAs instead of in front of the first one
    auto top1 = [&](int st) {
        if (st == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        if (st >= 4 && st < 9) wdma(45 + wave + 8 * (st - 4));
    };
    // wave-uniform choice of the instantiation: a wave multiplies only the M tiles that carry real slots (3 at least)
    if (mt1 > 5) conv_pass<6>(acc, wl, xl, koff, top1);
    else if (mt1 == 5) conv_pass<5>(acc, wl, xl, koff, top1);
    else if (mt1 == 4) conv_pass<4>(acc, wl, xl, koff, top1);
    else conv_pass<3>(acc, wl, xl, koff, top1);
    BB2_STAMP(3);

    // intermediate = ReLU(conv1 + b1) as bf16 on a grid of pitch PWi, zero where the position lies outside the image: packed and
    // addressed BEFORE the barrier (a wave that finished early does this beside its SIMD partner's MFMAs), written after it
    uint32_t mid[MW1][6];
    int maddr[MW1];
#ifdef PAM_KO_SERIAL                                       // timing knock-out (WRONG results): what the serial phases between and after the two
    // K loops cost -- the intermediate is neither packed nor written (conv2 multiplies whatever X holds), no barriers, no residual, one
    // store per lane instead of the epilogue.  The bound of any scheme that hides these phases behind another item's MFMAs.
#pragma unroll
    for (int i = 0; i < MW1; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
#pragma unroll
    for (int i = 0; i < MW1; ++i) {
        const int p = (wave + 8 * i) * 16 + l15;
        const int r1 = fdiv_small(p, a.inv_pwx), c1 = p - r1 * PWx;
        const int iy = ty0 - 1 + r1, ix = tx0 - 1 + c1;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        maddr[i] = (r1 < a.TR + 2 && c1 < a.TC + 2) ? (r1 * PWi + c1) * PA : -1;
        const uint32_t keep = ok ? 0xffffffffu : 0u;       // a mask, not a branch per value
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            mid[i][2 * j] = relu_bf16x2(pack_bf16x2(acc[i][j][0], acc[i][j][1])) & keep;
            mid[i][2 * j + 1] = relu_bf16x2(pack_bf16x2(acc[i][j][2], acc[i][j][3])) & keep;
        }
    }
    __syncthreads();                                     // every wave is done reading X (and all DMAs have landed: W2 is complete)
    BB2_STAMP(4);
#pragma unroll
    for (int i = 0; i < MW1; ++i) {
        if (maddr[i] < 0) continue;
        char* d = Xb + maddr[i];
        *(u32x4*)(d + 16 * g) = (u32x4){mid[i][0], mid[i][1], mid[i][2], mid[i][3]};
        *(u32x2*)(d + 64 + 8 * g) = (u32x2){mid[i][4], mid[i][5]};
    }
    __syncthreads();                                     // the intermediate is visible
#endif
    BB2_STAMP(5);

    // ---- conv2 + epilogue ------------------------------------------------------------------------------------------------------------
    // residual = the block's input at this lane's output pixels: plain global loads (L2-hot: the X tile was fetched from there),
    // requested now, used after the K loop
    u32x4 rq[MW2]; u32x2 rh[MW2];
    long ooff[MW2];
#pragma unroll
    for (int i = 0; i < MW2; ++i) {
        const int q = (wave + 8 * i) * 16 + l15;
        const int r2 = fdiv_small(q, a.inv_pwi), c2 = q - r2 * PWi;
        const int oy = ty0 + r2, ox = tx0 + c2;
        const bool ok = r2 < a.TR && c2 < a.TC && oy < a.H && ox < a.W;
        ooff[i] = ok ? (long)((((size_t)n * a.H + oy) * a.W + ox) * 48) : -1;
        const uint16_t* rp = a.in + (ok ? ooff[i] : 0);
        rq[i] = *(const u32x4*)(rp + 8 * g);
        rh[i] = *(const u32x2*)(rp + 32 + 4 * g);
    }
    mk_koff(PWi, koff);
    {
        const f32x4 b0 = *(const f32x4*)(bias + 48 + 8 * g), b1 = *(const f32x4*)(bias + 48 + 8 * g + 4), b2 = *(const f32x4*)(bias + 48 + 32 + 4 * g);
#pragma unroll
        for (int i = 0; i < MW2; ++i) { acc[i][0] = b0; acc[i][1] = b1; acc[i][2] = b2; }
    }
    auto top2 = [](int) {};
    if (mt2 > 4) conv_pass<5>(acc, wl + WIMG, xl, koff, top2);
    else if (mt2 == 4) conv_pass<4>(acc, wl + WIMG, xl, koff, top2);
    else if (mt2 == 3) conv_pass<3>(acc, wl + WIMG, xl, koff, top2);
    else conv_pass<2>(acc, wl + WIMG, xl, koff, top2);
    BB2_STAMP(6);
#ifdef PAM_KO_SERIAL
    {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MW2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { asm volatile("" ::"v"(acc[i][j])); sum += acc[i][j][0]; }
        if (ooff[0] >= 0) a.out[ooff[0]] = (uint16_t)(__builtin_bit_cast(uint32_t, sum) >> 16);
        (void)rq; (void)rh;
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < MW2; ++i) {
        if (ooff[i] < 0) continue;
        const uint32_t rr[6] = {rq[i][0], rq[i][1], rq[i][2], rq[i][3], rh[i][0], rh[i][1]};
        uint32_t ov[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float r0 = __builtin_bit_cast(float, rr[2 * j] << 16), r1 = __builtin_bit_cast(float, rr[2 * j] & 0xffff0000u);
            const float r2f = __builtin_bit_cast(float, rr[2 * j + 1] << 16), r3 = __builtin_bit_cast(float, rr[2 * j + 1] & 0xffff0000u);
            ov[2 * j] = relu_bf16x2(pack_bf16x2(acc[i][j][0] + r0, acc[i][j][1] + r1));
            ov[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc[i][j][2] + r2f, acc[i][j][3] + r3));
        }
        uint16_t* d = a.out + ooff[i];
        *(u32x4*)(d + 8 * g) = (u32x4){ov[0], ov[1], ov[2], ov[3]};
        *(u32x2*)(d + 32 + 4 * g) = (u32x2){ov[4], ov[5]};
    }
    BB2_STAMP(7);
}

// ====================================================================================================================================
// k_bblock2_96: the same fused block for the 96-channel branch (48 x 36 maps).  Its weights (2 x 166 KB) do not fit LDS, so they STREAM
// through a ring of six k-step images (96 rows x 64 B = 6 KB each) while the input tile stays resident:
//   LDS = [X tile, chunk-major: 3 chunks of 32 channels x XSP slots x 64 B, 16-byte pieces swizzled as k_conv3x3s's][bias 1 KB][ring 36 KB]
// Why it pays although a fused 96-channel item streams 332 KB of weights: the unfused layer is LDS-fill-bound (every workgroup of
// k_conv3x3s pulls its weight slab AND its patch for 300 pixels x 48 channels: 156 KB for 25 MFLOP); here one item is up to 12 rows x 36
// columns x all 96 channels x both convolutions (200 MFLOP for 450 KB): the K loops are MFMA-bound, a block costs half the CU-time, one
// launch instead of two, and the intermediate never leaves LDS.
//   * K = (chunk, tap) -- chunk-major, 27 k-steps of 32 per convolution: conv1 starts when the FIRST 32-channel chunk of X has landed.
//   * all eight waves multiply (5 x 6 accumulator tiles for conv1, 4 x 6 for conv2: 11 / 10 fragment reads per 30 / 24 MFMAs); waves 0-3
//     also feed the ring (3 DMA pieces each per two k-steps), waves 4-7 fetch X (all three chunks up front, 30 pieces each, counted vmcnt).
//   * one raw s_barrier per KPB k-steps (KPB = 2, or 4 where the X tile leaves room for a 12-slot ring): at the top of the last k-step of
//     group p everybody is done reading the group, so its slots are refilled with group p + 3; group p + 1 has landed (counted vmcnt: only
//     the youngest group is in flight).  Half-size items spend a third of their K loops at these barriers with KPB = 2.
//   * the weight fragments are single-buffered: the MFMAs of a k-step run N tile by N tile and fragment j of the next k-step is read right
//     behind the 5 MFMAs that used fragment j of this one; the pixel fragments are double-buffered.  (Register plan: 120 accumulators +
//     24 + 40 fragment registers; pinning the issue order with sched_group_barrier made hipcc spill 114 registers, plain program order
//     with one sched_barrier per k-step allocates 240 with no scratch.)
//   * the residual is read from X before the intermediate overwrites it and folded into conv2's accumulators (as k_conv3x3s does).
// ====================================================================================================================================
constexpr int B96_NW = 8;                  // waves per workgroup (two per SIMD)
constexpr int B96_NS = 27;                 // k-steps per convolution
constexpr int B96_KIMG = 96 * 64;          // one k-step's weight image
// M tiles per wave (conv1, conv2) are template parameters: <3, 2> is instantiated (items of up to 6 x 36 positions; a <5, 4> form for 12 x 36 spilled and lost),
constexpr int B96_TST = B96_NW * 16 * 64;  // byte distance between a wave's consecutive M tiles inside a chunk image (tile = wave + 4 i)
constexpr int B96_XSP_MAX = 640;           // X slots per chunk image at most: 10 DMA pieces (16 slots x 64 B) per chunk and loader wave (waves 4-7)

struct BB96Args {
    const uint16_t* in; const char* wpack; uint16_t* out;
    int N, H, W, TR, TC, tiles_y, tiles_x, nitems, xsp;
    float inv_pwx, inv_pwi;
#ifdef PAM_DIAG
    unsigned long long* stamps;
#endif
};
__device__ __attribute__((aligned(256))) const uint32_t g_bb96_zero[64] = {0};

#define B96_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// KPB = k-steps per ring barrier (2 or 4); the ring has 3 * KPB slots: the group being read, the one that has landed, the one in flight
// NPW = X pieces per chunk and loader wave: the item's XSP <= 64 NPW slots (a wave short of a piece re-sends its last one)
template <int MT1, int MT2, int KPB, int NPW>
__global__ __launch_bounds__(64 * B96_NW) void k_bblock2_96(BB96Args a) {
    constexpr int B96_NPW = NPW;
    constexpr int RING = 3 * KPB, NG = (2 * B96_NS + KPB - 1) / KPB;     // ring slots; groups of KPB k-steps in the 54-step weight stream
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx;
    { const int v = blockIdx.x, q = a.nitems >> 3, r = a.nitems & 7, xcd = v & 7, loc = v >> 3; bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = bx / per_img, trem = bx - n * per_img, tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int ty0 = tyi * a.TR, tx0 = txi * a.TC;
    const int PWx = a.TC + 4, PWi = a.TC + 2, XS = (a.TR + 4) * PWx;
    const unsigned CS = (unsigned)a.xsp * 64u;           // bytes of one chunk image
    char* Xb = smem;
    char* Bs = smem + 3 * CS;
    char* ring = Bs + 1024;
#ifdef PAM_DIAG
    const int stamp_wave = wave;
#define B96_STAMP(k) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 8 + stamp_wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define B96_STAMP(k) do { } while (0)
#endif
    B96_STAMP(0);

    // ---- DMA roles -------------------------------------------------------------------------------------------------------------------
    const char* wsrc = a.wpack + lane * 16;
    auto wdma_group = [&](int q) {                       // k-steps KPB q .. KPB q + KPB - 1 of the weight stream -> ring slots (KPB q) % RING ..: waves 0-3, 1.5 KPB pieces each
#pragma unroll
        for (int k = 0; k < 3 * KPB / 2; ++k) {
            const int e = wave + 4 * k;                  // piece of the group (6 per k-step)
            if (KPB * q + e / 6 < 2 * B96_NS)            // the last group of the stream may be short
                __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + 1024 + (size_t)(KPB * q) * B96_KIMG + e * 1024),
                                                 (lds_void*)(ring + ((KPB * q) % RING) * B96_KIMG + e * 1024), 16, 0, 0);
        }
    };
    // X, chunk-major: piece P of a chunk image = slots 16 P .. 16 P + 15, lane -> slot 16 P + (lane >> 2), physical piece lane & 3 holding
    // logical piece (lane & 3) ^ ((slot >> 1) & 2).  The pixel address is the same for the three chunks (+ 64 bytes per chunk).
    const int lw = wave & 3, np = a.xsp >> 4;
    const char* psrc[B96_NPW];
    auto xdma_chunk = [&](int c) {
#pragma unroll
        for (int m = 0; m < B96_NPW; ++m) {
            const int P = min(lw + 4 * m, np - 1);       // a wave short of a piece re-sends the last one (keeps the vmcnt counts fixed)
            __builtin_amdgcn_global_load_lds((glb_void*)(psrc[m] + 64 * c), (lds_void*)(Xb + c * CS + P * 1024), 16, 0, 0);
        }
    };
    if (wave < 4) {
        if (wave == 0) __builtin_amdgcn_global_load_lds((glb_void*)wsrc, (lds_void*)Bs, 16, 0, 0);     // bias
        wdma_group(0); wdma_group(1); wdma_group(2);
    } else {
        const char* img = (const char*)a.in + (size_t)n * a.H * a.W * 192;
#pragma unroll
        for (int m = 0; m < B96_NPW; ++m) {
            const int P = min(lw + 4 * m, np - 1);
            const int slot = 16 * P + (lane >> 2);
            const int gl = (lane & 3) ^ ((slot >> 1) & 2);
            const int row = fdiv_small(slot, a.inv_pwx), col = slot - row * PWx;
            const int iy = ty0 - 2 + row, ix = tx0 - 2 + col;
            const bool ok = slot < XS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            psrc[m] = ok ? img + ((size_t)iy * a.W + ix) * 192 + gl * 16 : (const char*)g_bb96_zero;
        }
        xdma_chunk(0);                                   // the other two chunks: behind the first barrier (they land under the first k-steps)
    }

    // this lane's fragment addresses.  B (activations): slot s of a chunk image at s * 64, logical piece g at g ^ ((s >> 1) & 2); M tile i of
    // the wave = slots (wave + 8 i) * 16 ..: + i * 8 KiB with the same swizzle (bit 2 of the slot unchanged).  A (weights): row j * 16 + l15.
    auto mk_aoff = [&](int PW, unsigned (&aoff)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int sl = wave * 16 + l15 + (t / 3) * PW + (t % 3);
            aoff[t] = (unsigned)(sl * 64 + ((g ^ ((sl >> 1) & 2)) << 4));
        }
    };
    unsigned aoff[9];
    mk_aoff(PWx, aoff);
    const char* wl = ring + l15 * 64 + ((g ^ ((lane >> 1) & 2)) << 4);
    const float* bias = (const float*)Bs;

    B96_STAMP(1);
    if (wave < 4) B96_VM(3 * KPB); else B96_VM(0);       // group 0 of the ring (and the bias) / chunk 0 of X have landed ...
    asm volatile("s_barrier" ::: "memory");              // ... everybody's
    if (wave >= 4) { xdma_chunk(1); xdma_chunk(2); }
    B96_STAMP(2);

    f32x4 acc[MT1][6];
    // One convolution: k-steps S0 .. S0 + 26 of the weight stream, MT M tiles per wave.
    auto conv = [&](auto MTC, auto S0C) {
        constexpr int MT = decltype(MTC)::value, S0 = decltype(S0C)::value;
        bf16x8 af[6], bf[2][MT];
        auto wrow = [&](int s, int j) { return *(const bf16x8*)(wl + (s % RING) * B96_KIMG + j * 1024); };
        auto xfrag = [&](int s, int i) {
            const int st = s - S0, c = st / 9, t = st - 9 * c;
            return *(const bf16x8*)(Xb + c * CS + aoff[t] + i * B96_TST);
        };
#pragma unroll
        for (int j = 0; j < 6; ++j) af[j] = wrow(S0, j);
#pragma unroll
        for (int i = 0; i < MT; ++i) bf[0][i] = xfrag(S0, i);
#pragma unroll
        for (int st = 0; st < B96_NS; ++st) {
            const int s = S0 + st, cur = st & 1, nxt = cur ^ 1;
            const bool more = st + 1 < B96_NS;
            if (s % KPB == KPB - 1) {                    // top of the last k-step of group p: the ring's barrier
                const int p = s / KPB;
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's reads of k-step s (the youngest it has issued) are done
                // everybody is done reading group p -> it is refilled with group p + 3; group p + 1 has landed, only group p + 2 may still fly
                if (wave < 4) { if (p + 2 < NG - 1) B96_VM(3 * KPB / 2); else B96_VM(0); }
                // chunk c of X is first read in k-step 9 c - 1 (fragments of k-step 9 c): waited for at the last barrier at or before that
                else if (S0 == 0 && s <= 8 && s + KPB > 8) B96_VM(B96_NPW);
                else if (S0 == 0 && s <= 17 && s + KPB > 17) B96_VM(0);
                asm volatile("s_barrier" ::: "memory");
                if (wave < 4 && p + 3 < NG) wdma_group(p + 3);
            }
            constexpr int BPG = (MT + 5) / 6;            // pixel-fragment reads of the next k-step per N-tile group
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[j]), __builtin_bit_cast(bf16x8_t, bf[cur][i]), acc[i][j], 0, 0, 0);
                if (more) {
                    af[j] = wrow(s + 1, j);
#pragma unroll
                    for (int k = 0; k < BPG; ++k)
#ifdef PAM_KO_PIXREADS
                        if ((st + 1) % 3 == 0)
#endif
                        if (j * BPG + k < MT) bf[nxt][j * BPG + k] = xfrag(s + 1, j * BPG + k);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- conv1 -------------------------------------------------------------------------------------------------------------------
    // a lane's 24 output channels: 24 g + 4 j + r (N tile j, accumulator element r)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x4 b4 = *(const f32x4*)(bias + 24 * g + 4 * j);
#pragma unroll
        for (int i = 0; i < MT1; ++i) acc[i][j] = b4;
    }
    conv(std::integral_constant<int, MT1>{}, std::integral_constant<int, 0>{});
    B96_STAMP(3);

    // intermediate = ReLU(conv1 + b1) as bf16: three 16-byte pieces per lane and tile (channels 24 g + 8 m .. + 7 -> chunk (24 g + 8 m) / 32,
    // logical piece ((24 g + 8 m) % 32) / 8), packed and addressed before the barrier, written after it
    uint32_t mid[MT1][12];
    int maddr[MT1];
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int p = (wave + B96_NW * i) * 16 + l15;
        const int r1 = fdiv_small(p, a.inv_pwx), c1 = p - r1 * PWx;
        const int iy = ty0 - 1 + r1, ix = tx0 - 1 + c1;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int si = r1 * PWi + c1;
        maddr[i] = (r1 < a.TR + 2 && c1 < a.TC + 2) ? si : -1;
        const uint32_t keep = ok ? 0xffffffffu : 0u;       // a mask, not a branch per value
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            mid[i][2 * j] = relu_bf16x2(pack_bf16x2(acc[i][j][0], acc[i][j][1])) & keep;
            mid[i][2 * j + 1] = relu_bf16x2(pack_bf16x2(acc[i][j][2], acc[i][j][3])) & keep;
        }
    }
    // residual = centre of X at this lane's conv2 output pixels, folded into conv2's accumulators (bias + residual, then the products)
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
        const int q = (wave + B96_NW * i) * 16 + l15;
        const int r2 = fdiv_small(q, a.inv_pwi), c2 = q - r2 * PWi;
        const bool ok = r2 < a.TR && c2 < a.TC;
        const int sx = ok ? (r2 + 2) * PWx + c2 + 2 : 0;
        const unsigned sw = (unsigned)((sx >> 1) & 2);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int ch = 24 * g + 8 * m;
            const u32x4 v = *(const u32x4*)(Xb + (ch >> 5) * CS + sx * 64 + ((((ch & 31) >> 3) ^ sw) << 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 b4 = *(const f32x4*)(bias + 96 + ch + 4 * h);
                acc[i][2 * m + h][0] = b4[0] + __builtin_bit_cast(float, v[2 * h] << 16);
                acc[i][2 * m + h][1] = b4[1] + __builtin_bit_cast(float, v[2 * h] & 0xffff0000u);
                acc[i][2 * m + h][2] = b4[2] + __builtin_bit_cast(float, v[2 * h + 1] << 16);
                acc[i][2 * m + h][3] = b4[3] + __builtin_bit_cast(float, v[2 * h + 1] & 0xffff0000u);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");              // every wave is done reading X (ring DMAs stay in flight: raw barrier)
    B96_STAMP(4);
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        if (maddr[i] < 0) continue;
        const unsigned sw = (unsigned)((maddr[i] >> 1) & 2);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int ch = 24 * g + 8 * m;
            *(u32x4*)(Xb + (ch >> 5) * CS + maddr[i] * 64 + ((((ch & 31) >> 3) ^ sw) << 4)) = (u32x4){mid[i][4 * m], mid[i][4 * m + 1], mid[i][4 * m + 2], mid[i][4 * m + 3]};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");              // the intermediate is visible
    B96_STAMP(5);

    // ---- conv2 + epilogue ------------------------------------------------------------------------------------------------------------
    mk_aoff(PWi, aoff);
    conv(std::integral_constant<int, MT2>{}, std::integral_constant<int, B96_NS>{});
    B96_STAMP(6);
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
        const int q = (wave + B96_NW * i) * 16 + l15;
        const int r2 = fdiv_small(q, a.inv_pwi), c2 = q - r2 * PWi;
        const int oy = ty0 + r2, ox = tx0 + c2;
        if (r2 < a.TR && c2 < a.TC && oy < a.H && ox < a.W) {
            uint16_t* d = a.out + (((size_t)n * a.H + oy) * a.W + ox) * 96 + 24 * g;
#pragma unroll
            for (int m = 0; m < 3; ++m)
                *(u32x4*)(d + 8 * m) = (u32x4){relu_bf16x2(pack_bf16x2(acc[i][2 * m][0], acc[i][2 * m][1])), relu_bf16x2(pack_bf16x2(acc[i][2 * m][2], acc[i][2 * m][3])),
                                               relu_bf16x2(pack_bf16x2(acc[i][2 * m + 1][0], acc[i][2 * m + 1][1])), relu_bf16x2(pack_bf16x2(acc[i][2 * m + 1][2], acc[i][2 * m + 1][3]))};
        }
    }
    B96_STAMP(7);
}

// instantiated (M tiles per wave for conv1, conv2) pairs, smallest first
// one pair: the <5, 4> instantiation (twice the tile, for more than 32 crops) spilled 30 VGPRs and lost to 6 x 36 tiles in two rounds
// (40 crops -1.4 %, 60 crops -0.3 % without it)
static const int kB96Inst[1][2] = {{3, 2}};
int b96_inst(int s1, int s2) {
    for (int k = 0; k < 1; ++k)
        if (s1 <= 16 * B96_NW * kB96Inst[k][0] && s2 <= 16 * B96_NW * kB96Inst[k][1]) return k;
    return -1;
}
// Tile for N x H x W: every item streams all the weights and multiplies the instantiated tile counts whatever it needs of them; the
// cost of a launch = rounds of 256 workgroups x (fixed part + MFMA time of the instantiation), ties -> less halo.
bool pick_tile96(int N, int H, int W, int& TR, int& TC) {
    static thread_local int cN = 0, cH = 0, cW = 0, cTR = 0, cTC = 0;
    if (N == cN && H == cH && W == cW) { TR = cTR; TC = cTC; return true; }
    long best = -1;
    for (int tr = 1; tr <= H; ++tr)
        for (int tc = 1; tc <= W; ++tc) {
            const int s1 = (tr + 2) * (tc + 4), s2 = tr * (tc + 2), xs = (tr + 4) * (tc + 4);
            const int k = b96_inst(s1, s2);
            if (k < 0 || (xs + 15) / 16 * 16 > B96_XSP_MAX) continue;
            const long items = (long)N * ((H + tr - 1) / tr) * ((W + tc - 1) / tc);
            const long per = 6 + kB96Inst[k][0] + kB96Inst[k][1];
            const long cost = ((items + 255) / 256) * per * (1L << 32) + items * (1L << 16) + (long)s1 + s2;
            if (best < 0 || cost < best) { best = cost; TR = tr; TC = tc; }
        }
    if (best < 0) return false;
    // forwards of a few crops (<= 32 items of the largest tile: 4 crops at 48 x 36; at 6 crops = 48 items halving costs what the 32-channel slabs gain): a launch is as long as ONE item, so items of half the
    // size on twice the CUs (round 5: 4 crops -2.0 ... -2.2 % per forward with 3 x 36 / 6 x 18 items; 9 crops = 72 items: +5.7 %)
    const long items0 = (long)N * ((H + TR - 1) / TR) * ((W + TC - 1) / TC);
    if (items0 <= 32 && TR >= 2) {
        const int tr = (TR + 1) / 2, tc = TC;
        const int s1 = (tr + 2) * (tc + 4), s2 = tr * (tc + 2), xs = (tr + 4) * (tc + 4);
        if (b96_inst(s1, s2) >= 0 && (xs + 15) / 16 * 16 <= B96_XSP_MAX) TR = tr;
    }
    cN = N; cH = H; cW = W; cTR = TR; cTC = TC;
    return true;
}

// Tile of the resident-weights block for an N x H x W tensor: the (TR, TC) that minimises rounds of workgroups x time of an item.
// An item's time: a fixed prologue (the X tile and the first weights must land before the first MFMA: ~8 M tiles' worth) plus, per
// convolution, the M tiles of the busiest SIMD (tile t -> wave t % 8 -> SIMD t % 4; a wave multiplies the instantiated count that
// holds its tiles: 3-6 for conv1, 2-5 for conv2).
int simd_tiles(int nt, int lo) {
    int worst = 0;
    for (int s = 0; s < 4; ++s) {
        int sum = 0;
        for (int w = s; w < 8; w += 4) { int m = nt > w ? (nt - w + 7) >> 3 : 0; sum += m < lo ? lo : m; }
        if (sum > worst) worst = sum;
    }
    return worst;
}
bool pick_tile(int N, int H, int W, int& TR, int& TC) {
    static thread_local int cN = 0, cH = 0, cW = 0, cTR = 0, cTC = 0;     // the search is ~H x W steps: keep the last answer
    if (N == cN && H == cH && W == cW) { TR = cTR; TC = cTC; return true; }
    long best = -1;
    for (int tr = 1; tr <= H; ++tr)
        for (int tc = 1; tc <= W; ++tc) {
            const int s1 = (tr + 2) * (tc + 4), s2 = tr * (tc + 2), xs = (tr + 4) * (tc + 4);
            if (s1 > 16 * 8 * MW1 || s2 > 16 * 8 * MW2 || xs > XSLOTS_MAX) continue;
            const long items = (long)N * ((H + tr - 1) / tr) * ((W + tc - 1) / tc);
            const long per = 8 + simd_tiles((s1 + 15) / 16, 3) + simd_tiles((s2 + 15) / 16, 2);
            const long cost = ((items + 255) / 256) * per * 4096 + items;     // whole rounds of 256 workgroups; ties -> fewer items
            if (best < 0 || cost < best) { best = cost; TR = tr; TC = tc; }
        }
    if (best < 0) return false;
    cN = N; cH = H; cW = W; cTR = TR; cTC = TC;
    return true;
}

}  // namespace

extern "C" int pam_basic_block2_tile(int C, int N, int H, int W, int32_t* out2) {
    if ((C != 48 && C != 96) || N < 1 || H < 1 || W < 1 || !out2) return PAM_E_ARG;
    int tr = 0, tc = 0;
    if (!(C == 48 ? pick_tile(N, H, W, tr, tc) : pick_tile96(N, H, W, tr, tc))) return PAM_E_ARG;
    out2[0] = tr; out2[1] = tc;
    return PAM_OK;
}

static int launch_bb96(void* stream, const void* in, const void* wpack, void* out, int N, int H, int W, int tile_rows, int tile_cols) {
    BB96Args a;
    a.in = (const uint16_t*)in; a.wpack = (const char*)wpack; a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W;
    if (tile_rows > 0 && tile_cols > 0) { a.TR = tile_rows; a.TC = tile_cols; }
    else if (!pick_tile96(N, H, W, a.TR, a.TC)) return PAM_E_ARG;
    a.xsp = ((a.TR + 4) * (a.TC + 4) + 15) / 16 * 16;
    const int inst = b96_inst((a.TR + 2) * (a.TC + 4), a.TR * (a.TC + 2));
    if (inst < 0 || a.xsp > B96_XSP_MAX) return PAM_E_ARG;
    a.tiles_y = (H + a.TR - 1) / a.TR; a.tiles_x = (W + a.TC - 1) / a.TC;
    a.nitems = N * a.tiles_y * a.tiles_x;
    a.inv_pwx = 1.0f / (float)(a.TC + 4); a.inv_pwi = 1.0f / (float)(a.TC + 2);
    // junk M tiles of conv1 read up to 2 rows + 2 slots past the last real slot of a chunk image: keep that inside the allocation
    // half-size items with room for it: a 12-slot ring and one barrier per four k-steps
    const int kpb = a.xsp <= 64 * 7 ? 4 : 2;          // 3 x 448 x 64 + 1 KB + 72 KB fits 160 KB
    size_t lds = (size_t)3 * a.xsp * 64 + 1024 + (size_t)3 * kpb * B96_KIMG;
    const size_t reach = (size_t)2 * a.xsp * 64 + (size_t)(16 * B96_NW * kB96Inst[inst][0] + 2 * (a.TC + 4) + 3) * 64;
    if (reach > lds) lds = reach;
    if (lds > 160 * 1024) return PAM_E_ARG;
    if (!pam_max_dynamic_lds((const void*)k_bblock2_96<3, 2, 2, 10>, 160 * 1024) ||
        !pam_max_dynamic_lds((const void*)k_bblock2_96<3, 2, 4, 7>, 160 * 1024)) return PAM_E_HIP;
#ifdef PAM_DIAG
    a.stamps = g_bb2_stamps;
#endif
    if (kpb == 4) pam_launch(k_bblock2_96<3, 2, 4, 7>, dim3(a.nitems), dim3(64 * B96_NW), lds, (hipStream_t)stream, a);
    else pam_launch(k_bblock2_96<3, 2, 2, 10>, dim3(a.nitems), dim3(64 * B96_NW), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_basic_block2_nhwc_bf16(void* stream, const void* in, const void* wpack, void* out, int N, int H, int W, int C, int tile_rows, int tile_cols) {
    if ((C != 48 && C != 96) || !in || !wpack || !out || in == out || N < 1 || H < 1 || W < 1) return PAM_E_ARG;
    if ((size_t)N * H * W * C * 2 >= (1ull << 31)) return PAM_E_ARG;
    if (C == 96) return launch_bb96(stream, in, wpack, out, N, H, W, tile_rows, tile_cols);
    BB2Args a;
    a.in = (const uint16_t*)in; a.wpack = (const char*)wpack; a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W;
    if (tile_rows > 0 && tile_cols > 0) { a.TR = tile_rows; a.TC = tile_cols; }
    else if (!pick_tile(N, H, W, a.TR, a.TC)) return PAM_E_ARG;
    if ((a.TR + 2) * (a.TC + 4) > 16 * 8 * MW1 || a.TR * (a.TC + 2) > 16 * 8 * MW2 || (a.TR + 4) * (a.TC + 4) > XSLOTS_MAX) return PAM_E_ARG;
    a.tiles_y = (H + a.TR - 1) / a.TR; a.tiles_x = (W + a.TC - 1) / a.TC;
    a.nitems = N * a.tiles_y * a.tiles_x;
    a.inv_pwx = 1.0f / (float)(a.TC + 4); a.inv_pwi = 1.0f / (float)(a.TC + 2);
    a.xbytes = ((a.TR + 4) * (a.TC + 4) * PA + 1023) / 1024 * 1024;
    const size_t lds = (size_t)a.xbytes + (size_t)WPIECES * 1024;
    if (!pam_max_dynamic_lds((const void*)k_bblock2_48, 160 * 1024)) return PAM_E_HIP;
#ifdef PAM_DIAG
    a.stamps = g_bb2_stamps;
#endif
    pam_launch(k_bblock2_48, dim3(a.nitems), dim3(512), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
