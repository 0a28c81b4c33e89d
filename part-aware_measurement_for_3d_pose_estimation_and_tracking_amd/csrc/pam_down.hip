// libpam_hip.so, fuse-layer part of a1 (round 5): the 3x3 STRIDE-2 convolutions of HRNet's fuse layers and transitions
//     out = [ReLU on channels >= relu_from]( conv3x3 s2 p1 (x) + b [+ res] )
// Call site these stand for: the absent HRNet backend inside HRNetPose.predict, /root/reference/src/ivclabpose.py:210 (SURVEY.md section 8,
// row a1; the module semantics are hrnet.py:64-100: every output of an HR module sums strided-conv chains from the finer branches).
//
// Until round 5 these 29 layers per forward ran on the generic gather kernel (k_conv_gs): every 16-byte piece of the im2col tile fetched
// from its own address, 13 VALU + SALU instructions per MFMA, 8 % of the MFMA roof and 11 % of the HBM roof, 14.6 us per launch.
//
// k_down48 (this file, Cin = 48: 17 of the 29 layers, among them the largest, 96 x 72 -> 48 x 36):
//   work item = a TR x TC tile of OUTPUT positions of one image; the input patch under it ((2 TR + 1) x (2 TC + 1) pixels of 96 B) is
//   RESIDENT in LDS for the whole item, its columns stored BY PARITY (even patch columns first, then the odd ones): the three taps kx of
//   output column c then read slots c, NE + c, c + 1 -- consecutive slots for consecutive output columns, the same conflict-free
//   ds_read_b128 pattern at a 96-byte pitch as the stride-1 fused block (csrc/pam_block2.hip), and the LDS-DMA does the de-interleaving
//   for free (a DMA piece is linear in LDS but every lane brings its own global address).
//   The item then walks the output channels in slabs of 48: a slab's weights (14 k-step images of [48 rows][64 B] = 42 KB, K = (tap, cin)
//   flattened as in k_bblock2_48) are resident too, DOUBLE-buffered: slab s + 1 lands while slab s is multiplied, so the patch is fetched
//   once per item whatever Cout is (the generic kernel gathered it once per 96-channel slab and tap).  All eight waves multiply (M tiles
//   wave, wave + 8, ... x 3 N tiles), no barrier inside a slab's K loop, ONE barrier per slab; a slab's epilogue (bias is in the
//   accumulators from the start; optional residual, ReLU, 16 + 8 byte stores) runs behind the barrier beside the next slab's first MFMAs
//   of the faster waves.
//   grid = (items, slab groups): layers with few items (12 x 9 outputs: 20 items at 20 crops) spread their slabs over workgroups.
//   Same K order per output element as the generic kernels (k = (ky, kx, cin), 32 per MFMA step, accumulators start from the bias):
//   results are bit-identical to pam_conv2d_nhwc_bf16's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/pam.h"
#include "pam_launch.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int PA = 96;                     // bytes per input slot (48 bf16)
constexpr int NST = 14;                    // k-steps of 32 (432 = 13.5 x 32, zero weights in the tail)
constexpr int SUB = 48 * 64;               // one k-step's weight image: [48 rows][64 B]
constexpr int WIMG = NST * SUB;            // 43 008 B per 48-channel slab = 42 DMA pieces
constexpr int WPW = 6;                     // DMA pieces per wave and slab (42 pieces over 8 waves; a wave short of one re-sends the last)
constexpr int BIASB = 2048;                // float32 bias of the workgroup's slabs (at most 8 x 48)
constexpr int MTMAX = 3;                   // M tiles per wave at most
constexpr int D48_XSLOTS_MAX = (160 * 1024 - 2 * WIMG - BIASB) / PA;      // 789 patch slots

struct D48Args {
    const uint16_t* in; const char* wpack; const float* bias; const uint16_t* res; uint16_t* out;
    int N, H, W, Ho, Wo, in_cs, out_cs, res_cs, relu, relu_from;
    int TR, TC, tiles_y, tiles_x, nitems, spw /* slabs per workgroup */, xbytes;
    float inv_tc;
};

__device__ __attribute__((aligned(64))) const uint32_t g_d48_zero[16] = {0};

__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }   // exact for x < 2^16
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (RNE) the compiler can see (hazard padding)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {           // bf16 is sign-magnitude: max(int16, 0) clears the negatives
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}

// issue order of one k-step: the next step's NR fragment reads alternate with the first NR of this step's NM MFMAs
template <int NM, int NR, int... R>
__device__ __forceinline__ void spread(std::integer_sequence<int, R...>) {
    (((void)R, __builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)), ...);
    if constexpr (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
}

// One slab of a wave: MT M tiles x 3 N tiles, K walked once, fragments of k-step st + 1 read under the MFMAs of k-step st.
// wl: this lane's row of k-step 0 in the slab's weight image; xl[i]: this lane's window corner of M tile i; koff[st]: byte offset of the
// lane's 8-channel slice of k-step st (tap shift in the parity-planar patch + channel).
template <int MT>
__device__ __forceinline__ void slab_pass(f32x4 (&acc)[MTMAX][3], const char* wl, const char* const (&xl)[MTMAX], const unsigned (&koff)[NST]) {
    bf16x8 af[2][3], bf[2][MT];
    auto ld = [&](int st, bf16x8* a_, bf16x8* b_) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a_[j] = *(const bf16x8*)(wl + st * SUB + j * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) b_[i] = *(const bf16x8*)(xl[i] + koff[st]);
    };
    ld(0, af[0], bf[0]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        if (st + 1 < NST) ld(st + 1, af[nxt], bf[nxt]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[cur][j]), __builtin_bit_cast(bf16x8_t, bf[cur][i]), acc[i][j], 0, 0, 0);
        if constexpr (MT >= 2) spread<3 * MT, MT + 3>(std::make_integer_sequence<int, MT + 3>{});
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(512) void k_down48(D48Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware item order: workgroups b, b + 8, ... share an XCD and its L2 -> every XCD gets a contiguous run of tiles (neighbouring
    // tiles re-read each other's halo row / column)
    int bx;
    { const int v = blockIdx.x, q = a.nitems >> 3, r = a.nitems & 7, xcd = v & 7, loc = v >> 3; bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = bx / per_img, trem = bx - n * per_img, tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int ty0 = tyi * a.TR, tx0 = txi * a.TC;                    // output tile origin
    const int NE = a.TC + 1, PWx = 2 * a.TC + 1, XR = 2 * a.TR + 1;    // even patch columns, slots per patch row, patch rows
    const int slab0 = (int)blockIdx.y * a.spw;                          // first 48-channel slab of this workgroup
    char* Xb = smem;
    float* bias_s = (float*)(smem + a.xbytes);
    char* Wb = smem + a.xbytes + BIASB;

    // ---- every byte by LDS-DMA, in the order it is needed: slab 0's weights and the patch, then slab 1's weights ----------------------
    const float bv = (tid < a.spw * 48 && a.bias) ? a.bias[slab0 * 48 + tid] : 0.0f;
    const char* wsrc = a.wpack + (size_t)slab0 * WIMG + lane * 16;
    auto wdma = [&](int s) {                               // slab s of this workgroup -> buffer s & 1: WPW pieces per wave
        char* dst = Wb + (s & 1) * WIMG;
#pragma unroll
        for (int k = 0; k < WPW; ++k) {
            const int p = min(wave + 8 * k, WIMG / 1024 - 1);
            __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + (size_t)s * WIMG + p * 1024), (lds_void*)(dst + p * 1024), 16, 0, 0);
        }
    };
    wdma(0);
    // patch row r = PWx slots = PWx * 96 contiguous bytes of LDS; piece k of a row is bytes k KiB .. of it -- the same slots in every row,
    // so a lane's column arithmetic (slot -> parity plane -> image column) is done once per k
    {
        const int rowb = PWx * PA;
        const char* img = (const char*)a.in + (size_t)n * a.H * a.W * a.in_cs * 2;
        for (int k = 0; k * 1024 < rowb; ++k) {
            const int ob = k * 1024 + lane * 16;
            const int q = (int)(((unsigned)ob * 43691u) >> 22);           // ob / 96, exact below 2^17
            const int rem = ob - q * PA;
            const int c = q < NE ? 2 * q : 2 * (q - NE) + 1;               // patch column of slot q
            const int ix = 2 * tx0 - 1 + c;
            const bool okx = ob < rowb && (unsigned)ix < (unsigned)a.W;
            const long offx = (long)ix * a.in_cs * 2 + rem;                // byte offset from the start of the image row
            if (ob < rowb) {
                for (int r = wave; r < XR; r += 8) {
                    const int iy = 2 * ty0 - 1 + r;
                    const bool ok = okx && (unsigned)iy < (unsigned)a.H;
                    const char* src = ok ? img + (size_t)iy * a.W * a.in_cs * 2 + offx : (const char*)g_d48_zero;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + r * rowb + k * 1024), 16, 0, 0);
                }
            }
        }
    }
    if (tid < a.spw * 48) bias_s[tid] = bv;
    if (a.spw > 1) {
        wdma(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");       // all but slab 1's pieces: this wave's share of the patch and of slab 0
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // per-lane byte offsets of the 14 k-steps: k0 = 32 st + 8 g = (tap, channel); tap (ky, kx) of the window whose corner is the even
    // slot c: slots c, NE + c, c + 1 of patch row 2 r + ky.  The upper half of the last k-step (k0 >= 432: zero weights) aliases the
    // lower half's address -- valid, finite data of the same window
    unsigned koff[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        int k0 = 32 * st + 8 * g;
        if (k0 >= 432) k0 -= 16;
        const int t = k0 / 48, ch = k0 - 48 * t, ky = t / 3, kx = t - 3 * ky;
        koff[st] = (unsigned)((ky * PWx + (kx == 1 ? NE : (kx == 2 ? 1 : 0))) * PA + ch * 2);
    }
    const int npos = a.TR * a.TC;                                       // output positions of a full tile
    const int ntile = (npos + 15) >> 4, mt = (ntile - wave + 7) >> 3;   // M tiles that carry positions / of this wave (tile = wave + 8 i)
    const char* xl[MTMAX];
    long ooff[MTMAX];                                                   // element offset of the lane's output pixel (channel 0), -1 = none
#pragma unroll
    for (int i = 0; i < MTMAX; ++i) {
        const int m = (wave + 8 * i) * 16 + l15;
        const int r2 = fdiv_small(m, a.inv_tc), c2 = m - r2 * a.TC;
        const bool in_tile = m < npos;
        xl[i] = Xb + (in_tile ? (2 * r2 * PWx + c2) * PA : 0);
        const int oy = ty0 + r2, ox = tx0 + c2;
        ooff[i] = (in_tile && oy < a.Ho && ox < a.Wo) ? (long)(((size_t)n * a.Ho + oy) * a.Wo + ox) : -1;
    }
    const int sw = (0x78 >> ((l15 >> 2) * 2)) & 3;                      // the weight images' piece swizzle sigma = (0, 2, 3, 1)
    const unsigned wlo = (unsigned)(l15 * 64 + ((g ^ sw) * 16));

    asm volatile("s_barrier" ::: "memory");                              // patch, bias and slab 0 are in LDS (everybody's share)

    for (int s = 0; s < a.spw; ++s) {
        const int cb = (slab0 + s) * 48;                                 // first output channel of the slab
        // a lane's 12 output channels: cb + 8 g .. + 7 (N tiles 0, 1) and cb + 32 + 4 g .. + 3 (N tile 2)
        u32x4 rq[MTMAX]; u32x2 rh[MTMAX];
        if (a.res) {                                                      // residual rows: requested now, used behind the K loop
#pragma unroll
            for (int i = 0; i < MTMAX; ++i) {
                const uint16_t* rp = a.res + (ooff[i] >= 0 ? ooff[i] * a.res_cs + cb : 0);
                rq[i] = *(const u32x4*)(rp + 8 * g);
                rh[i] = *(const u32x2*)(rp + 32 + 4 * g);
            }
        }
        f32x4 acc[MTMAX][3];
        {
            const float* b = bias_s + s * 48;
            const f32x4 b0 = *(const f32x4*)(b + 8 * g), b1 = *(const f32x4*)(b + 8 * g + 4), b2 = *(const f32x4*)(b + 32 + 4 * g);
#pragma unroll
            for (int i = 0; i < MTMAX; ++i) { acc[i][0] = b0; acc[i][1] = b1; acc[i][2] = b2; }
        }
        const char* wl = Wb + (s & 1) * WIMG + wlo;
        // wave-uniform choice of the instantiation: a wave multiplies only the M tiles that carry positions
        if (mt >= 3) slab_pass<3>(acc, wl, xl, koff);
        else if (mt == 2) slab_pass<2>(acc, wl, xl, koff);
        else if (mt == 1) slab_pass<1>(acc, wl, xl, koff);
        if (s + 1 < a.spw) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of slab s + 1 (issued a slab ago) have landed
            asm volatile("s_barrier" ::: "memory");                       // everybody is done reading buffer s & 1; slab s + 1 is complete
            if (s + 2 < a.spw) wdma(s + 2);
        }
        // ---- epilogue of the slab, straight from the accumulators
        const bool relu_lo = a.relu && cb + 8 * g >= a.relu_from, relu_hi = a.relu && cb + 32 + 4 * g >= a.relu_from;
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) {
            if (i >= mt || ooff[i] < 0) continue;
            uint32_t ov[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (a.res) {
                    const uint32_t r01 = j < 2 ? rq[i][2 * j] : rh[i][0], r23 = j < 2 ? rq[i][2 * j + 1] : rh[i][1];
                    v0 += __builtin_bit_cast(float, r01 << 16); v1 += __builtin_bit_cast(float, r01 & 0xffff0000u);
                    v2 += __builtin_bit_cast(float, r23 << 16); v3 += __builtin_bit_cast(float, r23 & 0xffff0000u);
                }
                ov[2 * j] = pack_bf16x2(v0, v1); ov[2 * j + 1] = pack_bf16x2(v2, v3);
                if (j < 2 ? relu_lo : relu_hi) { ov[2 * j] = relu_bf16x2(ov[2 * j]); ov[2 * j + 1] = relu_bf16x2(ov[2 * j + 1]); }
            }
            uint16_t* d = a.out + ooff[i] * a.out_cs + cb;
            *(u32x4*)(d + 8 * g) = (u32x4){ov[0], ov[1], ov[2], ov[3]};
            *(u32x2*)(d + 32 + 4 * g) = (u32x2){ov[4], ov[5]};
        }
    }
}

// worst number of M tiles on one SIMD (tile t -> wave t % 8; waves w and w + 4 share a SIMD)
int d48_simd_tiles(int ntile) {
    int worst = 0;
    for (int s = 0; s < 4; ++s) {
        int sum = 0;
        for (int w = s; w < 8; w += 4) sum += ntile > w ? (ntile - w + 7) >> 3 : 0;
        if (sum > worst) worst = sum;
    }
    return worst;
}

// Tile and slab grouping of a layer: the cheapest (rounds of 256 workgroups) x (time of a workgroup), where a workgroup's time is its
// LDS fill (patch once + its slabs' weights, ~24 B/clk/CU) against its MFMA time on the fullest SIMD, plus a fixed part.
bool d48_pick(int N, int Ho, int Wo, int nslab, int& TR, int& TC, int& G) {
    static thread_local int memo[8][7];                                 // the search is ~Ho x Wo x divisors steps: keep the last answers
    static thread_local int memo_n = 0;
    for (int k = 0; k < 8 && k < memo_n; ++k)
        if (memo[k][0] == N && memo[k][1] == Ho && memo[k][2] == Wo && memo[k][3] == nslab) { TR = memo[k][4]; TC = memo[k][5]; G = memo[k][6]; return true; }
    long best = -1;
    for (int tr = 1; tr <= Ho; ++tr)
        for (int tc = 1; tc <= Wo; ++tc) {
            const int xs = (2 * tr + 1) * (2 * tc + 1), nt = (tr * tc + 15) / 16;
            if (xs > D48_XSLOTS_MAX || nt > 8 * MTMAX) continue;
            const long items = (long)N * ((Ho + tr - 1) / tr) * ((Wo + tc - 1) / tc);
            for (int gsz = 1; gsz <= nslab; ++gsz) {
                if (nslab % gsz) continue;
                const int spw = nslab / gsz;
                if (spw * 48 * 4 > BIASB) continue;
                const long fill = ((long)xs * PA + (long)spw * WIMG) / 24;
                const long mfma = (long)d48_simd_tiles(nt) * 3 * NST * 16 * spw;
                const long t = 5000 + (fill > mfma ? fill + mfma / 4 : mfma + fill / 4) + 500L * spw;
                const long wgs = items * gsz;
                const long cost = ((wgs + 255) / 256) * t * 4096 + wgs;
                if (best < 0 || cost < best) { best = cost; TR = tr; TC = tc; G = gsz; }
            }
        }
    if (best < 0) return false;
    int* m = memo[memo_n++ & 7];
    m[0] = N; m[1] = Ho; m[2] = Wo; m[3] = nslab; m[4] = TR; m[5] = TC; m[6] = G;
    return true;
}


// ====================================================================================================================================
// k_down_s (Cin = 96 / 192: the other 12 strided layers): the streamed 3x3 kernel of csrc/pam_conv.hip (k_conv3x3s: loader waves 4-7 move
// 32-channel chunks -- patch rows + the chunk's [9 taps][BN rows] weight image -- by LDS-DMA into a ring of chunk buffers, multiplier
// waves 0-3 read fragments and multiply) with a STRIDE-2 patch: tile = TH output rows x the full output width, patch = (2 TH + 1) rows of
// PWp = 2 Wo + 2 slots of 64 B, the columns of a row stored by parity (even patch columns in slots 0 .. Wo, odd ones in Wo + 1 .. 2 Wo), so
// the taps kx = 0, 1, 2 of output column c are slots c, Wo + 1 + c, c + 1 and a 16-position M tile reads (runs of) consecutive slots:
// the conflict-free ds_read_b128 pattern of the stride-1 kernel (piece g of slot r at position g ^ ((r >> 1) & 2)).  The loaders'
// per-lane source addresses do the de-interleaving (and take the channel stride of a sliced input); the weight images are the
// stride-1 kernel's ([slab][chunk][tap][row][4 pieces], rows permuted so that a lane ends with 4 NTW contiguous channels).
// The generic gather kernel spent 13 VALU + SALU instructions per MFMA on im2col addresses; here a multiplier lane holds MT x 9 offsets.
// ====================================================================================================================================
struct DSArgs {
    const uint16_t* in; const uint16_t* wimg; const float* bias; uint16_t* out;
    int N, H, W, Ho, Wo, in_cs, Cout, TH, tiles_y, relu, relu_from;
    float inv_pwp, inv_wo;
};

template <int NTW>
__device__ __forceinline__ void ds_row_store(uint16_t* p, int g, const uint32_t* d) {
    if constexpr (NTW == 4) {
        *(u32x4*)p = (u32x4){d[0], d[1], d[2], d[3]}; *(u32x4*)(p + 8) = (u32x4){d[4], d[5], d[6], d[7]};
    } else {
        static_assert(NTW == 3, "slab width");            // 24 bytes: 16 + 8 for even g, 8 + 16 for odd g (the 16-byte half stays aligned)
        const bool odd = g & 1;
        *(u32x4*)(p + (odd ? 4 : 0)) = odd ? (u32x4){d[2], d[3], d[4], d[5]} : (u32x4){d[0], d[1], d[2], d[3]};
        *(u32x2*)(p + (odd ? 0 : 8)) = odd ? (u32x2){d[0], d[1]} : (u32x2){d[4], d[5]};
    }
}

template <int CIN, int NTW, int MT, int PMAX, int NBUF>
__global__ __launch_bounds__(512, 1) void k_down_s(DSArgs a) {
    constexpr int BN = 16 * NTW, NCHUNK = CIN / 32;
    constexpr int PIMG = PMAX * 64, WIMGS = 9 * BN * 64, BUF = PIMG + WIMGS;
    constexpr int PPW = PMAX / 64, WPIECES = WIMGS / 1024, WPWS = (WPIECES + 3) / 4, NPER = PPW + WPWS;   // DMA pieces per loader wave per chunk
    static_assert(PMAX % 64 == 0 && WIMGS % 1024 == 0 && NPER * (NBUF - 1) <= 60 && NBUF >= 2 && NBUF <= 3, "ring shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.tiles_y * a.N;
    const int bx = [&] {                                // XCD-aware tile order: every XCD gets a contiguous run of tiles
        const int v = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, loc = v >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }();
    const int n = bx / a.tiles_y, ty0 = (bx - n * a.tiles_y) * a.TH;                     // first OUTPUT row of the tile
    const int NEp = a.Wo + 1, PWp = 2 * a.Wo + 2, npatch = (2 * a.TH + 1) * PWp;
    const int npos = min(a.TH, a.Ho - ty0) * a.Wo;                                       // output positions of this tile
    const int n0 = blockIdx.y * BN;

    if (wave >= 4) {
        // ---- loader waves ---------------------------------------------------------------------------------------------------
        const int lw = wave - 4;
        const char* psrc[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int s = (lw + 4 * i) * 16 + (lane >> 2);                               // slot this lane fills in piece lw + 4 i
            const int gsrc = (lane & 3) ^ ((s >> 1) & 2);
            const int py = fdiv_small(s, a.inv_pwp), q = s - py * PWp;
            const int c = q < NEp ? 2 * q : 2 * (q - NEp) + 1;                            // patch column of slot q (q = 2 Wo + 1: the pad slot)
            const int iy = 2 * ty0 - 1 + py, ix = c - 1;
            const bool ok = s < npatch && q <= 2 * a.Wo && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            psrc[i] = ok ? (const char*)a.in + (((size_t)n * a.H + iy) * a.W + ix) * (size_t)(a.in_cs * 2) + gsrc * 16 : (const char*)g_d48_zero;
        }
        const char* wsrc = (const char*)a.wimg + (size_t)blockIdx.y * NCHUNK * WIMGS + lane * 16;
        auto issue = [&](int c) {
            char* dst = smem + (size_t)((unsigned)c % (unsigned)NBUF) * BUF;
#pragma unroll
            for (int i = 0; i < (PPW > WPWS ? PPW : WPWS); ++i) {
                if (i < WPWS) {
                    const int j = min(lw + 4 * i, WPIECES - 1);                          // a wave short of a piece re-sends the last one
                    __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + (size_t)c * WIMGS + j * 1024), (lds_void*)(dst + PIMG + j * 1024), 16, 0, 0);
                }
                if (i < PPW) {
                    const bool z = psrc[i] == (const char*)g_d48_zero;                   // padding / outside the image: the page of zeros, whatever the chunk
                    __builtin_amdgcn_global_load_lds((glb_void*)(psrc[i] + (z ? 0 : c * 64)), (lds_void*)(dst + (lw + 4 * i) * 1024), 16, 0, 0);
                }
            }
        };
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c)
            if (c < NCHUNK) issue(c);
        for (int k = 0; k < NCHUNK; ++k) {
            const int fly = min(NCHUNK - 1 - k, NBUF - 2);                               // younger chunks that may stay in flight
            if (fly <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
            asm volatile("s_barrier" ::: "memory");
            if (k + NBUF - 1 < NCHUNK) issue(k + NBUF - 1);
        }
        return;
    }

    // ---- multiplier waves ---------------------------------------------------------------------------------------------------
    f32x4 acc[MT][NTW];
    {
        f32x4 bias4[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) bias4[j] = a.bias ? *(const f32x4*)(a.bias + n0 + g * 4 * NTW + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = bias4[j];
    }
    // LDS byte offsets (inside a chunk buffer) of this lane's patch fragment for M tile i and tap t, swizzle included
    unsigned aoff[MT][9];
    int opix[MT];                                                                        // output pixel of the lane's position (-1: none)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wave * MT + i) * 16 + (lane & 15);
        const int mm = m < npos ? m : 0;
        const int oy = fdiv_small(mm, a.inv_wo), ox = mm - oy * a.Wo;
        opix[i] = m < npos ? ((n * a.Ho + ty0 + oy) * a.Wo + ox) : -1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int s = (2 * oy + ky) * PWp + ox + (kx == 1 ? NEp : (kx == 2 ? 1 : 0));
            aoff[i][t] = (unsigned)(s * 64 + ((g ^ ((s >> 1) & 2)) << 4));
        }
    }
    const unsigned woff = (unsigned)(PIMG + (lane & 15) * 64 + ((g ^ ((lane >> 1) & 2)) << 4));   // row j*16 + (lane & 15): bit 2 of the row = bit 2 of the lane

    bf16x8 af[2][MT], bfr[2][NTW];
    auto ldfrag = [&](int k, int t, bf16x8* af_, bf16x8* bf_) {
        const char* buf = smem + (size_t)((unsigned)k % (unsigned)NBUF) * BUF;
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf_[j] = *(const bf16x8*)(buf + woff + (t * BN + j * 16) * 64);
#pragma unroll
        for (int i = 0; i < MT; ++i) af_[i] = *(const bf16x8*)(buf + aoff[i][t]);
    };
    auto chunk = [&](int k, auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int cur = (t + PAR) & 1, nxt = cur ^ 1;
            if (t + 1 < 9) ldfrag(k, t + 1, af[nxt], bfr[nxt]);
            else if (k + 1 < NCHUNK) {
                __builtin_amdgcn_s_waitcnt(0xC07F);                                       // lgkmcnt(0): this wave is done reading chunk k
                asm volatile("s_barrier" ::: "memory");                                   // chunk k + 1 has landed; chunk k's buffer is free
                ldfrag(k + 1, 0, af[nxt], bfr[nxt]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[cur][j]),
                                                                       __builtin_bit_cast(bf16x8_t, af[cur][i]), acc[i][j], 0, 0, 0);
            spread<MT * NTW, MT + NTW>(std::make_integer_sequence<int, MT + NTW>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    asm volatile("s_barrier" ::: "memory");                                              // chunk 0 has landed (and is visible)
    ldfrag(0, 0, af[0], bfr[0]);
    for (int k = 0; k < NCHUNK; k += 2) {
        chunk(k, std::integral_constant<int, 0>{});
        if (k + 1 < NCHUNK) chunk(k + 1, std::integral_constant<int, 1>{});
    }
    // ---- epilogue straight from the accumulators: 4 NTW contiguous channels of one pixel per lane
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (opix[i] < 0) continue;
        uint32_t ov[2 * NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            ov[2 * j] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); ov[2 * j + 1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
            if (a.relu && n0 + g * 4 * NTW + j * 4 >= a.relu_from) { ov[2 * j] = relu_bf16x2(ov[2 * j]); ov[2 * j + 1] = relu_bf16x2(ov[2 * j + 1]); }
        }
        ds_row_store<NTW>(a.out + (size_t)opix[i] * a.Cout + n0 + g * 4 * NTW, g, ov);
    }
}

// rows per tile and the instantiation: M tiles per multiplier wave x patch slots.  false = this layer stays on the generic kernels
bool ds_pick(int Ho, int Wo, int Cin, int Cout, int& TH, int& bn) {
    if (Cin != 96 && Cin != 192) return false;
    bn = Cout % 64 == 0 ? 64 : (Cout % 48 == 0 ? 48 : 0);
    if (!bn) return false;
    const int PWp = 2 * Wo + 2;
    TH = 0;
    long best = 0;
    for (int t = Ho; t >= 1; --t) {                     // the tile height that costs the fewest workgroups x M tiles; at most 8 M tiles, 512 slots
        if (t * Wo > 128 || (2 * t + 1) * PWp > 512) continue;
        const long cost = (Ho + t - 1) / t;
        if (TH == 0 || cost < best) { TH = t; best = cost; }
    }
    return TH >= 1;
}
template <int CIN, int NTW>
int launch_ds(hipStream_t s, const DSArgs& a) {
    constexpr int MT = 2, PMAX = 512, NBUF = 2;
    constexpr size_t lds = (size_t)NBUF * (PMAX * 64 + 9 * 16 * NTW * 64);
    static_assert(lds <= 160 * 1024, "LDS");
    if (!pam_max_dynamic_lds((const void*)k_down_s<CIN, NTW, MT, PMAX, NBUF>, (int)lds)) return PAM_E_HIP;
    pam_launch(k_down_s<CIN, NTW, MT, PMAX, NBUF>, dim3(a.tiles_y * a.N, a.Cout / (16 * NTW)), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

}  // namespace

// wpack: Cout / 48 slab images of 43 008 bytes, slab s = output channels 48 s .. 48 s + 47: [14 k-steps][48 rows][64 bytes] exactly as one
// convolution of pam_basic_block2_nhwc_bf16's C = 48 image (include/pam.h): k-step = 32 K elements of the flattened (tap, cin) index, zero
// tail; row j*16 + q = channel 8*(q >> 2) + 4*j + (q & 3) of the slab for j < 2 and 32 + 4*(q >> 2) + (q & 3) for j = 2; physical
// 16-byte piece p of row R holds K elements 8*(p ^ s) .. + 7 with s = (0,2,3,1)[(R % 16) >> 2].
extern "C" int pam_conv3x3s2_c48_tile(int N, int H, int W, int Cout, int32_t* out3) {
    if (N < 1 || H < 2 || W < 2 || Cout < 48 || Cout % 48 != 0 || !out3) return PAM_E_ARG;
    int tr = 0, tc = 0, gs = 0;
    if (!d48_pick(N, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Cout / 48, tr, tc, gs)) return PAM_E_ARG;
    out3[0] = tr; out3[1] = tc; out3[2] = gs;
    return PAM_OK;
}

extern "C" int pam_conv3x3s2_c48_nhwc_bf16(void* stream, const void* in, int in_cstride, const void* wpack, const float* bias,
                                           const void* res, int res_cstride, void* out, int out_cstride, int N, int H, int W, int Cout,
                                           int relu, int relu_from, int tile_rows, int tile_cols, int slab_groups) {
    if (!in || !wpack || !out || N < 1 || H < 2 || W < 2 || Cout < 48 || Cout % 48 != 0) return PAM_E_ARG;
    if (in_cstride < 48 || in_cstride % 8 != 0 || out_cstride < Cout || out_cstride % 8 != 0 || relu_from < 0 || relu_from % 8 != 0) return PAM_E_ARG;
    if (res && (res_cstride < Cout || res_cstride % 8 != 0)) return PAM_E_ARG;
    D48Args a;
    a.in = (const uint16_t*)in; a.wpack = (const char*)wpack; a.bias = bias; a.res = (const uint16_t*)res; a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.in_cs = in_cstride; a.out_cs = out_cstride; a.res_cs = res ? res_cstride : 0; a.relu = relu ? 1 : 0; a.relu_from = relu_from;
    if ((size_t)N * H * W * in_cstride * 2 >= (1ull << 40)) return PAM_E_ARG;
    const int nslab = Cout / 48;
    int G = slab_groups;
    if (tile_rows > 0 && tile_cols > 0) {
        a.TR = tile_rows; a.TC = tile_cols;
        if (G <= 0) G = 1;
    } else {
        int g2 = 1;
        if (!d48_pick(N, a.Ho, a.Wo, nslab, a.TR, a.TC, g2)) return PAM_E_ARG;
        if (G <= 0) G = g2;
    }
    if (a.TR > a.Ho) a.TR = a.Ho;
    if (a.TC > a.Wo) a.TC = a.Wo;
    if (nslab % G != 0) return PAM_E_ARG;
    a.spw = nslab / G;
    const int xs = (2 * a.TR + 1) * (2 * a.TC + 1);
    if (xs > D48_XSLOTS_MAX || (a.TR * a.TC + 15) / 16 > 8 * MTMAX || a.spw * 48 * 4 > BIASB) return PAM_E_ARG;
    a.xbytes = (xs * PA + 1023) / 1024 * 1024;
    a.tiles_y = (a.Ho + a.TR - 1) / a.TR; a.tiles_x = (a.Wo + a.TC - 1) / a.TC;
    const long long items = (long long)N * a.tiles_y * a.tiles_x;
    if (items >= (1ll << 30)) return PAM_E_ARG;
    a.nitems = (int)items;
    a.inv_tc = 1.0f / (float)a.TC;
    const size_t lds = (size_t)a.xbytes + BIASB + (size_t)(a.spw > 1 ? 2 : 1) * WIMG;
    if (!pam_max_dynamic_lds((const void*)k_down48, 160 * 1024)) return PAM_E_HIP;
    pam_launch(k_down48, dim3(a.nitems, G), dim3(512), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// Cin = 96 / 192 on the streamed kernel with a stride-2 patch (k_down_s).  w_img: the streamed 3x3 image of pam_conv2d_nhwc_bf16
// ([Cout / BN][Cin / 32][9 taps][BN rows][4 pieces of 8 bf16], see there) for BN = pam_conv3x3s2_slab(); 0 = this shape is not taken.
extern "C" int pam_conv3x3s2_slab(int H, int W, int Cin, int Cout) {
    int th = 0, bn = 0;
    if (H < 2 || W < 2) return 0;
    return ds_pick((H - 1) / 2 + 1, (W - 1) / 2 + 1, Cin, Cout, th, bn) ? bn : 0;
}
extern "C" int pam_conv3x3s2_nhwc_bf16(void* stream, const void* in, int in_cstride, const void* w_img, const float* bias, void* out,
                                       int N, int H, int W, int Cin, int Cout, int relu, int relu_from) {
    if (!in || !w_img || !out || N < 1 || H < 2 || W < 2 || in_cstride < Cin || in_cstride % 8 != 0 || relu_from < 0 || relu_from % 16 != 0) return PAM_E_ARG;
    DSArgs a;
    a.in = (const uint16_t*)in; a.wimg = (const uint16_t*)w_img; a.bias = bias; a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.in_cs = in_cstride; a.Cout = Cout;
    a.relu = relu ? 1 : 0; a.relu_from = relu_from;
    int bn = 0;
    if (!ds_pick(a.Ho, a.Wo, Cin, Cout, a.TH, bn)) return PAM_E_ARG;
    if ((size_t)N * a.Ho * a.Wo >= (1ull << 31) || (size_t)N * H * W * in_cstride * 2 >= (1ull << 40)) return PAM_E_ARG;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    a.inv_pwp = 1.0f / (float)(2 * a.Wo + 2); a.inv_wo = 1.0f / (float)a.Wo;
    hipStream_t s = (hipStream_t)stream;
    switch (Cin * 10 + bn / 16) {
        case 964: return launch_ds<96, 4>(s, a);
        case 963: return launch_ds<96, 3>(s, a);
        case 1924: return launch_ds<192, 4>(s, a);
        case 1923: return launch_ds<192, 3>(s, a);
    }
    return PAM_E_ARG;
}
