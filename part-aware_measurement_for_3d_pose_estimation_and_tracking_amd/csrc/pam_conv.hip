// libpam_hip.so, conv part of a1: the HRNet-W48 convolution stack as hand-written MFMA kernels for gfx950.
//
// k_conv_igemm: NHWC bf16 convolution (1x1 / 3x3, stride 1 / 2) as an implicit GEMM
//     D[pixel][cout] = sum_k A[pixel][k] * Wt[cout][k],   k = (ky, kx, cin) flattened, cin fastest
// on v_mfma_f32_16x16x32_bf16 (wave64), with the whole epilogue fused: + bias (folded BatchNorm) [+ residual] [ReLU],
// fp32 accumulate, one bf16 rounding.  The A operand is gathered on the fly (no im2col buffer): each 16-byte piece is
// 8 consecutive input channels of one tap of one output pixel (Cin % 8 == 0), zero-filled outside the image.
// k_upsample_add: the HRNet fuse-layer sum  out = [ReLU](base + sum_t nearest_upsample(term_t)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvArgs {
    const uint16_t* in; const uint16_t* w; const float* bias; const uint16_t* res; uint16_t* out;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu, Ktot, Kpad, M;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

constexpr int KC = 64;               // K elements staged per LDS chunk (two 32-deep MFMA steps)
constexpr int ROWB = KC * 2 + 16;    // LDS row pitch in bytes: 128 B of data + 16 B pad (spreads ds_read_b128 over banks)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define OOB_OFFSET 0x80000000u          // beyond any buffer's num_records: the hardware bounds check returns zeros

// Block tile: BM = 64*WM output pixels x BN = 16*NTW*WN output channels; each wave owns 64 pixels x 16*NTW channels
// (4 x NTW accumulator tiles of 16x16).  K is walked in chunks of 64; chunk c+1 is fetched (buffer_load, zero-fill by the
// descriptor's range check, no branches) while chunk c is multiplied out of LDS.
template <int NTW, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_igemm(ConvArgs a) {
    constexpr int T = 64 * WM * WN, BM = 64 * WM, BN = 16 * NTW * WN;
    constexpr int APT = BM * 8 / T;                    // A pieces (16 B) per thread per chunk
    constexpr int BPT = (BN * 8 + T - 1) / T;          // B pieces per thread per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                   // [2][BM][ROWB]
    char* Bs = smem + 2 * BM * ROWB;                   // [2][BN][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kq = tid & 7;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Cout * a.Kpad * 2), 0x00020000);

    // per-thread output-pixel rows of the A tile (fixed over the K loop): byte offset of the window corner and a
    // validity bit per tap
    unsigned rowoff[APT], tapmask[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int row = (tid >> 3) + i * (T / 8);
        const int m = m0 + row;
        rowoff[i] = 0; tapmask[i] = 0;
        if (m < a.M) {
            const int hw = a.Ho * a.Wo;
            const int n = m / hw, r = m - n * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            rowoff[i] = (unsigned)((((long)n * a.H + iy0) * a.W + ix0) * a.Cin * 2);   // may wrap; only used with valid taps
            unsigned mk = 0;
            for (int ky = 0; ky < a.KH; ++ky)
                for (int kx = 0; kx < a.KW; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W) mk |= 1u << (ky * a.KW + kx);
            tapmask[i] = mk;
        }
    }
    int kc_c = kq * 8, kc_tap = 0;                      // channel / tap of this thread's piece in the current chunk
    while (kc_c >= a.Cin) { kc_c -= a.Cin; ++kc_tap; }
    unsigned woff[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        const int p = tid + i * T;
        woff[i] = (p < BN * 8) ? (unsigned)(((size_t)(n0 + (p >> 3)) * a.Kpad + (p & 7) * 8) * 2) : OOB_OFFSET;
    }

    u32x4 areg[APT], breg[BPT];
    auto load_chunk = [&]() {
        const int ky = kc_tap / a.KW, kx = kc_tap - ky * a.KW;
        const unsigned tapoff = (unsigned)(((ky * a.W + kx) * a.Cin + kc_c) * 2);
        const unsigned tbit = (kc_tap < a.KH * a.KW) ? (1u << kc_tap) : 0u;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const unsigned off = (tapmask[i] & tbit) ? rowoff[i] + tapoff : OOB_OFFSET;
            areg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[i], 0, 0);
            if (woff[i] != OOB_OFFSET) woff[i] += KC * 2;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int row = (tid >> 3) + i * (T / 8);
            *(u32x4*)(As + (size_t)buf * BM * ROWB + row * ROWB + kq * 16) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int p = tid + i * T;
            if (p < BN * 8) *(u32x4*)(Bs + (size_t)buf * BN * ROWB + (p >> 3) * ROWB + (p & 7) * 16) = breg[i];
        }
    };
    auto advance = [&]() {
        kc_c += KC;
        while (kc_c >= a.Cin) { kc_c -= a.Cin; ++kc_tap; }
    };

    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Kpad / KC;
    load_chunk();
    store_chunk(0);
    advance();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk();             // loads in flight under the MFMAs below
        const char* Ab = As + (size_t)buf * BM * ROWB + (wm * 64 + (lane & 15)) * ROWB + (lane >> 4) * 16;
        const char* Bb = Bs + (size_t)buf * BN * ROWB + (wn * 16 * NTW + (lane & 15)) * ROWB + (lane >> 4) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[NTW];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(Ab + i * 16 * ROWB + ks * 64);
#pragma unroll
            for (int j = 0; j < NTW; ++j) bfr[j] = *(const bf16x8*)(Bb + j * 16 * ROWB + ks * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                       __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) { store_chunk(buf ^ 1); advance(); }
        __syncthreads();
    }

    // epilogue.  D tile (16x16): column = lane & 15 (channel), rows = (lane >> 4) * 4 + r (pixels).  The wave's 64 x 16*NTW
    // tile goes through LDS (fp32, + bias) so that residual loads and output stores are 16 B per lane over whole pixels.
    constexpr int P = 16 * NTW + 4;                     // floats per staged pixel row (pad keeps 16-B alignment, breaks banks)
    float* Ew = (float*)smem + (size_t)wave * 64 * P;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const float b = a.bias ? a.bias[n0 + wn * 16 * NTW + j * 16 + (lane & 15)] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ew[(i * 16 + (lane >> 4) * 4 + r) * P + j * 16 + (lane & 15)] = acc[i][j][r] + b;
    }
    __syncthreads();
    constexpr int PPX = 2 * NTW;                        // 16-byte output pieces per pixel of this wave's channel span
    const int mw0 = m0 + wm * 64, cw0 = n0 + wn * 16 * NTW;
#pragma unroll
    for (int t = 0; t < PPX; ++t) {
        const int q = lane + 64 * t;
        const int px = q / PPX, c8 = q - px * PPX;
        const int m = mw0 + px;
        if (m < a.M) {
            const f32x4 v0 = *(const f32x4*)(Ew + px * P + c8 * 8);
            const f32x4 v1 = *(const f32x4*)(Ew + px * P + c8 * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const size_t o = (size_t)m * a.Cout + cw0 + c8 * 8;
            if (a.res) {
                const bf16x8 rr = *(const bf16x8*)(a.res + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += bf16_to_f32((uint16_t)rr[k]);
            }
            bf16x8 ov;
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[k] = (short)f32_to_bf16_rne(a.relu ? fmaxf(v[k], 0.0f) : v[k]);
            *(bf16x8*)(a.out + o) = ov;
        }
    }
}

template <int NTW, int WM, int WN>
static int launch_conv(hipStream_t s, const ConvArgs& a) {
    constexpr int BM = 64 * WM, BN = 16 * NTW * WN;
    dim3 grid((a.M + BM - 1) / BM, a.Cout / BN);
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    hipLaunchKernelGGL((k_conv_igemm<NTW, WM, WN>), grid, dim3(64 * WM * WN), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// tile choice: widest block tile that still gives the chip >= ~2 workgroups per CU, else the smallest
template <int NTW>
static int dispatch_conv(hipStream_t s, const ConvArgs& a, int force) {
    const int nb = a.Cout / (16 * NTW);                 // N tiles of one wave
    auto blocks = [&](int wm, int wn) { return (long)((a.M + 64 * wm - 1) / (64 * wm)) * (nb / wn); };
    int cfg = force;
    if (cfg < 0) {
        if (nb % 2 == 0 && blocks(2, 2) >= 512) cfg = 3;
        else if (blocks(4, 1) >= 512) cfg = 4;
        else if (blocks(2, 1) >= 512) cfg = 2;
        else if (nb % 2 == 0 && blocks(1, 2) >= 384) cfg = 1;
        else cfg = 0;
    }
    switch (cfg) {
        case 0: return launch_conv<NTW, 1, 1>(s, a);
        case 1: return (nb % 2) ? PAM_E_ARG : launch_conv<NTW, 1, 2>(s, a);
        case 2: return launch_conv<NTW, 2, 1>(s, a);
        case 3: return (nb % 2) ? PAM_E_ARG : launch_conv<NTW, 2, 2>(s, a);
        case 4: return launch_conv<NTW, 4, 1>(s, a);
    }
    return PAM_E_ARG;
}

// ====================================================================================================================
// k_conv3x3: 3x3 / stride 1 / pad 1 convolutions (85 % of HRNet-W48's FLOPs) with the INPUT PATCH resident in LDS.
// A workgroup owns a TH x TW tile of one image and one slab of 16*NTW output channels.  The (TH+2) x (TW+2) x CC input
// patch (CC = channel chunk <= 96) is fetched ONCE (not once per tap), every MFMA A fragment is a ds_read_b128 at
// [pixel + tap][channel slice], and the B (weight) fragments stream straight from L2 into registers one K-step ahead
// (no weight staging, no barrier inside the K loop).
// ====================================================================================================================
struct C3Args {
    const uint16_t* in; const uint16_t* w; const float* bias; const uint16_t* res; uint16_t* out;
    int N, H, W, Cout, Kpad, TH, TW, tiles_x, tiles_y, relu, dbg;
};

__host__ __device__ constexpr int c3_maxpatch(int nwaves) { return nwaves == 1 ? 112 : nwaves == 2 ? 192 : nwaves == 3 ? 272 : 352; }

template <int CIN, int NTW, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void k_conv3x3(C3Args a) {
    constexpr int T = 64 * NWAVES, BN = 16 * NTW;
    constexpr int CK = (CIN == 48) ? 48 : 32;           // input channels per resident chunk
    constexpr int NCHUNK = CIN / CK, PC8 = CK / 8;
    constexpr int PITCH_A = CK * 2 + 16;                // bytes per patch pixel   (pad: conflict-free ds_read_b128)
    constexpr int PITCH_W = 9 * CK * 2 + 16;            // bytes per weight row    (9 taps x CK channels of one cout)
    constexpr int NPP = (c3_maxpatch(NWAVES) * PC8 + T - 1) / T;    // patch pieces per thread per chunk
    constexpr int NWQ = BN * 9 * PC8;                                // weight pieces per chunk
    constexpr int NWP = (NWQ + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int bx = blockIdx.x;
    const int tx = bx % a.tiles_x, ty = (bx / a.tiles_x) % a.tiles_y, n = bx / (a.tiles_x * a.tiles_y);
    const int ty0 = ty * a.TH, tx0 = tx * a.TW;
    const int PW = a.TW + 2, PH = a.TH + 2, npatch = PW * PH, npx = a.TH * a.TW;
    const int n0 = blockIdx.y * BN;
    char* Wsm = smem + (((size_t)npatch * PITCH_A + 15) & ~(size_t)15);
    char* zero_slot = Wsm + (size_t)BN * PITCH_W;       // 48 zero bytes: K-tail A lanes + slack behind the last weight row
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * CIN * 2), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Cout * a.Kpad * 2), 0x00020000);
    if (tid < 3) *(u32x4*)(zero_slot + tid * 16) = (u32x4){0, 0, 0, 0};
    if (CIN == 48 && tid < BN) *(u32x4*)(Wsm + (size_t)tid * PITCH_W + 9 * CK * 2) = (u32x4){0, 0, 0, 0};   // row pads feed the K tail

    // ---- per-thread piece descriptors (fixed over the chunk loop) ------------------------------------------------------
    unsigned goffA[NPP], goffW[NWP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int q = tid + i * T;
        goffA[i] = OOB_OFFSET;
        if (q < npatch * PC8) {
            const int pp = q / PC8, c8 = q - pp * PC8;
            const int pyy = pp / PW, pxx = pp - pyy * PW;
            const int iy = ty0 - 1 + pyy, ix = tx0 - 1 + pxx;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                goffA[i] = (unsigned)((((size_t)n * a.H + iy) * a.W + ix) * CIN * 2 + c8 * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int w = tid + i * T;
        goffW[i] = OOB_OFFSET;
        if (w < NWQ) {
            const int co = w / (9 * PC8), r = w - co * (9 * PC8), t = r / PC8, c8 = r - t * PC8;
            goffW[i] = (unsigned)(((size_t)(n0 + co) * a.Kpad + t * CIN + c8 * 8) * 2);
        }
    }
    // LDS byte offset of each of this lane's 4 output-pixel slots (window corner in the haloed patch)
    unsigned lanebase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = wave * 64 + i * 16 + (lane & 15);
        if (p >= npx) p = 0;
        const int py = p / a.TW, px = p - py * a.TW;
        lanebase[i] = (unsigned)((py * PW + px) * PITCH_A);
    }

    u32x4 ra[NPP], rw[NWP];
    auto gload = [&](int cc) {
        const unsigned so = (unsigned)(cc * CK * 2);
#pragma unroll
        for (int i = 0; i < NPP; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, goffA[i], goffA[i] == OOB_OFFSET ? 0 : so, 0);
#pragma unroll
        for (int i = 0; i < NWP; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, goffW[i], goffW[i] == OOB_OFFSET ? 0 : so, 0);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = tid + i * T;
            if (q < npatch * PC8) { const int pp = q / PC8, c8 = q - pp * PC8; *(u32x4*)(smem + (size_t)pp * PITCH_A + c8 * 16) = ra[i]; }
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int w = tid + i * T;
            if (w < NWQ) {
                const int co = w / (9 * PC8), r = w - co * (9 * PC8), t = r / PC8, c8 = r - t * PC8;
                *(u32x4*)(Wsm + (size_t)co * PITCH_W + (t * CK + c8 * 8) * 2) = rw[i];
            }
        }
    };

    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const char* wl = Wsm + (size_t)(lane & 15) * PITCH_W;               // this lane's weight row inside each 16-row N tile
    if (!(a.dbg & 1)) gload(0);
    for (int cc = 0; cc < NCHUNK; ++cc) {
        if (cc > 0) __syncthreads();                    // every wave is done reading the previous chunk
        lstore();
        __syncthreads();
        if (cc + 1 < NCHUNK && !(a.dbg & 1)) gload(cc + 1);             // next chunk in flight under the MFMAs below
        if (a.dbg & 2) continue;
        if constexpr (CIN == 48) {
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                const int k0 = 32 * s + 8 * g;                           // flattened (tap, c); an 8-slice never straddles taps
                const int t = k0 / 48, c = k0 - t * 48;
                const int ky = t / 3, kx = t - ky * 3;
                const bool zero = k0 >= 432;
                const unsigned aoff = (unsigned)((ky * PW + kx) * PITCH_A + c * 2);
                bf16x8 af[4], bfr[NTW];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(zero ? zero_slot : smem + lanebase[i] + aoff);
#pragma unroll
                for (int j = 0; j < NTW; ++j) bfr[j] = *(const bf16x8*)(wl + (size_t)j * 16 * PITCH_W + k0 * 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                           __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const unsigned aoff = (unsigned)((ky * PW + kx) * PITCH_A + g * 16);
                bf16x8 af[4], bfr[NTW];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(smem + lanebase[i] + aoff);
#pragma unroll
                for (int j = 0; j < NTW; ++j) bfr[j] = *(const bf16x8*)(wl + (size_t)j * 16 * PITCH_W + t * 64 + g * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                           __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- epilogue through LDS (fp32 + bias), then 16-byte residual loads / output stores over whole pixels ----------
    constexpr int P = 16 * NTW + 4;
    float* Ew = (float*)smem + (size_t)wave * 64 * P;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const float b = a.bias ? a.bias[n0 + j * 16 + (lane & 15)] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ew[(i * 16 + g * 4 + r) * P + j * 16 + (lane & 15)] = acc[i][j][r] + b;
    }
    __syncthreads();
    constexpr int PPX = 2 * NTW;
    if (a.dbg & 4) { if (tid == 0) a.out[(size_t)blockIdx.x * 8] = (uint16_t)Ew[0]; return; }
#pragma unroll
    for (int t = 0; t < PPX; ++t) {
        const int q = lane + 64 * t;
        const int slot = q / PPX, c8 = q - slot * PPX;
        const int p = wave * 64 + slot;
        if (p < npx) {
            const int py = p / a.TW, px = p - py * a.TW;
            const size_t m = ((size_t)n * a.H + ty0 + py) * a.W + tx0 + px;
            const f32x4 v0 = *(const f32x4*)(Ew + slot * P + c8 * 8);
            const f32x4 v1 = *(const f32x4*)(Ew + slot * P + c8 * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const size_t o = m * a.Cout + n0 + c8 * 8;
            if (a.res) {
                const bf16x8 rr = *(const bf16x8*)(a.res + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += bf16_to_f32((uint16_t)rr[k]);
            }
            bf16x8 ov;
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[k] = (short)f32_to_bf16_rne(a.relu ? fmaxf(v[k], 0.0f) : v[k]);
            *(bf16x8*)(a.out + o) = ov;
        }
    }
}

template <int CIN, int NTW>
static int launch_c3(hipStream_t s, const C3Args& a, int nwaves, size_t lds) {
    dim3 grid(a.tiles_x * a.tiles_y * a.N, a.Cout / (16 * NTW));
    switch (nwaves) {
        case 2: hipLaunchKernelGGL((k_conv3x3<CIN, NTW, 2>), grid, dim3(128), lds, s, a); break;
        case 3: hipLaunchKernelGGL((k_conv3x3<CIN, NTW, 3>), grid, dim3(192), lds, s, a); break;
        case 4: hipLaunchKernelGGL((k_conv3x3<CIN, NTW, 4>), grid, dim3(256), lds, s, a); break;
        default: return PAM_E_ARG;
    }
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// choose the spatial tile: divisors of H, W; >= 2 waves (the block's waves share the staged weights), prefer full MFMA
// rows, enough workgroups to fill 256 CUs, big tiles (fewer weight re-fetches) and small halo
static bool pick_tile(int N, int H, int W, int Cin, int Cout, int ntw, int& TH, int& TW, int& nwaves, size_t& lds) {
    const int CK = Cin == 48 ? 48 : 32;
    const int pitch_a = CK * 2 + 16, pitch_w = 9 * CK * 2 + 16, bn = 16 * ntw;
    double best = -1;
    for (int th = 1; th <= H; ++th) {
        if (H % th) continue;
        for (int tw = 1; tw <= W; ++tw) {
            if (W % tw) continue;
            const int npx = th * tw;
            if (npx > 256 || npx <= 64) continue;
            const int nw = (npx + 63) / 64;
            const int npatch = (th + 2) * (tw + 2);
            if (npatch > c3_maxpatch(nw)) continue;
            const size_t main_b = (((size_t)npatch * pitch_a + 15) & ~(size_t)15) + (size_t)bn * pitch_w + 64;
            const size_t epi = (size_t)nw * 64 * (16 * ntw + 4) * 4;
            const size_t need = main_b > epi ? main_b : epi;
            if (need > 80 * 1024) continue;
            const double util = (double)npx / (64.0 * nw);
            const double blocks = (double)(H / th) * (W / tw) * N * (Cout / bn);
            const double fill = blocks >= 512 ? 1.0 : blocks / 512.0;
            const double halo = (double)npx / npatch;
            const double share = nw >= 3 ? 1.0 : 0.85;
            const double score = util * (0.3 + 0.7 * fill) * (0.6 + 0.4 * halo) * share;
            if (score > best) { best = score; TH = th; TW = tw; nwaves = nw; lds = need; }
        }
    }
    return best > 0;
}

extern "C" int pam_conv2d_nhwc_bf16(void* stream, const void* in, const void* w_packed, const float* bias,
                                    const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad, int relu, int tile_cfg) {
    if (!in || !w_packed || !out || N <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || (Cout % 48 != 0 && Cout % 64 != 0) ||
        KH < 1 || KW < 1 || KH > 3 || KW > 3 || stride < 1)
        return PAM_E_ARG;
    ConvArgs a;
    a.in = (const uint16_t*)in; a.w = (const uint16_t*)w_packed; a.bias = bias; a.res = (const uint16_t*)residual;
    a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.Ktot = KH * KW * Cin; a.Kpad = (a.Ktot + KC - 1) / KC * KC; a.M = N * a.Ho * a.Wo;
    if (H >= 32768 || W >= 32768) return PAM_E_ARG;
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && (tile_cfg < 0 || tile_cfg >= 100) &&
        (Cin == 48 || Cin == 64 || Cin == 96 || Cin == 192 || Cin == 384)) {
        const int ntw = (Cout % 48 == 0) ? 3 : 4;
        C3Args c;
        c.in = a.in; c.w = a.w; c.bias = bias; c.res = a.res; c.out = a.out;
        c.N = N; c.H = H; c.W = W; c.Cout = Cout; c.Kpad = a.Kpad; c.relu = relu; c.dbg = tile_cfg >= 100 ? tile_cfg - 100 : 0;
        int nw = 0; size_t lds = 0;
        if (pick_tile(N, H, W, Cin, Cout, ntw, c.TH, c.TW, nw, lds)) {
            c.tiles_x = W / c.TW; c.tiles_y = H / c.TH;
            hipStream_t s = (hipStream_t)stream;
            if (ntw == 3) {
                switch (Cin) {
                    case 48: return launch_c3<48, 3>(s, c, nw, lds);
                    case 64: return launch_c3<64, 3>(s, c, nw, lds);
                    case 96: return launch_c3<96, 3>(s, c, nw, lds);
                    case 192: return launch_c3<192, 3>(s, c, nw, lds);
                    case 384: return launch_c3<384, 3>(s, c, nw, lds);
                }
            } else {
                switch (Cin) {
                    case 48: return launch_c3<48, 4>(s, c, nw, lds);
                    case 64: return launch_c3<64, 4>(s, c, nw, lds);
                    case 96: return launch_c3<96, 4>(s, c, nw, lds);
                    case 192: return launch_c3<192, 4>(s, c, nw, lds);
                    case 384: return launch_c3<384, 4>(s, c, nw, lds);
                }
            }
        }
    }
    return (Cout % 48 == 0) ? dispatch_conv<3>((hipStream_t)stream, a, tile_cfg) : dispatch_conv<4>((hipStream_t)stream, a, tile_cfg);
}

// out[n,y,x,c] = [relu](base[n,y,x,c] + sum_t term_t[n, y >> sh_t, x >> sh_t, c]); 8 channels (16 B) per thread
struct UpArgs { const uint16_t* base; const uint16_t* term[3]; int sh[3]; int nterms; uint16_t* out; int N, H, W, C, relu; };
__global__ __launch_bounds__(256) void k_upsample_add(UpArgs a) {
    const size_t total = (size_t)a.N * a.H * a.W * (a.C / 8);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % (a.C / 8));
        const size_t pix = e / (a.C / 8);
        const int x = (int)(pix % a.W);
        const size_t t2 = pix / a.W;
        const int y = (int)(t2 % a.H), n = (int)(t2 / a.H);
        bf16x8 b = *(const bf16x8*)(a.base + pix * a.C + c8 * 8);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32((uint16_t)b[k]);
        for (int t = 0; t < a.nterms; ++t) {
            const int hs = a.H >> a.sh[t], ws = a.W >> a.sh[t];
            const bf16x8 q = *(const bf16x8*)(a.term[t] + (((size_t)n * hs + (y >> a.sh[t])) * ws + (x >> a.sh[t])) * a.C + c8 * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bf16_to_f32((uint16_t)q[k]);
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (short)f32_to_bf16_rne(a.relu ? fmaxf(v[k], 0.0f) : v[k]);
        *(bf16x8*)(a.out + pix * a.C + c8 * 8) = o;
    }
}

extern "C" int pam_upsample_add_nhwc_bf16(void* stream, const void* base, int n_terms, const void* const* terms,
                                          const int32_t* shifts, void* out, int N, int H, int W, int C, int relu) {
    if (!base || !out || n_terms < 0 || n_terms > 3 || C % 8 != 0) return PAM_E_ARG;
    UpArgs a;
    a.base = (const uint16_t*)base; a.out = (uint16_t*)out; a.nterms = n_terms;
    for (int t = 0; t < 3; ++t) { a.term[t] = t < n_terms ? (const uint16_t*)terms[t] : nullptr; a.sh[t] = t < n_terms ? shifts[t] : 0; }
    a.N = N; a.H = H; a.W = W; a.C = C; a.relu = relu;
    const size_t total = (size_t)N * H * W * (C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_upsample_add, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
