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
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/pam.h"
#include "pam_launch.hpp"

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// fused epilogue activation.  act & 3: 0 linear, 1 ReLU, 2 leaky ReLU (slope 0.1); act & 4: the residual is added AFTER the
// activation (Darknet shortcut layers) instead of before it (ResNet / HRNet blocks)
__device__ __forceinline__ float epi_act1(float v, int kind) {
    return kind == 1 ? fmaxf(v, 0.0f) : (kind == 2 ? (v > 0.0f ? v : 0.1f * v) : v);
}
__device__ __forceinline__ float epi_act(float v, float r, int act) {
    return (act & 4) ? epi_act1(v, act & 3) + r : epi_act1(v + r, act & 3);
}

struct ConvArgs {
    const uint16_t* in; const uint16_t* w; const float* bias; const uint16_t* res; uint16_t* out;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu, Ktot, Kpad, M;
    int in_cs;        // channel stride of the input pixels (= Cin unless the input is a channel slice of a wider tensor)
    int relu_from;    // the activation applies to output channels >= relu_from (0 = all); multiple of 16
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {     // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE) on gfx950
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {     // two floats -> one dword of two bf16 (one instruction)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

constexpr int KC = 64;               // K elements staged per LDS chunk (two 32-deep MFMA steps)
constexpr int ROWB = KC * 2 + 16;    // LDS row pitch in bytes: 128 B of data + 16 B pad (spreads ds_read_b128 over banks)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#define OOB_OFFSET 0x80000000u          // beyond any buffer's num_records: the hardware bounds check returns zeros

// One lane's row piece of 4*NTW contiguous bf16 channels (8*NTW bytes at byte offset o, 8-byte aligned; 16-byte aligned
// when NTW is even or the lane group g is even) as 16-byte accesses where possible.  NTW = 3 (24 bytes) splits 16 + 8 for
// even g and 8 + 16 for odd g, so the 16-byte half is always aligned.
template <int NTW>
__device__ __forceinline__ void c3_row_load(__amdgpu_buffer_rsrc_t rs, unsigned o, int g, uint32_t* d) {
    if constexpr (NTW == 1) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0); d[0] = v[0]; d[1] = v[1];
    } else if constexpr (NTW == 2) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    } else if constexpr (NTW == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0), w = __builtin_amdgcn_raw_buffer_load_b128(rs, o + 16, 0, 0);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; d[4] = w[0]; d[5] = w[1]; d[6] = w[2]; d[7] = w[3];
    } else if constexpr (NTW == 6) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o + 16 * k, 0, 0);
            d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
        }
    } else {
        static_assert(NTW == 3, "slab width");
        const bool odd = g & 1;
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, o + (odd ? 8u : 0u), 0, 0);
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, o + (odd ? 0u : 16u), 0, 0);
        d[0] = odd ? h[0] : q[0]; d[1] = odd ? h[1] : q[1]; d[2] = odd ? q[0] : q[2];
        d[3] = odd ? q[1] : q[3]; d[4] = odd ? q[2] : h[0]; d[5] = odd ? q[3] : h[1];
    }
}
template <int NTW>
__device__ __forceinline__ void c3_row_store(uint16_t* p, int g, const uint32_t* d) {
    if constexpr (NTW == 1) {
        *(u32x2*)p = (u32x2){d[0], d[1]};
    } else if constexpr (NTW == 2) {
        *(u32x4*)p = (u32x4){d[0], d[1], d[2], d[3]};
    } else if constexpr (NTW == 4) {
        *(u32x4*)p = (u32x4){d[0], d[1], d[2], d[3]}; *(u32x4*)(p + 8) = (u32x4){d[4], d[5], d[6], d[7]};
    } else if constexpr (NTW == 6) {
#pragma unroll
        for (int k = 0; k < 3; ++k) *(u32x4*)(p + 8 * k) = (u32x4){d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]};
    } else {
        const bool odd = g & 1;
        *(u32x4*)(p + (odd ? 4 : 0)) = odd ? (u32x4){d[2], d[3], d[4], d[5]} : (u32x4){d[0], d[1], d[2], d[3]};
        *(u32x2*)(p + (odd ? 0 : 8)) = odd ? (u32x2){d[0], d[1]} : (u32x2){d[4], d[5]};
    }
}


// Block tile: BM = 64*WM output pixels x BN = 16*NTW*WN output channels; each wave owns 64 pixels x 16*NTW channels
// (4 x NTW accumulator tiles of 16x16).  K is walked in chunks of 64; chunk c+1 is fetched (buffer_load, zero-fill by the
// descriptor's range check, no branches) while chunk c is multiplied out of LDS.
template <int NTW, int WM, int WN, bool GEN>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int bx, const int by) {
    constexpr int T = 64 * WM * WN, BM = 64 * WM, BN = 16 * NTW * WN;
    constexpr int APT = (BM * 8 + T - 1) / T;          // A pieces (16 B) per thread per chunk (the last pass is partial when T does not divide BM * 8)
    constexpr int BPT = (BN * 8 + T - 1) / T;          // B pieces per thread per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two chunk buffers each -- or one when the whole K is a single chunk (1x1 layers with 64 input channels: half the LDS, twice
    // the workgroups per CU for layers that are one load -> multiply -> store chain per workgroup)
    const int nbuf = a.Kpad > KC ? 2 : 1;
    char* As = smem;                                   // [nbuf][BM][ROWB]
    char* Bs = smem + nbuf * BM * ROWB;                // [nbuf][BN][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = bx * BM, n0 = by * BN;
    const int kq = tid & 7;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * a.in_cs * 2), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Cout * a.Kpad * 2), 0x00020000);

    // per-thread output-pixel rows of the A tile (fixed over the K loop): byte offset of the window corner and a
    // validity bit per tap
    unsigned rowoff[APT], tapmask[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int row = (tid >> 3) + i * (T / 8);
        const int m = m0 + row;
        rowoff[i] = 0; tapmask[i] = 0;
        if (m < a.M && row < BM) {
            const int hw = a.Ho * a.Wo;
            const int n = m / hw, r = m - n * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            rowoff[i] = (unsigned)((((long)n * a.H + iy0) * a.W + ix0) * a.in_cs * 2); // may wrap; only used with valid taps
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
        // LDS weight row wn*16*NTW + j*16 + q holds output channel wn*16*NTW + 4*NTW*(q >> 2) + 4*j + (q & 3): as the MFMA A operand
        // this leaves every lane with 4*NTW contiguous channels of its pixel (same scheme as k_conv3x3)
        const int row = p >> 3, wnb = row / (16 * NTW), rem = row - wnb * 16 * NTW, q = rem & 15;
        const int ch = wnb * 16 * NTW + 4 * NTW * (q >> 2) + 4 * (rem >> 4) + (q & 3);
        woff[i] = (p < BN * 8) ? (unsigned)(((size_t)(n0 + ch) * a.Kpad + (p & 7) * 8) * 2) : OOB_OFFSET;
    }

    u32x4 areg[APT], breg[BPT];
    auto load_chunk = [&]() {
        const int ky = kc_tap / a.KW, kx = kc_tap - ky * a.KW;
        const unsigned tapoff = (unsigned)(((ky * a.W + kx) * a.in_cs + kc_c) * 2);
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
            if (BM * 8 % T == 0 || row < BM) *(u32x4*)(As + (size_t)buf * BM * ROWB + row * ROWB + kq * 16) = areg[i];
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

    const int g = lane >> 4, cw0 = n0 + wn * 16 * NTW;
    f32x4 acc[4][NTW];                                  // [pixel tile][channel tile], started from the bias
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const f32x4 b4 = a.bias ? *(const f32x4*)(a.bias + cw0 + g * 4 * NTW + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = b4;
    }

    // residual rows of this lane's 4 pixel tiles: issued BEFORE the K loop, so their latency runs under the operand loads and the
    // MFMAs instead of in front of the stores (the 1x1 64 -> 256 + residual layers of layer1 are one K chunk long: nothing else hides it)
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, (int)((size_t)a.M * a.Cout * 2), 0x00020000);
    uint32_t rres[4][2 * NTW];
    if (a.res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            const size_t o = (size_t)m * a.Cout + cw0 + g * 4 * NTW;
            c3_row_load<NTW>(rs_res, m < a.M ? (unsigned)(o * 2) : OOB_OFFSET, g, rres[i]);
        }
    }

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
        // both 32-deep steps of the chunk: all fragment reads first, pinned ahead of the MFMAs (hipcc otherwise sinks each
        // ds_read to just before its first use and the LDS latency is exposed every three MFMAs)
        bf16x8 af[2][4], bfr[2][NTW];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *(const bf16x8*)(Ab + i * 16 * ROWB + ks * 64);
#pragma unroll
            for (int j = 0; j < NTW; ++j) bfr[ks][j] = *(const bf16x8*)(Bb + j * 16 * ROWB + ks * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[ks][j]),
                                                                       __builtin_bit_cast(bf16x8_t, af[ks][i]), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunks) { store_chunk(buf ^ 1); advance(); }
        __syncthreads();
    }

    // epilogue straight from the accumulators: with the weights as the A operand the D tile has channels on its rows, so this
    // lane holds channels cw0 + 4*NTW*g + 4*j + r of pixel mw0 + i*16 + (lane & 15): 4*NTW contiguous channels, 16-byte accesses
    const int mw0 = m0 + wm * 64;
    typedef __attribute__((ext_vector_type(2))) short s16x2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mw0 + i * 16 + (lane & 15);
        if (m < a.M) {
            const size_t o = (size_t)m * a.Cout + cw0 + g * 4 * NTW;
            uint32_t ov[2 * NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                float rr[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.res) {
                    rr[0] = __builtin_bit_cast(float, rres[i][2 * j] << 16); rr[1] = __builtin_bit_cast(float, rres[i][2 * j] & 0xffff0000u);
                    rr[2] = __builtin_bit_cast(float, rres[i][2 * j + 1] << 16); rr[3] = __builtin_bit_cast(float, rres[i][2 * j + 1] & 0xffff0000u);
                }
                const bool act_on = cw0 + g * 4 * NTW + j * 4 >= a.relu_from;   // merged fuse-layer convs: only the upper channels
                if constexpr (GEN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_act(v[r], rr[r], act_on ? a.relu : (a.relu & 4));
                    ov[2 * j] = pack_bf16x2(v[0], v[1]); ov[2 * j + 1] = pack_bf16x2(v[2], v[3]);
                } else {                                // HRNet's codes 0 / 1; ReLU as a packed int16 max on the bf16 pairs
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    ov[2 * j] = pack_bf16x2(v[0], v[1]); ov[2 * j + 1] = pack_bf16x2(v[2], v[3]);
                    if (a.relu && act_on) {
                        ov[2 * j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j]), (s16x2){0, 0}));
                        ov[2 * j + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j + 1]), (s16x2){0, 0}));
                    }
                }
            }
            c3_row_store<NTW>(a.out + o, g, ov);
        }
    }
}

template <int NTW, int WM, int WN, bool GEN>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_igemm(ConvArgs a) {
    conv_igemm_body<NTW, WM, WN, GEN>(a, blockIdx.x, blockIdx.y);
}

template <int NTW, int WM, int WN>
static int launch_conv(hipStream_t s, const ConvArgs& a) {
    constexpr int BM_ = 64 * WM, BN_ = 16 * NTW * WN;
    if (a.relu > 1) {                                   // Darknet activation codes: the general-epilogue instantiation
        dim3 grid((a.M + BM_ - 1) / BM_, a.Cout / BN_);
        const size_t lds = (a.Kpad > KC ? 2 : 1) * (size_t)(BM_ + BN_) * ROWB;
        pam_launch(k_conv_igemm<NTW, WM, WN, true>, grid, dim3(64 * WM * WN), lds, s, a);
        return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
    }
    constexpr int BM = 64 * WM, BN = 16 * NTW * WN;
    dim3 grid((a.M + BM - 1) / BM, a.Cout / BN);
    const size_t lds = (a.Kpad > KC ? 2 : 1) * (size_t)(BM + BN) * ROWB;
    pam_launch(k_conv_igemm<NTW, WM, WN, false>, grid, dim3(64 * WM * WN), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// tile choice: share the staged input tile between two N tiles when there are two; otherwise 128-pixel blocks while they still
// give the chip >= 2 workgroups per CU, else 64-pixel blocks
template <int NTW>
static int dispatch_conv(hipStream_t s, const ConvArgs& a, int force) {
    const int nb = a.Cout / (16 * NTW);                 // N tiles of one wave
    auto blocks = [&](int wm, int wn) { return (long)((a.M + 64 * wm - 1) / (64 * wm)) * (nb / wn); };
    int cfg = force;
    if (cfg < 0) {                                      // measured per HRNet layer shape with tools/bench_conv.py --tiles=...
        // 48-wide N tiles, 3 or 4 of them: the whole layer per pixel tile, the gathered input staged ONCE (merged fuse-layer heads
        // 48 -> 144 / 192, stride 2: 33.6 -> 25.4 us, 28.5 -> 25.9 us at 20 crops; 6 tiles: no gain over two workgroups of 3)
        if (NTW == 3 && nb == 3) cfg = 5;
        else if (NTW == 3 && nb == 4) cfg = 6;
        else if (nb % 2 == 0) cfg = blocks(2, 2) >= 400 ? 3 : 1;   // two N tiles per workgroup: the input tile is staged once for both
        else cfg = blocks(2, 1) >= 512 ? 2 : 0;
    }
    switch (cfg) {
        case 0: return launch_conv<NTW, 1, 1>(s, a);
        case 1: return (nb % 2) ? PAM_E_ARG : launch_conv<NTW, 1, 2>(s, a);
        case 2: return launch_conv<NTW, 2, 1>(s, a);
        case 3: return (nb % 2) ? PAM_E_ARG : launch_conv<NTW, 2, 2>(s, a);
        case 4: return launch_conv<NTW, 4, 1>(s, a);
        case 5: return (nb % 3) ? PAM_E_ARG : launch_conv<NTW, 1, 3>(s, a);     // all of a 144- / 192-channel layer per pixel tile: input staged once
        case 6: return (nb % 4) ? PAM_E_ARG : launch_conv<NTW, 1, 4>(s, a);
        case 7: return (nb % 3) ? PAM_E_ARG : launch_conv<NTW, 2, 3>(s, a);
    }
    return PAM_E_ARG;
}

// ====================================================================================================================
// k_conv3x3: 3x3 / stride 1 / pad 1 convolutions (85 % of HRNet-W48's FLOPs) with the input rows resident in LDS.
//
// A workgroup owns TH full image rows of one image and one slab of BN = 16*NTW output channels.  Output "slots" are the
// positions of the PADDED row-major grid (PW = W + 2 columns): slot p <-> window corner at patch pixel p, so the MFMA A
// fragment of tap (ky,kx) is one ds_read_b128 at (p + ky*PW + kx) * PITCH_A -- linear in p, conflict-free at
// PITCH_A = 96 B -- and the two junk columns per row are simply not stored.  K is walked in chunks of CK input channels:
// the (TH+2) x PW x CK patch chunk and the [BN][9][CK] weight chunk (pre-packed on the host as an LDS image, so its load
// is a linear 16-B copy) go global -> registers -> LDS, the next chunk in flight under the current chunk's MFMAs.
// ====================================================================================================================
struct C3Args {
    const uint16_t* in; const uint16_t* wimg; const float* bias; const uint16_t* res; uint16_t* out;
    int N, H, W, Cout, TH, tiles_y, relu;
    float inv_pw;
#ifdef PAM_DIAG
    int dbg;                         // phase knock-outs / stamps for tools/stamp_conv.py
    unsigned long long* stamps;      // (dbg & 64): per-workgroup s_memtime stamps, never read by the kernel
#endif
};
// Diagnostic build only (make DIAG=1): phase knock-outs, in-kernel stamps and environment tuning overrides.  The shipped library is
// compiled without them -- C3_DBG folds to false, C3_STAMP to nothing, no getenv on the launch path.
#ifdef PAM_DIAG
static unsigned long long* g_c3_stamps = nullptr;
extern "C" int pam_conv_debug_stamps(void* dev_buf) { g_c3_stamps = (unsigned long long*)dev_buf; return PAM_OK; }
#define C3_DBG(bit) (a.dbg & (bit))
#else
extern "C" int pam_conv_debug_stamps(void*) { return PAM_E_ARG; }     // kept in the ABI; only the diagnostic build records stamps
#define C3_DBG(bit) false
#endif
// chunk of input channels resident in LDS per K pass: all 48 for Cin = 48 (K walked as the flattened (tap, c) index), 64 for the
// deep small-image layers (fewer, longer passes hide the load latency), 32 otherwise.  Pitches from tools/lds_sim.py:
// conflict-free ds_read_b128 needs pitch = 32 (mod 64) bytes for the pixel rows and these row pitches for the weights.
// widths whose instantiation carries the general (Darknet) activation epilogue; the others take codes 0 / 1 only
__host__ __device__ constexpr bool c3_general_act(int cin) { return cin == 64 || cin == 128 || cin == 256 || cin == 512; }
__host__ __device__ constexpr int c3_ck(int cin) { return cin == 48 ? 48 : (cin >= 192 ? 64 : 32); }
__host__ __device__ constexpr int c3_pitch_a(int cin) { return c3_ck(cin) == 64 ? 160 : 96; }
__host__ __device__ constexpr int c3_pitch_w(int cin) { return cin == 48 ? 864 : (c3_ck(cin) == 64 ? 1184 : 608); }

__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }   // exact for x < 2^16

// the deep small-image layers run one workgroup per CU (one wave per SIMD): give those instantiations the whole register
// file, otherwise the scheduler, starved by the chunk-prefetch registers, reads each MFMA fragment right before its use
template <int CIN, int NTW, int MT, int NWAVES, int PMAX>
__global__ __launch_bounds__(64 * NWAVES, (c3_ck(CIN) == 64 ? 1 : 2)) void k_conv3x3(C3Args a) {
    constexpr int T = 64 * NWAVES, BN = 16 * NTW;
    constexpr int CK = c3_ck(CIN), NCHUNK = CIN / CK, PC8 = CK / 8;
    constexpr int PITCH_A = c3_pitch_a(CIN), PITCH_W = c3_pitch_w(CIN);
    constexpr int NPP = (PMAX * PC8 + T - 1) / T;                        // patch pieces per thread per chunk (PMAX >= patch pixels)
    constexpr int WIMG = BN * PITCH_W;                                    // bytes of one weight chunk image
    constexpr int NWP = (WIMG / 16 + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (C3_DBG(8)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2), so give each XCD a contiguous run of
    // tiles -- vertically adjacent tiles re-read each other's halo rows, which then hit that L2 instead of the fabric
    // Single-chunk layers (Cin = 48) are PERSISTENT: the grid is capped at the resident workgroups and each one walks tiles
    // v = blockIdx.x, + gridDim.x, ... with the slab's weights staged once and the next tile's patch in flight (registers)
    // under the current tile's epilogue.  gridDim.x is a multiple of 8 then, so a workgroup stays on its XCD's run.
    const int ntiles = a.tiles_y * a.N;
    auto tile_of = [&](int v) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, loc = v >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    };
    int vtile = blockIdx.x;
    int bx = tile_of(vtile);
    int n = bx / a.tiles_y, ty0 = (bx - n * a.tiles_y) * a.TH;
    const int PW = a.W + 2, npatch = (a.TH + 2) * PW;   // full-height patch: rows below a ragged last tile load as zeros
    int nslots = min(a.TH, a.H - ty0) * PW;
    const int n0 = blockIdx.y * BN;
    char* Wsm = smem + ((((size_t)npatch + 2) * PITCH_A + 15) & ~(size_t)15);   // junk-slot reads past the patch land in the weights (in bounds)
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * CIN * 2), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, (int)((size_t)(a.Cout / BN) * NCHUNK * WIMG), 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
    uint32_t rres[MT][2 * NTW];                         // residual row piece of this lane: 4*NTW contiguous channels
    char* zero_slot = Wsm + WIMG;                       // 64 zero bytes: K-tail A lanes (CIN = 48) + slack behind the last weight row
    if (tid < 4) *(u32x4*)(zero_slot + tid * 16) = (u32x4){0, 0, 0, 0};

    const unsigned wimg0 = (unsigned)((size_t)blockIdx.y * NCHUNK * WIMG);
    u32x4 ra[NPP], rw[NWP];
    auto gload_w = [&](int cc) {                        // weight chunk image: linear 16-byte copy, no descriptors
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int q = tid + i * T;
            rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, q < WIMG / 16 ? (unsigned)(q * 16) : OOB_OFFSET, wimg0 + (unsigned)cc * WIMG, 0);
        }
    };
#ifdef PAM_DIAG
#define C3_STAMP(k) do { if ((a.dbg & 64) && tid == 0) a.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C3_STAMP(k) do { } while (0)
#endif
    C3_STAMP(0);
    f32x4 bias4[NTW];                                   // lane group g ends with channels n0 + 4*NTW*g + 4*j + r (see the epilogue)
#pragma unroll
    for (int j = 0; j < NTW; ++j) bias4[j] = a.bias ? *(const f32x4*)(a.bias + n0 + g * 4 * NTW + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!C3_DBG(1)) gload_w(0);                         // in flight while the patch descriptors are computed

    // ---- per-thread patch piece descriptors (fixed over the chunk loop) -------------------------------------------------
    unsigned goffA[NPP];
    int tid_v;
    auto descriptors = [&](int n_, int ty0_) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = tid_v + i * T;
            goffA[i] = OOB_OFFSET;
            if (q < npatch * PC8) {
                const int pp = q / PC8, c8 = q - pp * PC8;
                const int pyy = fdiv_small(pp, a.inv_pw), pxx = pp - pyy * PW;
                const int iy = ty0_ - 1 + pyy, ix = pxx - 1;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    goffA[i] = (unsigned)((((size_t)n_ * a.H + iy) * a.W + ix) * CIN * 2 + c8 * 16);
            }
        }
    };
    tid_v = tid;
    descriptors(n, ty0);
    auto gload_a = [&](int cc) {
        const unsigned so = (unsigned)(cc * CK * 2);
#pragma unroll
        for (int i = 0; i < NPP; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, goffA[i], goffA[i] == OOB_OFFSET ? 0 : so, 0);
    };
    auto gload = [&](int cc) { gload_w(cc); gload_a(cc); };
    auto lstore = [&](bool with_weights) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = tid_v + i * T;
            if (q < npatch * PC8) { const int pp = q / PC8, c8 = q - pp * PC8; *(u32x4*)(smem + (size_t)pp * PITCH_A + c8 * 16) = ra[i]; }
        }
        if (with_weights) {
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                const int q = tid + i * T;
                if (q < WIMG / 16) *(u32x4*)(Wsm + (size_t)q * 16) = rw[i];
            }
        }
    };

    f32x4 acc[MT][NTW];

    // slot of this lane in M tile i: p = wave*16*MT + i*16 + (lane & 15); A byte offset = p * PITCH_A (+ tap, + k slice)
    const int p_lane = wave * 16 * MT + (lane & 15);
    const char* al = smem + (size_t)p_lane * PITCH_A;
    const char* wl = Wsm + (size_t)(lane & 15) * PITCH_W;

    if (C3_DBG(16)) { if (goffA[0] == 12345u) a.out[0] = 1; return; }
    if (!C3_DBG(1)) gload_a(0);
    C3_STAMP(1);
    if constexpr (NCHUNK == 1) {                        // the slab's only weight chunk is staged once, outside the tile loop
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int q = tid + i * T;
            if (q < WIMG / 16) *(u32x4*)(Wsm + (size_t)q * 16) = rw[i];
        }
    }
    tid_v = tid;                                        // opaque per iteration: keeps per-piece addresses from being hoisted (VGPRs)
    for (bool first = true;; first = false) {           // tile loop (one pass unless persistent)
    if constexpr (NCHUNK == 1) asm volatile("" : "+v"(tid_v));
    const int vnext = vtile + (int)gridDim.x;
    const bool has_next = NCHUNK == 1 && vnext < ntiles;
    int n_nx = 0, ty0_nx = 0;
    // accumulators start from the bias (loaded first of all, so waiting for it never waits for the tile loads behind it)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = bias4[j];
    for (int cc = 0; cc < NCHUNK; ++cc) {
        if (cc > 0 || !first) __syncthreads();          // every wave is done reading the previous chunk / tile
        C3_STAMP(2 + 4 * cc);
        if (!C3_DBG(32)) lstore(NCHUNK > 1);
        C3_STAMP(3 + 4 * cc);
        __syncthreads();
        C3_STAMP(4 + 4 * cc);
        if (cc + 1 < NCHUNK && !C3_DBG(1)) gload(cc + 1);               // next chunk in flight under the MFMAs below
        if (cc == NCHUNK - 1 && a.res) {                                // residual tile in flight under the last chunk's MFMAs
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int p = wave * 16 * MT + i * 16 + (lane & 15);
                const int py = fdiv_small(p, a.inv_pw), px = p - py * PW;
                const bool ok = p < nslots && px < a.W;
                const unsigned o = ok ? (unsigned)(((((size_t)n * a.H + ty0 + py) * a.W + px) * a.Cout + n0 + g * 4 * NTW) * 2) : OOB_OFFSET;
                c3_row_load<NTW>(rs_res, o, g, rres[i]);
            }
        }
        if (C3_DBG(2)) continue;
        // K loop, software-pipelined by hand: the fragments of step s+1 are read from LDS while the MFMAs of step s issue
        // (with one wave per SIMD nothing else hides the ds_read latency)
        constexpr int NSTEP = (CIN == 48) ? 14 : 9 * (CK / 32);
        bf16x8 af[2][MT], bfr[2][NTW];
        auto ldfrag = [&](int st, bf16x8* af_, bf16x8* bf_) {
            if constexpr (CIN == 48) {
                const int k0 = 32 * st + 8 * g;                          // flattened (tap, c); an 8-slice never straddles taps
                const int t = k0 / 48, c = k0 - t * 48;
                const int ky = t / 3, kx = t - ky * 3;
                const bool zero = k0 >= 432;
                const unsigned aoff = (unsigned)((ky * PW + kx) * PITCH_A + c * 2);
#pragma unroll
                for (int j = 0; j < NTW; ++j) bf_[j] = *(const bf16x8*)(wl + (size_t)j * 16 * PITCH_W + k0 * 2);
#pragma unroll
                for (int i = 0; i < MT; ++i) af_[i] = *(const bf16x8*)(zero ? zero_slot : al + (size_t)i * 16 * PITCH_A + aoff);
            } else {
                constexpr int KS = CK / 32;
                const int t = st / KS, ks = st - t * KS;
                const int ky = t / 3, kx = t - ky * 3;
                const unsigned aoff = (unsigned)((ky * PW + kx) * PITCH_A + ks * 64 + g * 16);
#pragma unroll
                for (int j = 0; j < NTW; ++j) bf_[j] = *(const bf16x8*)(wl + (size_t)j * 16 * PITCH_W + st * 64 + g * 16);
#pragma unroll
                for (int i = 0; i < MT; ++i) af_[i] = *(const bf16x8*)(al + (size_t)i * 16 * PITCH_A + aoff);
            }
        };
        ldfrag(0, af[0], bfr[0]);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 1 < NSTEP) ldfrag(st + 1, af[(st + 1) & 1], bfr[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);          // keep the next step's ds_reads ahead of this step's MFMAs (hipcc sinks them otherwise)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[st & 1][j]),
                                                                       __builtin_bit_cast(bf16x8_t, af[st & 1][i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        C3_STAMP(5 + 4 * cc);
    }

    // ---- epilogue straight from the accumulators.  With the weights as the MFMA A operand the D tile has channels on
    // its rows, and the host packs the slab's weight rows so that row j*16 + 4g + r is channel n0 + 4*NTW*g + 4j + r: this
    // lane then holds 4*NTW CONTIGUOUS channels of pixel slot i*16 + (lane & 15) -> residual loads and stores are 16 bytes
    // wide (the store tail is issue-bound), and the 4 lane groups of a pixel cover the slab's 32*NTW contiguous bytes.
    C3_STAMP(60);
    if (C3_DBG(4)) { if (tid == 0) a.out[(size_t)blockIdx.x * 8] = (uint16_t)acc[0][0][0]; return; }
    if (has_next) {     // next TILE's patch goes in flight under this tile's epilogue (the fragment registers are free again by now)
        const int b2 = tile_of(vnext);
        n_nx = b2 / a.tiles_y; ty0_nx = (b2 - n_nx * a.tiles_y) * a.TH;
        descriptors(n_nx, ty0_nx);
        gload_a(0);
    }
    // RES / RELU are compile-time in the HRNet instantiations (a uniform branch picks one of four copies): per 4 values the
    // epilogue is then 4 unpack + 4 add (residual only), 2 v_cvt_pk_bf16_f32 and ReLU as ONE packed integer max per dword
    // (bf16 is sign-magnitude: max(int16, 0) clears exactly the negative values) -- the tail is VALU-issue bound.
    auto epilogue = [&](auto RESC, auto RELUC, auto GENC) {
        constexpr bool RES = decltype(RESC)::value, RELU = decltype(RELUC)::value, GEN = decltype(GENC)::value;
        typedef __attribute__((ext_vector_type(2))) short s16x2;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int p = wave * 16 * MT + i * 16 + (lane & 15);
            const int py = fdiv_small(p, a.inv_pw), px = p - py * PW;
            if (p < nslots && px < a.W) {
                uint32_t ov[2 * NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float rr[4] = {0.f, 0.f, 0.f, 0.f};
                    if (GEN ? (a.res != nullptr) : RES) {
                        rr[0] = __builtin_bit_cast(float, rres[i][2 * j] << 16); rr[1] = __builtin_bit_cast(float, rres[i][2 * j] & 0xffff0000u);
                        rr[2] = __builtin_bit_cast(float, rres[i][2 * j + 1] << 16); rr[3] = __builtin_bit_cast(float, rres[i][2 * j + 1] & 0xffff0000u);
                    }
                    if constexpr (GEN) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = epi_act(v[r], rr[r], a.relu);
                        ov[2 * j] = pack_bf16x2(v[0], v[1]); ov[2 * j + 1] = pack_bf16x2(v[2], v[3]);
                    } else {
                        if constexpr (RES) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += rr[r];
                        }
                        ov[2 * j] = pack_bf16x2(v[0], v[1]); ov[2 * j + 1] = pack_bf16x2(v[2], v[3]);
                        if constexpr (RELU) {
                            ov[2 * j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j]), (s16x2){0, 0}));
                            ov[2 * j + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j + 1]), (s16x2){0, 0}));
                        }
                    }
                }
                c3_row_store<NTW>(a.out + (((size_t)n * a.H + ty0 + py) * a.W + px) * a.Cout + n0 + g * 4 * NTW, g, ov);
            }
        }
    };
    typedef std::true_type T_; typedef std::false_type F_;
    if constexpr (c3_general_act(CIN)) {
        epilogue(F_{}, F_{}, T_{});
    } else if (a.res) {
        if (a.relu) epilogue(T_{}, T_{}, F_{}); else epilogue(T_{}, F_{}, F_{});
    } else {
        if (a.relu) epilogue(F_{}, T_{}, F_{}); else epilogue(F_{}, F_{}, F_{});
    }
    if (!has_next) break;
    vtile = vnext; n = n_nx; ty0 = ty0_nx; nslots = min(a.TH, a.H - ty0) * PW;
    }   // tile loop
    C3_STAMP(61);
}

static size_t c3_lds_bytes(int cin, int ntw, int npatch) {
    return ((((size_t)npatch + 2) * c3_pitch_a(cin) + 15) & ~(size_t)15) + (size_t)16 * ntw * c3_pitch_w(cin) + 64;
}
template <int CIN, int NTW, int MT, int NWAVES, int PMAX>
static int launch_c3_one(hipStream_t s, const C3Args& a) {
    const int npatch = (a.TH + 2) * (a.W + 2);
    if (npatch > PMAX) return PAM_E_ARG;
    dim3 grid(a.tiles_y * a.N, a.Cout / (16 * NTW));
    const size_t lds = c3_lds_bytes(CIN, NTW, npatch);
    if (lds > 150 * 1024) return PAM_E_ARG;
    if (CIN / c3_ck(CIN) == 1 && !C3_DBG(128)) {        // single-chunk layers: persistent workgroups (see the kernel)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv3x3<CIN, NTW, MT, NWAVES, PMAX>, 64 * NWAVES, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        // One workgroup per CU while a workgroup has only a few tiles to walk: the layer then leaves half of every CU's LDS and
        // registers to the kernels of the other branch streams (measured +2 % on the 20-crop forward; with two per CU the Cin-48
        // chain shuts the other branches out and they run after it).  Large batches fill the chip on their own: all resident slots.
#ifdef PAM_DIAG
        static const int cap = getenv("PAM_C3_PERSIST_SLOTS") ? atoi(getenv("PAM_C3_PERSIST_SLOTS")) : 0;     // tuning override
#else
        constexpr int cap = 0;
#endif
        int slots = 256 * per_cu / (int)grid.y / 8 * 8;
        const int few = cap > 0 ? cap : ((int)grid.x < 4 * 256 ? 256 : slots);
        if (slots > few) slots = few / 8 * 8;
        if (slots >= 8 && (int)grid.x > slots) grid.x = slots;
    }
    pam_launch(k_conv3x3<CIN, NTW, MT, NWAVES, PMAX>, grid, dim3(64 * NWAVES), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
// cfg = MT * 10 + NWAVES  (MT = 4; NWAVES in {2, 3, 4}); the patch bound PMAX is picked from the actual tile
template <int CIN, int NTW>
static int launch_c3(hipStream_t s, const C3Args& a, int cfg) {
    const int npatch = (a.TH + 2) * (a.W + 2);
    switch (cfg) {
        case 42: return npatch <= 160 ? launch_c3_one<CIN, NTW, 4, 2, 160>(s, a) : launch_c3_one<CIN, NTW, 4, 2, 288>(s, a);
        case 43: return npatch <= 160 ? launch_c3_one<CIN, NTW, 4, 3, 160>(s, a) : launch_c3_one<CIN, NTW, 4, 3, 352>(s, a);
        case 44: return npatch <= 288 ? launch_c3_one<CIN, NTW, 4, 4, 288>(s, a) : launch_c3_one<CIN, NTW, 4, 4, 416>(s, a);
        case 54:                                         // 5 M tiles per wave (320 slots): taller tiles -> a layer of <= 256 workgroups, each alone on its CU
            if constexpr (CIN == 96 || CIN == 192) return launch_c3_one<CIN, NTW, 5, 4, 416>(s, a);
            else return PAM_E_ARG;
    }
    return PAM_E_ARG;
}

// choose rows per tile and the wave shape.  Measured (tools/tune_conv3x3.py): 4 M-tiles per wave beat 8, and the best
// tile is the tallest one that fits the widest block; small images take the smallest block that holds them whole.
static void pick_rows(int N, int H, int W, int Cout, int ntw, int& TH, int& cfg) {
    const int PW = W + 2;
    (void)N; (void)Cout; (void)ntw;
    if (H * PW <= 128) { TH = H; cfg = 42; return; }
    if (H * PW <= 192) { TH = H; cfg = 43; return; }
    cfg = 44;
    TH = 256 / PW;
    if (TH < 1) TH = 1;
    if (TH > H) TH = H;
    // prefer a divisor of H close to the cap (no ragged last tile) when it costs < 15 % of the tile height
    for (int t = TH; t >= 1 && t * 100 >= TH * 85; --t) if (H % t == 0) { TH = t; break; }
    // wide rows (the detector's 104-wide layers): shrink the tile until its patch fits, then the block to the slots left
    while (TH > 1 && (TH + 2) * PW > 416) --TH;
    if (TH * PW <= 128 && (TH + 2) * PW <= 288) cfg = 42;
    else if (TH * PW <= 192 && (TH + 2) * PW <= 352) cfg = 43;
}

// output channels per workgroup slab of k_conv3x3 (the host packs the weight images with the same number).  The deep, small
// images (24x18, 12x9) have too few pixel tiles to fill 256 CUs, so their slabs are narrower: more, shorter workgroups.
extern "C" int pam_conv3x3_slab(int H, int W, int Cin, int Cout) {
    const int wide = (Cout % 48 == 0) ? 48 : 64;
    if (Cin < 192) return wide;
    if (Cout % 48 != 0) return H * W <= 1024 ? 32 : 64;                            // Darknet's 256- / 512-channel 3x3 layers
#ifdef PAM_DIAG
    const int env = getenv("PAM_C3_SLAB") ? atoi(getenv("PAM_C3_SLAB")) : 0;      // tuning hook
    if (env == 16 || env == 32 || env == 48) return env;
#endif
    return H * W <= 128 ? 16 : (H * W <= 512 ? 32 : 48);
}

// ====================================================================================================================
// k_conv3x3s: the same convolution (Cin = 192 / 384, activation codes 0 / 1) with SPECIALISED waves.  In k_conv3x3 at 20 crops a
// workgroup is alone on its CU, so every wave pays for its own operand traffic in its own instruction stream: the burst of
// buffer_loads for the next chunk holds the MFMA stream for ~1.6 k cycles per chunk, the register -> LDS pass for another ~1 k, against
// 1.7-2.3 k cycles of MFMAs (tools/stamp_conv.py knock-outs).  Here waves 4-7 only move bytes -- LDS-DMA (global_load_lds_dwordx4: no
// registers, no ds_write pass) of 32-channel chunks into a ring of NBUF chunk buffers, NBUF - 1 chunks ahead -- and waves 0-3 only
// read fragments and multiply (one per SIMD, MT x NTW accumulator tiles each as before).  One raw s_barrier per chunk: the loaders
// arrive once chunk k has landed (counted vmcnt, younger chunks stay in flight), the multipliers once they are done with chunk
// k - 1, whose buffer the loaders then refill.
//   chunk buffer = [PMAX patch slots][64 B] + [9 taps][BN rows][64 B], both dense (a DMA piece is 1 KiB = 16 rows, lane-linear) with
//   the 16-B piece g of row r stored at position g ^ ((r >> 1) & 2): conflict-free ds_read_b128 for a 16-row window at ANY row offset
//   (tools/lds_sim.py).  Patch rows outside the image are fetched from a page of zeros; the weight images are host-packed in exactly
//   this layout (pam_conv3x3_layout() == 1).  The residual is added to the bias before the K loop (its loads run beside the first
//   chunk's DMA), so the epilogue is convert + ReLU + store.
// ====================================================================================================================
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
__device__ __attribute__((aligned(64))) const uint32_t g_c3_zero[16 + 16 * 16] = {0};   // 64 B + the largest chunk offset (Cin = 512)

// scheduling pattern of one k-step: the next step's NR LDS reads alternate with the first NR of this step's NM MFMAs, the other MFMAs
// follow (slack for the last read's latency); the builtin takes literal counts
template <int NM, int NR, int... R>
__device__ __forceinline__ void c3s_spread(std::integer_sequence<int, R...>) {
    (((void)R, __builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)), ...);
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
}

template <int CIN, int NTW, int MT, int PMAX, int NBUF, bool GEN = false>
__global__ __launch_bounds__(512, 1) void k_conv3x3s(C3Args a) {
    constexpr int BN = 16 * NTW, NCHUNK = CIN / 32;
    constexpr int PIMG = PMAX * 64, WIMG = 9 * BN * 64, BUF = PIMG + WIMG;
    constexpr int PPW = PMAX / 64, WPIECES = WIMG / 1024, WPW = (WPIECES + 3) / 4, NPER = PPW + WPW;   // DMA pieces per loader wave per chunk
    static_assert(PMAX % 64 == 0 && BUF % 512 == 0 && WIMG % 1024 == 0 && NPER * (NBUF - 1) <= 60 && NBUF >= 2 && NBUF <= 4, "ring shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.tiles_y * a.N;
    const int bx = [&] {                                // XCD-aware tile order (see k_conv3x3)
        const int v = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, loc = v >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }();
    const int n = bx / a.tiles_y, ty0 = (bx - n * a.tiles_y) * a.TH;
    const int PW = a.W + 2, npatch = (a.TH + 2) * PW;
    const int nslots = min(a.TH, a.H - ty0) * PW;
    const int n0 = blockIdx.y * BN;

    if (wave >= 4) {
        // ---- loader waves ---------------------------------------------------------------------------------------------------
        const int lw = wave - 4;
        const char* psrc[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int s = (lw + 4 * i) * 16 + (lane >> 2);                               // slot this lane fills in piece lw + 4 i
            const int gsrc = (lane & 3) ^ ((s >> 1) & 2);
            const int py = fdiv_small(s, a.inv_pw), px = s - py * PW;
            const int iy = ty0 - 1 + py, ix = px - 1;
            const bool ok = s < npatch && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            psrc[i] = ok ? (const char*)a.in + (((size_t)n * a.H + iy) * a.W + ix) * (CIN * 2) + gsrc * 16 : (const char*)g_c3_zero + (lane & 3) * 16;
        }
        const char* wsrc = (const char*)a.wimg + (size_t)blockIdx.y * NCHUNK * WIMG + lane * 16;
        auto issue = [&](int c) {
            char* dst = smem + (size_t)((unsigned)c % (unsigned)NBUF) * BUF;
#pragma unroll
            for (int i = 0; i < (PPW > WPW ? PPW : WPW); ++i) {
                if (i < WPW) {
                    const int j = min(lw + 4 * i, WPIECES - 1);                          // a wave short of a piece re-sends the last one
                    __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + (size_t)c * WIMG + j * 1024), (lds_void*)(dst + PIMG + j * 1024), 16, 0, 0);
                }
                if (i < PPW)
                    __builtin_amdgcn_global_load_lds((glb_void*)(psrc[i] + c * 64), (lds_void*)(dst + (lw + 4 * i) * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c)
            if (c < NCHUNK) issue(c);
        for (int k = 0; k < NCHUNK; ++k) {
            const int fly = min(NCHUNK - 1 - k, NBUF - 2);                               // younger chunks that may stay in flight
            if (fly <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPER) : "memory");
            asm volatile("s_barrier" ::: "memory");
            if (k + NBUF - 1 < NCHUNK) issue(k + NBUF - 1);
        }
        return;
    }

    // ---- multiplier waves ---------------------------------------------------------------------------------------------------
    C3_STAMP(0);
    f32x4 acc[MT][NTW];
    uint32_t gres[GEN ? MT : 1][2 * NTW];                 // GEN (Darknet activation codes): the residual rows stay in registers until the epilogue
    {
        f32x4 bias4[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) bias4[j] = a.bias ? *(const f32x4*)(a.bias + n0 + g * 4 * NTW + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (GEN && a.res) {
            const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int p = wave * 16 * MT + i * 16 + (lane & 15);
                const int py = fdiv_small(p, a.inv_pw), px = p - py * PW;
                const bool ok = p < nslots && px < a.W;
                const unsigned o = ok ? (unsigned)(((((size_t)n * a.H + ty0 + py) * a.W + px) * a.Cout + n0 + g * 4 * NTW) * 2) : OOB_OFFSET;
                c3_row_load<NTW>(rs_res, o, g, gres[GEN ? i : 0]);
            }
        }
        if (!GEN && a.res) {
            const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
            uint32_t rres[MT][2 * NTW];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int p = wave * 16 * MT + i * 16 + (lane & 15);
                const int py = fdiv_small(p, a.inv_pw), px = p - py * PW;
                const bool ok = p < nslots && px < a.W;
                const unsigned o = ok ? (unsigned)(((((size_t)n * a.H + ty0 + py) * a.W + px) * a.Cout + n0 + g * 4 * NTW) * 2) : OOB_OFFSET;
                c3_row_load<NTW>(rs_res, o, g, rres[i]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    acc[i][j][0] = bias4[j][0] + __builtin_bit_cast(float, rres[i][2 * j] << 16);
                    acc[i][j][1] = bias4[j][1] + __builtin_bit_cast(float, rres[i][2 * j] & 0xffff0000u);
                    acc[i][j][2] = bias4[j][2] + __builtin_bit_cast(float, rres[i][2 * j + 1] << 16);
                    acc[i][j][3] = bias4[j][3] + __builtin_bit_cast(float, rres[i][2 * j + 1] & 0xffff0000u);
                }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = bias4[j];
        }
    }
    // LDS byte offsets (inside a chunk buffer) of this lane's patch fragment for M tile i and tap t, swizzle included
    // M tile i sits 16 slots = 1024 bytes behind tile 0 and has the same swizzle (it depends on bit 2 of the slot only): nine offsets
    // + immediates instead of MT x 9 registers
    unsigned aoff0[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int s = wave * 16 * MT + (lane & 15) + (t / 3) * PW + (t % 3);
        aoff0[t] = (unsigned)(s * 64 + ((g ^ ((s >> 1) & 2)) << 4));
    }
    const unsigned woff = (unsigned)(PIMG + (lane & 15) * 64 + ((g ^ ((lane >> 1) & 2)) << 4));   // row j*16 + (lane & 15): bit 2 of the row = bit 2 of the lane

    // One software pipeline over all NCHUNK * 9 k-steps: the fragments of step s + 1 are read while the MFMAs of step s issue, and a
    // chunk boundary (drain this wave's LDS reads, barrier, first reads of the next chunk) sits between the last tap's reads and its
    // MFMAs, so the barrier and the first read latency of a chunk hide under 12 MFMAs.  Fragment slots alternate with (k + t) & 1.
    bf16x8 af[2][MT], bfr[2][NTW];
    auto ldfrag = [&](int k, int t, bf16x8* af_, bf16x8* bf_) {
        const char* buf = smem + (size_t)((unsigned)k % (unsigned)NBUF) * BUF;
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf_[j] = *(const bf16x8*)(buf + woff + (t * BN + j * 16) * 64);
#ifdef PAM_KO_PIXREADS                                     // timing knock-out (wrong results): pixel fragments read for the kx = 0 taps only
        if (t % 3 == 0)
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) af_[i] = *(const bf16x8*)(buf + aoff0[t] + i * 1024);
    };
    auto chunk = [&](int k, auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
        C3_STAMP(3 + 3 * (k & 15));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            constexpr int dummy = 0; (void)dummy;
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
            // issue order inside the step: the next step's MT + NTW fragment reads spread between this step's MFMAs (a burst of reads
            // ahead of the MFMAs holds the wave's issue slot ~100 cycles per step with the matrix pipe idle)
            if constexpr (MT * NTW >= MT + NTW) c3s_spread<MT * NTW, MT + NTW>(std::make_integer_sequence<int, MT + NTW>{});   // (16-channel slabs: more reads than MFMAs, the compiler's order)
            __builtin_amdgcn_sched_barrier(0);
        }
        C3_STAMP(4 + 3 * (k & 15));
    };
    asm volatile("s_barrier" ::: "memory");                                              // chunk 0 has landed (and is visible)
    ldfrag(0, 0, af[0], bfr[0]);
    for (int k = 0; k < NCHUNK; k += 2) {
        chunk(k, std::integral_constant<int, 0>{});
        if (k + 1 < NCHUNK) chunk(k + 1, std::integral_constant<int, 1>{});
    }
    C3_STAMP(60);

    // ---- epilogue straight from the accumulators (row permutation of the slab as in k_conv3x3: 4*NTW contiguous channels per lane)
    typedef __attribute__((ext_vector_type(2))) short s16x2;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int p = wave * 16 * MT + i * 16 + (lane & 15);
        const int py = fdiv_small(p, a.inv_pw), px = p - py * PW;
        if (p < nslots && px < a.W) {
            uint32_t ov[2 * NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if constexpr (GEN) {                     // epi_act's arithmetic: act & 3 = 0 linear / 1 ReLU / 2 leaky; act & 4: the residual is added after it
                    const uint32_t r01 = a.res ? gres[GEN ? i : 0][2 * j] : 0u, r23 = a.res ? gres[GEN ? i : 0][2 * j + 1] : 0u;
                    acc[i][j][0] = epi_act(acc[i][j][0], __builtin_bit_cast(float, r01 << 16), a.relu);
                    acc[i][j][1] = epi_act(acc[i][j][1], __builtin_bit_cast(float, r01 & 0xffff0000u), a.relu);
                    acc[i][j][2] = epi_act(acc[i][j][2], __builtin_bit_cast(float, r23 << 16), a.relu);
                    acc[i][j][3] = epi_act(acc[i][j][3], __builtin_bit_cast(float, r23 & 0xffff0000u), a.relu);
                }
                ov[2 * j] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); ov[2 * j + 1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                if (!GEN && a.relu) {
                    ov[2 * j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j]), (s16x2){0, 0}));
                    ov[2 * j + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j + 1]), (s16x2){0, 0}));
                }
            }
            c3_row_store<NTW>(a.out + (((size_t)n * a.H + ty0 + py) * a.W + px) * a.Cout + n0 + g * 4 * NTW, g, ov);
        }
    }
    C3_STAMP(61);
}

// Which layers the streamed kernel takes, and its tile.  The register tile of a multiplier wave sets the LDS traffic per MFMA
// ((MT + NTW) fragment reads per MT * NTW MFMAs): with 32-channel slabs the four multipliers' reads take 75-85 % of the LDS cycles of
// their MFMAs and a chunk runs at half the matrix rate; slabs of 64 channels halve that and give the same latency
// from HALF the workgroups (120-160 at 20 crops), which leaves the other CUs to the other branches' kernels.  One workgroup per CU
// (LDS ring): the tile is the tallest whole-row tile that fits (a divisor of H when that costs < 15 %).
// Cin 96 stays on k_conv3x3: its streamed form (96-channel slabs, 160 workgroups) is as fast alone (12.9 vs 13.0 us) but 3-5 % slower
// end to end -- a 150 KB workgroup shuts the other branches out of its CU, and the 48 x 36 layers have enough tiles to fill the chip.
// c96_slab: 0 = 96 -> 96 layers stay on k_conv3x3; 48 = they run here with 48-channel slabs and a two-slot ring (the caller states it per
// launch: tile_cfg -5 of pam_conv2d_nhwc_bf16_ex, and packs the weight image for that slab: pam_conv3x3_layout_ex)
static bool c3s_pick(int H, int W, int Cin, int Cout, int& TH, int& mt, int& pmax, int& ntw, int c96_slab = 0) {
    // layer1 / transition1 of HRNet (64 -> 64 and 256 -> 48 at 96 x 72: ReLU layers; the detector's 64- and 256-channel 3x3 layers have
    // other widths and a leaky activation and stay on k_conv3x3): two rounds of 480 workgroups, still 24 -> 16 us and 59 -> 30 us
    const bool l1 = (Cin == 64 && Cout == 64) || (Cin == 256 && Cout == 48);
    const bool c96 = Cin == 96 && Cout == 96 && c96_slab == 48;
    if (Cin != 192 && Cin != 384 && !l1 && !c96) return false;
#ifdef PAM_DIAG
    static const int mask = getenv("PAM_C3S_MASK") ? atoi(getenv("PAM_C3S_MASK")) : 14;      // tuning hook: 2 = Cin 192, 4 = 384, 8 = 64 / 256
    if (!(mask & (Cin == 192 ? 2 : (Cin == 384 ? 4 : 8)))) return false;
#endif
    const int PW = W + 2, smax = 320, pcap = l1 ? 448 : 384;
    // rows per tile: the height that costs the fewest M tiles over the image (a tile always multiplies whole 64-slot wave shares, 3 to
    // 5 of them, and a ragged last tile multiplies as much as a full one); ties go to the taller tile = fewer workgroups
    TH = 0; mt = 0; pmax = 0;
    long best = 0;
    for (int t = (H < smax / PW ? H : smax / PW); t >= 1; --t) {
        const int sl = t * PW, np = (t + 2) * PW;
        if (np > pcap) continue;
        // the instantiated (M tiles per wave, patch slots) shapes: (3, 192), (4, 320), (5, 384 | 448) -- the smallest that holds the tile
        if (sl > 320 || np > 448) continue;
        const int m = (sl <= 192 && np <= 192) ? 3 : ((sl <= 256 && np <= 320) ? 4 : 5);
        const long cost = (long)((H + t - 1) / t) * m;
        if (TH == 0 || cost < best) { TH = t; best = cost; mt = m; pmax = m == 3 ? 192 : (m == 4 ? 320 : (np <= 384 ? 384 : 448)); }
    }
    if (TH < 1) return false;
    const int bn = c96 ? c96_slab : (Cout == 48 ? 48 : 64);
    if (Cout % bn != 0) return false;
    ntw = bn / 16;
    if (c96) return true;
    // only shapes launch_c3s() instantiates: 64-channel slabs for Cin 192 / 384 (and 64 -> 64), the 48-channel slab for 256 -> 48;
    // anything else (e.g. Cin 192 -> Cout 48) stays on k_conv3x3 / the implicit GEMM and keeps the classic weight image
    if (Cin == 256 ? ntw != 3 : ntw != 4) return false;
    return true;
}
// Darknet's 3x3 layers (leaky activation, shortcut added after it) on the streamed kernel: Cin 128 / 256 / 512 with 64-channel slabs, for
// the (M tiles, patch) shapes instantiated below; anything else stays on the classic kernel.  Round 5: the detector's 29 such layers ran
// at 12-17 % of the MFMA roof on k_conv3x3 (26 / 18.6 / 27.5 us at 52 x 52 / 26 x 26 / 13 x 13 x 5 views).
static bool c3s_pick_gen(int H, int W, int Cin, int Cout, int& TH, int& mt, int& pmax) {
    if ((Cin != 128 && Cin != 256 && Cin != 512) || Cout % 64 != 0) return false;
    const int PW = W + 2;
    TH = 0; mt = 0; pmax = 0;
    long best = 0;
    for (int t = (H < 320 / PW ? H : 320 / PW); t >= 1; --t) {
        const int sl = t * PW, np = (t + 2) * PW;
        if (sl > 320 || np > 384) continue;
        const int m = (sl <= 256 && np <= 320) ? 4 : 5;
        const long cost = (long)((H + t - 1) / t) * m;
        if (TH == 0 || cost < best) { TH = t; best = cost; mt = m; pmax = m == 4 ? 320 : 384; }
    }
    return TH >= 1;
}
// slab width of the general-activation form: 64 channels for Cin = 128 (52 x 52 maps: 220 workgroups), 32 for the deeper, smaller maps
// (26 x 26, 13 x 13: 80-120 workgroups of 64-channel slabs left most of the chip idle while each streamed 300-590 KB of weights)
static int c3s_gen_slab(int Cin) { return Cin == 128 ? 64 : 32; }
extern "C" int pam_conv3x3_layout_gen(int H, int W, int Cin, int Cout) {
    int th, mt, pmax;
    return c3s_pick_gen(H, W, Cin, Cout, th, mt, pmax) ? c3s_gen_slab(Cin) : 0;
}
template <int CIN, int NTW, int MT, int PMAX>
static int launch_c3s_gen_one(hipStream_t s, const C3Args& a) {
    constexpr size_t lds = (size_t)2 * (PMAX * 64 + 9 * 16 * NTW * 64);
    if (!pam_max_dynamic_lds((const void*)k_conv3x3s<CIN, NTW, MT, PMAX, 2, true>, (int)lds)) return PAM_E_HIP;
    pam_launch(k_conv3x3s<CIN, NTW, MT, PMAX, 2, true>, dim3(a.tiles_y * a.N, a.Cout / (16 * NTW)), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
static int launch_c3s_gen(hipStream_t s, const C3Args& a, int Cin, int mt) {
    switch (Cin * 10 + mt) {
        case 1284: return launch_c3s_gen_one<128, 4, 4, 320>(s, a);
        case 1285: return launch_c3s_gen_one<128, 4, 5, 384>(s, a);
        case 2564: return launch_c3s_gen_one<256, 2, 4, 320>(s, a);
        case 2565: return launch_c3s_gen_one<256, 2, 5, 384>(s, a);
        case 5124: return launch_c3s_gen_one<512, 2, 4, 320>(s, a);
        case 5125: return launch_c3s_gen_one<512, 2, 5, 384>(s, a);
    }
    return PAM_E_ARG;
}
extern "C" int pam_conv3x3_layout_ex(int H, int W, int Cin, int Cout, int c96_slab) {
    int th, mt, pmax, ntw;
    return c3s_pick(H, W, Cin, Cout, th, mt, pmax, ntw, c96_slab) ? 16 * ntw : 0;
}
extern "C" int pam_conv3x3_layout(int H, int W, int Cin, int Cout) { return pam_conv3x3_layout_ex(H, W, Cin, Cout, 0); }
// 32 = this 192- / 384-channel layer can also run with 32-channel slabs (tile_cfg -8 of pam_conv2d_nhwc_bf16_ex; image packed for that width)
extern "C" int pam_conv3x3_layout_small(int H, int W, int Cin, int Cout) {
    int th, mt, pmax, ntw;
    return ((Cin == 192 || Cin == 384) && Cout % 64 == 0 && c3s_pick(H, W, Cin, Cout, th, mt, pmax, ntw, 0) && pmax != 448) ? 32 : 0;
}
template <int CIN, int NTW, int MT, int PMAX, int NBUF>
static int launch_c3s_one(hipStream_t s, const C3Args& a) {
    constexpr size_t lds = (size_t)NBUF * (PMAX * 64 + 9 * 16 * NTW * 64);
    static_assert(lds <= 160 * 1024, "LDS");
    if (!pam_max_dynamic_lds((const void*)k_conv3x3s<CIN, NTW, MT, PMAX, NBUF>, (int)lds)) return PAM_E_HIP;
    pam_launch(k_conv3x3s<CIN, NTW, MT, PMAX, NBUF>, dim3(a.tiles_y * a.N, a.Cout / (16 * NTW)), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
static int launch_c3s(hipStream_t s, const C3Args& a, int Cin, int ntw, int mt, int pmax) {
#ifdef PAM_DIAG
    static const int nbuf = getenv("PAM_C3S_NBUF") ? atoi(getenv("PAM_C3S_NBUF")) : 0;       // tuning hook
    if (Cin == 384 && ntw == 4 && mt == 3 && nbuf == 2) return launch_c3s_one<384, 4, 3, 192, 2>(s, a);
#endif
    if (Cin == 64 || Cin == 256) {                       // layer1 / transition1: every (M tiles, patch) shape the pick can return for them
        switch ((Cin == 64 ? 0 : 10) + (pmax == 448 ? 6 : mt)) {
            case 3: return launch_c3s_one<64, 4, 3, 192, 2>(s, a);
            case 4: return launch_c3s_one<64, 4, 4, 320, 2>(s, a);
            case 5: return launch_c3s_one<64, 4, 5, 384, 2>(s, a);
            case 6: return launch_c3s_one<64, 4, 5, 448, 2>(s, a);
            case 13: return launch_c3s_one<256, 3, 3, 192, 3>(s, a);
            case 14: return launch_c3s_one<256, 3, 4, 320, 2>(s, a);
            case 15: return launch_c3s_one<256, 3, 5, 384, 2>(s, a);
            case 16: return launch_c3s_one<256, 3, 5, 448, 2>(s, a);
        }
        return PAM_E_ARG;
    }
    if (pmax == 448) return PAM_E_ARG;
    switch (Cin * 100 + ntw * 10 + mt) {
        case 9633: return launch_c3s_one<96, 3, 3, 192, 2>(s, a);
        case 9634: return launch_c3s_one<96, 3, 4, 320, 2>(s, a);
        case 9635: return launch_c3s_one<96, 3, 5, 384, 2>(s, a);
        case 19243: return launch_c3s_one<192, 4, 3, 192, 3>(s, a);
        case 19244: return launch_c3s_one<192, 4, 4, 320, 2>(s, a);
        case 19245: return launch_c3s_one<192, 4, 5, 384, 2>(s, a);
        case 38443: return launch_c3s_one<384, 4, 3, 192, 3>(s, a);
        case 38444: return launch_c3s_one<384, 4, 4, 320, 2>(s, a);
        case 38445: return launch_c3s_one<384, 4, 5, 384, 2>(s, a);
        // 32-channel slabs (tile_cfg -8: forwards of a few crops, where a launch is as long as ONE workgroup): twice the workgroups, half
        // the MFMAs and 18 instead of 36 KB of weights per chunk each, three chunk buffers
        case 19223: return launch_c3s_one<192, 2, 3, 192, 3>(s, a);
        case 19224: return launch_c3s_one<192, 2, 4, 320, 3>(s, a);
        case 19225: return launch_c3s_one<192, 2, 5, 384, 3>(s, a);
        case 38423: return launch_c3s_one<384, 2, 3, 192, 3>(s, a);
        case 38424: return launch_c3s_one<384, 2, 4, 320, 3>(s, a);
        case 38425: return launch_c3s_one<384, 2, 5, 384, 3>(s, a);
    }
    return PAM_E_ARG;
}

// ====================================================================================================================
// k_conv_gs: the implicit GEMM (1x1 and strided 3x3 layers) with the wave roles of k_conv3x3s.  Waves 4-7 gather the im2col tile
// -- 256 output pixels x 64 K values per chunk, every 16-byte piece fetched from its own address (or from a page of zeros: padding,
// K tail, rows past M) -- and the slab's weight rows straight from the packed [Cout][Kpad] matrix, by LDS-DMA into a ring of NBUF chunk
// buffers; waves 0-3 read fragments and multiply (64 pixels x 16*NTW channels each).  Rows are dense 128 bytes with the 16-byte
// piece q of row r at position q ^ ((r >> 1) & 7): conflict-free ds_read_b128 for a 16-row window x 4 k-groups, and a lane's
// fragment addresses are constants + immediates.  One s_barrier per chunk (loaders: chunk k has landed; multipliers: chunk k - 1 is
// consumed), residual folded into the bias before the K loop, epilogue = convert + ReLU (channels >= relu_from) + 16-byte stores.
// ====================================================================================================================
template <int NTW, int NBUF, bool RES, int BM>
__global__ __launch_bounds__(512, 1) void k_conv_gs(ConvArgs a) {
    constexpr int MTW = BM / 64;                            // 16-pixel tiles per multiplier wave (4 at BM = 256, 2 at 128, 1 at 64)
    constexpr int BN = 16 * NTW, AIMG = BM * 128, BIMG = BN * 128, BUF = AIMG + BIMG;
    constexpr int APW = AIMG / 1024 / 4, BPIECES = BIMG / 1024, BPW = (BPIECES + 3) / 4, NPER = APW + BPW;
    static_assert(NPER * (NBUF - 1) <= 60 && NBUF >= 2 && NBUF <= 6 && (BM == 256 || BM == 128 || BM == 64), "ring shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = a.Kpad >> 6;
    // PERSISTENT: workgroup b walks tiles b, b + gridDim.x, ... (tile = (256-pixel block, slab), slabs of one block adjacent so that
    // they fetch the same pixels at about the same time); the chunk ring runs on across tile boundaries, so a tile's loads overlap
    // the previous tile's multiplies and epilogue -- with one workgroup per CU nothing else would
    const int nslab = a.Cout / BN, ntile = ((a.M + BM - 1) / BM) * nslab;
    const int mine = (ntile - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = mine * nchunks;
    // XCD-aware order: workgroups b, b + 8, ... share an XCD and its L2 (gridDim.x is a multiple of 8 whenever a workgroup walks more
    // than one tile), so every XCD gets a contiguous run of the (block, slab) list: the slabs of one pixel block then gather the same
    // input pixels through ONE L2 instead of pulling them over the fabric once per XCD
    auto tile_of = [&](int t) {
        const int v = (int)blockIdx.x + t * (int)gridDim.x, qn = ntile >> 3, rn = ntile & 7, xcd = v & 7, loc = v >> 3;
        return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + loc;
    };

    if (wave >= 4) {
        // ---- loader waves ---------------------------------------------------------------------------------------------------
        const int lw = wave - 4;
        // A piece lw + 4 i covers rows 8 (lw + 4 i) .. + 7; this lane fills row 8 (lw + 4 i) + (lane >> 3), position lane & 7, i.e. the
        // logical piece q = (lane & 7) ^ ((row >> 1) & 7) -- the same q for all of its rows
        const int q = (lane & 7) ^ ((4 * lw + (lane >> 4)) & 7);
        int base[APW]; unsigned mask[APW];
        const char* bsrc[BPW]; int bdst[BPW];
        const int hw = a.Ho * a.Wo;
        const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)a.Wo;
        auto setup_tile = [&](int t) __attribute__((always_inline)) {
            const int T = tile_of(t), m0 = (T / nslab) * BM, n0 = (T % nslab) * BN;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int m = m0 + (lw + 4 * i) * 8 + (lane >> 3);
                base[i] = 0; mask[i] = 0;
                if (m < a.M) {
                    // m -> (n, oy, ox) by float reciprocals + one integer correction (exact below 2^24; integer division is ~40 instructions)
                    int n = (int)((float)m * inv_hw), r = m - n * hw;
                    if (r < 0) { --n; r += hw; } else if (r >= hw) { ++n; r -= hw; }
                    int oy = (int)((float)r * inv_wo), ox = r - oy * a.Wo;
                    if (ox < 0) { --oy; ox += a.Wo; } else if (ox >= a.Wo) { ++oy; ox -= a.Wo; }
                    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                    base[i] = (int)((((long)n * a.H + iy0) * a.W + ix0) * a.in_cs * 2);    // may be "before" the tensor; used with valid taps only
                    unsigned cols = 0, mk = 0;                                            // valid taps = valid rows x valid columns
                    for (int kx = 0; kx < a.KW; ++kx) if ((unsigned)(ix0 + kx) < (unsigned)a.W) cols |= 1u << kx;
                    for (int ky = 0; ky < a.KH; ++ky) if ((unsigned)(iy0 + ky) < (unsigned)a.H) mk |= cols << (ky * a.KW);
                    mask[i] = mk;
                }
            }
            // B piece j covers slab rows 8 j .. + 7; LDS row jt*16 + qq holds output channel 4*NTW*(qq >> 2) + 4*jt + (qq & 3) of the slab
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int j = min(lw + 4 * i, BPIECES - 1);                              // a wave short of a piece re-sends the last one
                const int row = j * 8 + (lane >> 3), qq = row & 15, jt = row >> 4;
                const int ch = n0 + 4 * NTW * (qq >> 2) + 4 * jt + (qq & 3);
                const int qb = (lane & 7) ^ ((row >> 1) & 7);
                bsrc[i] = (const char*)a.w + ((size_t)ch * a.Kpad + qb * 8) * 2;
                bdst[i] = AIMG + j * 1024;
            }
        };
        const float inv_cin = 1.0f / (float)a.Cin;
        int issued = 0, it = 0, ic = 0;                                                  // issue pointer: chunk ic of my tile it
        auto issue_next = [&]() __attribute__((always_inline)) {
            char* dst = smem + (size_t)((unsigned)issued % (unsigned)NBUF) * BUF;
            const int k0 = ic * 64 + q * 8;                                              // this lane's 8 K values: one tap, 8 channels
            const int tap = (int)(((float)k0 + 0.5f) * inv_cin), cc = k0 - tap * a.Cin;
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            const int delta = ((ky * a.W + kx) * a.in_cs + cc) * 2;
            const unsigned bit = k0 < a.Ktot ? 1u << tap : 0u;
#pragma unroll
            for (int i = 0; i < BPW; ++i)
                __builtin_amdgcn_global_load_lds((glb_void*)(bsrc[i] + ic * 128), (lds_void*)(dst + bdst[i]), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const char* src = (mask[i] & bit) ? (const char*)a.in + (long)base[i] + delta : (const char*)g_c3_zero;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + (lw + 4 * i) * 1024), 16, 0, 0);
            }
            ++issued;
            if (++ic == nchunks) { ic = 0; if (++it < mine) setup_tile(it); }
        };
        setup_tile(0);
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c)
            if (issued < total) issue_next();
        for (int k = 0; k < total; ++k) {
            const int fly = min(total - 1 - k, NBUF - 2);                                // younger chunks that may stay in flight
            if (fly <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
            else if (fly == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPER) : "memory");
            else if (fly == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF >= 5 ? 3 * NPER : 0) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF >= 6 ? 4 * NPER : 0) : "memory");
            asm volatile("s_barrier" ::: "memory");
            if (issued < total) issue_next();
        }
        return;
    }

    // ---- multiplier waves ---------------------------------------------------------------------------------------------------
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, (int)((size_t)a.M * a.Cout * 2), 0x00020000);
    uint32_t rres[MTW][2 * NTW];
    auto res_load = [&](int t) {                                                         // residual rows of my tile t -> registers (in flight)
        const int T = tile_of(t), mw0 = (T / nslab) * BM + wave * 16 * MTW, n0 = (T % nslab) * BN;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int m = mw0 + i * 16 + (lane & 15);
            c3_row_load<NTW>(rs_res, m < a.M ? (unsigned)(((size_t)m * a.Cout + n0 + g * 4 * NTW) * 2) : OOB_OFFSET, g, rres[i]);
        }
    };
    if (RES && a.res) res_load(0);
    // fragment addresses: row (lane & 15) of a 16-row window, k-group g of k-step ks -> piece (4 ks + g) ^ ((lane & 15) >> 1)
    const unsigned sw = (unsigned)((g ^ ((lane & 15) >> 1)) << 4);                       // ks = 0; ks = 1 is sw ^ 64
    const unsigned aoff0 = (unsigned)((wave * 16 * MTW + (lane & 15)) * 128) + sw;
    const unsigned boff0 = (unsigned)(AIMG + (lane & 15) * 128) + sw;
    f32x4 acc[MTW][NTW];
    bf16x8 af[2][MTW], bfr[2][NTW];
    auto ldfrag = [&](int G, int ks, bf16x8* af_, bf16x8* bf_) {
        const char* buf = smem + (size_t)((unsigned)G % (unsigned)NBUF) * BUF;
        const unsigned x = ks ? 64u : 0u;
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf_[j] = *(const bf16x8*)(buf + ((boff0 ^ x) + j * 16 * 128));
#pragma unroll
        for (int i = 0; i < MTW; ++i) af_[i] = *(const bf16x8*)(buf + ((aoff0 ^ x) + i * 16 * 128));
    };
    auto mfmas = [&](const bf16x8* af_, const bf16x8* bf_) {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf_[j]), __builtin_bit_cast(bf16x8_t, af_[i]), acc[i][j], 0, 0, 0);
    };
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    asm volatile("s_barrier" ::: "memory");                                              // my first chunk has landed (and is visible)
    int G = 0;
    for (int t = 0; t < mine; ++t) {
        const int T = tile_of(t), mw0 = (T / nslab) * BM + wave * 16 * MTW, n0 = (T % nslab) * BN;
        {   // accumulators start from bias (+ residual, requested a tile ago)
            f32x4 bias4[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) bias4[j] = a.bias ? *(const f32x4*)(a.bias + n0 + g * 4 * NTW + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    acc[i][j] = bias4[j];
                    if (RES && a.res) {
                        acc[i][j][0] += __builtin_bit_cast(float, rres[i][2 * j] << 16);
                        acc[i][j][1] += __builtin_bit_cast(float, rres[i][2 * j] & 0xffff0000u);
                        acc[i][j][2] += __builtin_bit_cast(float, rres[i][2 * j + 1] << 16);
                        acc[i][j][3] += __builtin_bit_cast(float, rres[i][2 * j + 1] & 0xffff0000u);
                    }
                }
        }
        ldfrag(G, 0, af[0], bfr[0]);
        for (int c = 0; c < nchunks; ++c, ++G) {
            ldfrag(G, 1, af[1], bfr[1]);
            mfmas(af[0], bfr[0]);
            c3s_spread<MTW * NTW, MTW + NTW>(std::make_integer_sequence<int, MTW + NTW>{});
            __builtin_amdgcn_sched_barrier(0);
            if (G + 1 < total) {
                __builtin_amdgcn_s_waitcnt(0xC07F);                                       // lgkmcnt(0): this wave is done reading chunk G
                asm volatile("s_barrier" ::: "memory");                                   // chunk G + 1 has landed; chunk G's buffer is free
                if (c + 1 < nchunks) ldfrag(G + 1, 0, af[0], bfr[0]);
                else if (RES && a.res) res_load(t + 1);                                         // next tile's residual: lands under the epilogue
            }
            mfmas(af[1], bfr[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: this lane holds channels n0 + 4*NTW*g + 4*j + r of pixel mw0 + i*16 + (lane & 15)
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int m = mw0 + i * 16 + (lane & 15);
            if (m < a.M) {
                uint32_t ov[2 * NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (a.relu == 2) {                                                       // Darknet's leaky ReLU (slope 0.1), epi_act1's arithmetic; round 5
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] > 0.0f ? acc[i][j][r] : 0.1f * acc[i][j][r];
                    }
                    ov[2 * j] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); ov[2 * j + 1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                    if (a.relu == 1 && n0 + g * 4 * NTW + j * 4 >= a.relu_from) {
                        ov[2 * j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j]), (s16x2){0, 0}));
                        ov[2 * j + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ov[2 * j + 1]), (s16x2){0, 0}));
                    }
                }
                c3_row_store<NTW>(a.out + (size_t)m * a.Cout + n0 + g * 4 * NTW, g, ov);
            }
        }
    }
}
// Layers the streamed implicit GEMM takes by default (measured against k_conv_igemm per HRNet layer shape at 20 crops, tools/bench_conv.py
// --fuse --tiles=-2,8): at least two K chunks (a one-chunk tile is a load -> multiply -> store chain with nothing to overlap), and a
// tile count that suits one persistent workgroup per CU: a single round, or well-filled rounds, or a single slab (no other workgroup
// shares the gathered pixels, where k_conv_igemm's two-slab tiles would win).  270 tiles = 256 + 14 is the case to avoid: two rounds for 5 %.
static bool conv_gs_auto(const ConvArgs& a) {
    const int bn = a.Cout % 48 == 0 ? 48 : 64, nslab = a.Cout / bn, tiles = ((a.M + 255) / 256) * nslab;
    if (a.Kpad < 128) return false;
#ifdef PAM_DIAG
    static const int mode = getenv("PAM_GS") ? atoi(getenv("PAM_GS")) : 1;                // tuning hook: 0 = never, 2 = always
    if (mode != 1) return mode == 2;
#endif
    const int rounds = (tiles + 255) / 256;
    return nslab == 1 || tiles <= 256 || tiles * 100 >= rounds * 256 * 78;
}
template <int NTW, bool RES, int BM = 256, int NBUF = 3>
static int launch_conv_gs_r(hipStream_t s, const ConvArgs& a) {
    constexpr size_t lds = (size_t)NBUF * (BM + 16 * NTW) * 128;
    static_assert(lds <= 160 * 1024, "LDS");
    if (!pam_max_dynamic_lds((const void*)k_conv_gs<NTW, NBUF, RES, BM>, (int)lds)) return PAM_E_HIP;
    const int ntile = ((a.M + BM - 1) / BM) * (a.Cout / (16 * NTW));
    pam_launch(k_conv_gs<NTW, NBUF, RES, BM>, dim3(ntile < 256 ? ntile : 256), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
template <int NTW>
static int launch_conv_gs(hipStream_t s, const ConvArgs& a) {
    if constexpr (NTW >= 6) return a.res ? PAM_E_ARG : launch_conv_gs_r<NTW, false>(s, a);   // the wide slab has no registers left for the residual rows
    else return a.res ? launch_conv_gs_r<NTW, true>(s, a) : launch_conv_gs_r<NTW, false>(s, a);
}

// ====================================================================================================================
// k_conv_stem: the first convolution of both networks' stems -- 3x3 / stride 1 or 2 / pad 1 from the 8-channel (RGB + zeros)
// input to 32 or 64 channels.  K per tap ROW is 3 taps x 8 channels = 24 <= 32, so one v_mfma_f32_16x16x32_bf16 covers a whole tap row:
// lane (pixel l & 15, k-group g = l >> 4) supplies as its B fragment the 16-byte input pixel (2y + ky - 1, 2x + g - 1)
// straight from global memory (g = 3 and the zero padding come from the buffer bounds check), 3 loads and 12 MFMAs per
// 16 output pixels x 64 channels.  The weights (A operand, 12 fragments) live in registers for the wave's whole row; their
// rows are permuted like k_conv3x3's so a lane ends with 16 contiguous channels: each pixel's 128 output bytes are written
// by 4 lanes x 2 x 16 B.  Pure streaming: ~35 MB in, ~71 MB out at 20 crops.  One wave per output row.
// ====================================================================================================================
struct StemArgs { const uint16_t* in; const uint16_t* wfrag; const float* bias; uint16_t* out; int N, H, W, Ho, Wo, relu; };
template <int S, int NT>                                // S = stride (1: Darknet's first layer, 2: HRNet's), NT = Cout / 16 (2 or 4)
__global__ __launch_bounds__(256) void k_conv_stem(StemArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_id = blockIdx.x * 4 + wave;
    if (row_id >= a.N * a.Ho) return;
    const int n = row_id / a.Ho, oy = row_id - n * a.Ho;
    const int px = lane & 15, g = lane >> 4;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * 16), 0x00020000);
    bf16x8 wf[NT][3];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wf[j][ky] = *(const bf16x8*)(a.wfrag + ((size_t)(j * 3 + ky) * 64 + lane) * 8);
    f32x4 bias4[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias4[j] = a.bias ? *(const f32x4*)(a.bias + g * 4 * NT + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned rowoff[3];                                  // byte offset of input row S*oy + ky - 1, or OOB
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = S * oy + ky - 1;
        rowoff[ky] = (iy >= 0 && iy < a.H && g < 3) ? (unsigned)((((size_t)n * a.H + iy) * a.W) * 16) : OOB_OFFSET;
    }
    const int ntiles = (a.Wo + 15) >> 4;
    auto load_tile = [&](int t, bf16x8* b) {
        const int ix = S * (t * 16 + px) + g - 1;
        const bool ok = ix >= 0 && ix < a.W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            b[ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (ok && rowoff[ky] != OOB_OFFSET) ? rowoff[ky] + (unsigned)ix * 16u : OOB_OFFSET, 0, 0));
    };
    bf16x8 cur[3], nxt[3];
    load_tile(0, cur);
    uint16_t* orow = a.out + (((size_t)n * a.Ho + oy) * a.Wo) * (16 * NT) + g * 4 * NT;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) load_tile(t + 1, nxt);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = bias4[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j][ky]), __builtin_bit_cast(bf16x8_t, cur[ky]), acc[j], 0, 0, 0);
        const int ox = t * 16 + px;
        if (ox < a.Wo) {
            uint32_t d[2 * NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = epi_act1(acc[j][r], a.relu & 3);
                d[2 * j] = pack_bf16x2(v[0], v[1]); d[2 * j + 1] = pack_bf16x2(v[2], v[3]);
            }
            c3_row_store<NT>(orow + (size_t)ox * (16 * NT), g, d);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) cur[ky] = nxt[ky];
    }
}

extern "C" int pam_conv2d_nhwc_bf16_ex(void* stream, const void* in, const void* w_packed, const void* w_img, const float* bias,
                                       const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                       int KH, int KW, int stride, int pad, int relu, int tile_cfg, int in_cstride, int relu_from);
extern "C" int pam_conv2d_nhwc_bf16(void* stream, const void* in, const void* w_packed, const void* w_img, const float* bias,
                                    const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad, int relu, int tile_cfg) {
    return pam_conv2d_nhwc_bf16_ex(stream, in, w_packed, w_img, bias, residual, out, N, H, W, Cin, Cout, KH, KW, stride, pad, relu,
                                   tile_cfg, Cin, 0);
}
// which kernel the last pam_conv2d_nhwc_bf16[_ex] call of this thread launched (profiling labels: bench.py's per-family roofline)
static thread_local int g_last_conv_kernel = 0;
extern "C" int pam_conv_last_kernel(void) { return g_last_conv_kernel; }
#define CONV_KIND(k) (g_last_conv_kernel = (k))
extern "C" int pam_conv2d_nhwc_bf16_ex(void* stream, const void* in, const void* w_packed, const void* w_img, const float* bias,
                                       const void* residual, void* out, int N, int H, int W, int Cin, int Cout,
                                       int KH, int KW, int stride, int pad, int relu, int tile_cfg, int in_cstride, int relu_from) {
    if (in_cstride <= 0) in_cstride = Cin;
    if (in_cstride < Cin || in_cstride % 8 != 0 || relu_from < 0 || relu_from % 16 != 0) return PAM_E_ARG;
    // tile_cfg -3 / -4: automatic like -1, but the caller STATES the layout of w_img (streamed / classic) instead of leaving it to
    // pam_conv3x3_layout() at call time -- a launch recorded under one setting of pam_conv_option and re-issued under another must not
    // read an image in the other layout
    // -5: streamed, and a 96 -> 96 layer's image is packed for slabs of 48 output channels
    // -7: a Darknet layer (activation code > 1 allowed) on the streamed kernel, image packed for 64-channel slabs (pam_conv3x3_layout_gen)
    if (tile_cfg == -7) {
        C3Args c;
        int mt = 0, pmax = 0;
        if (!in || !w_packed || !w_img || !out || N <= 0 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || in_cstride != Cin || relu_from != 0 ||
            !c3s_pick_gen(H, W, Cin, Cout, c.TH, mt, pmax) || (size_t)N * H * W * Cout * 2 >= (1ull << 31)) return PAM_E_ARG;
        c.in = (const uint16_t*)in; c.wimg = (const uint16_t*)w_img; c.bias = bias; c.res = (const uint16_t*)residual; c.out = (uint16_t*)out;
        c.N = N; c.H = H; c.W = W; c.Cout = Cout; c.relu = relu; c.inv_pw = 1.0f / (float)(W + 2);
        c.tiles_y = (H + c.TH - 1) / c.TH;
#ifdef PAM_DIAG
        c.dbg = 0; c.stamps = nullptr;
#endif
        CONV_KIND(PAM_CONV_KERNEL_3X3S);
        return launch_c3s_gen((hipStream_t)stream, c, Cin, mt);
    }
    // -8: a 192- / 384-channel ReLU / linear layer on the streamed kernel with 32-channel slabs (pam_conv3x3_layout_small; same arithmetic;
    // 16-channel slabs were measured too: no faster at 2-6 crops, slower from 9)
    if (tile_cfg == -8) {
        C3Args c;
        int mt = 0, pmax = 0, ntw = 0;
        if (!in || !w_img || !out || N <= 0 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || in_cstride != Cin || relu_from != 0 || relu > 1 ||
            !pam_conv3x3_layout_small(H, W, Cin, Cout) || !c3s_pick(H, W, Cin, Cout, c.TH, mt, pmax, ntw, 0) || (size_t)N * H * W * Cout * 2 >= (1ull << 31))
            return PAM_E_ARG;
        c.in = (const uint16_t*)in; c.wimg = (const uint16_t*)w_img; c.bias = bias; c.res = (const uint16_t*)residual; c.out = (uint16_t*)out;
        c.N = N; c.H = H; c.W = W; c.Cout = Cout; c.relu = relu; c.inv_pw = 1.0f / (float)(W + 2);
        c.tiles_y = (H + c.TH - 1) / c.TH;
#ifdef PAM_DIAG
        c.dbg = 0; c.stamps = nullptr;
#endif
        CONV_KIND(PAM_CONV_KERNEL_3X3S);
        return launch_c3s((hipStream_t)stream, c, Cin, 2, mt, pmax);
    }
    const int c96_slab = tile_cfg == -5 ? 48 : 0;
    const bool force_streamed = tile_cfg == -3 || c96_slab != 0, no_streamed = tile_cfg == -4;
    if (force_streamed || no_streamed) tile_cfg = -1;
    if (in_cstride != Cin || relu_from != 0) { if (force_streamed) return PAM_E_ARG; w_img = nullptr; }   // sliced input / partial activation: generic kernel only
    const bool stem32 = w_img && Cin == 8 && Cout == 32 && KH == 3 && KW == 3 && pad == 1 && !residual && tile_cfg < 0 && stride <= 2;
    if (!in || !w_packed || !out || N <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || (Cout % 48 != 0 && Cout % 64 != 0 && !stem32) ||
        KH < 1 || KW < 1 || KH > 3 || KW > 3 || stride < 1)
        return PAM_E_ARG;
    ConvArgs a;
    a.in = (const uint16_t*)in; a.w = (const uint16_t*)w_packed; a.bias = bias; a.res = (const uint16_t*)residual;
    a.out = (uint16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu;
    a.in_cs = in_cstride; a.relu_from = relu_from;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.Ktot = KH * KW * Cin; a.Kpad = (a.Ktot + KC - 1) / KC * KC; a.M = N * a.Ho * a.Wo;
    if (H >= 32768 || W >= 32768) return PAM_E_ARG;
    if (w_img && Cin == 8 && (Cout == 64 || Cout == 32) && KH == 3 && KW == 3 && (stride == 1 || stride == 2) && pad == 1 &&
        !residual && tile_cfg < 0) {
        StemArgs t;                                      // w_img = the pre-permuted A fragments (see pam.h)
        t.in = a.in; t.wfrag = (const uint16_t*)w_img; t.bias = bias; t.out = a.out;
        t.N = N; t.H = H; t.W = W; t.Ho = a.Ho; t.Wo = a.Wo; t.relu = relu;
        const dim3 grid((N * a.Ho + 3) / 4), blk(256);
        hipStream_t s = (hipStream_t)stream;
        CONV_KIND(PAM_CONV_KERNEL_STEM);
        if (stride == 2 && Cout == 64) pam_launch(k_conv_stem<2, 4>, grid, blk, 0, s, t);
        else if (stride == 2) pam_launch(k_conv_stem<2, 2>, grid, blk, 0, s, t);
        else if (Cout == 64) pam_launch(k_conv_stem<1, 4>, grid, blk, 0, s, t);
        else pam_launch(k_conv_stem<1, 2>, grid, blk, 0, s, t);
        return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
    }
    if (force_streamed && !(w_img && KH == 3 && KW == 3 && stride == 1 && pad == 1)) return PAM_E_ARG;
    if (w_img && KH == 3 && KW == 3 && stride == 1 && pad == 1 && tile_cfg == -1 && !no_streamed) {
        // streamed kernel (specialised loader / multiplier waves): w_img then has the layout pam_conv3x3_layout() > 0 announces.
        // Any other tile_cfg (-2 = classic kernel, >= 100 = tuning hooks) takes the classic kernel and the classic images.
        C3Args c;
        int mt = 0, pmax = 0, ntw = 0;
        const bool picked = c3s_pick(H, W, Cin, Cout, c.TH, mt, pmax, ntw, c96_slab);
        if (force_streamed && !picked) return PAM_E_ARG;
        if (picked) {
            if (relu > 1) return PAM_E_ARG;
            c.in = a.in; c.wimg = (const uint16_t*)w_img; c.bias = bias; c.res = a.res; c.out = a.out;
            c.N = N; c.H = H; c.W = W; c.Cout = Cout; c.relu = relu; c.inv_pw = 1.0f / (float)(W + 2);
            c.tiles_y = (H + c.TH - 1) / c.TH;
#ifdef PAM_DIAG
            c.dbg = g_c3_stamps ? 64 : 0; c.stamps = g_c3_stamps;
#endif
            CONV_KIND(PAM_CONV_KERNEL_3X3S);
            return launch_c3s((hipStream_t)stream, c, Cin, ntw, mt, pmax);
        }
    }
    const bool classic = tile_cfg == -2;                 // -2: the classic kernels (k_conv3x3 / k_conv_igemm), automatic tiles
    if (classic) tile_cfg = -1;
    if (w_img && KH == 3 && KW == 3 && stride == 1 && pad == 1 && (tile_cfg < 0 || tile_cfg >= 100) &&
        (Cin == 48 || Cin == 64 || Cin == 96 || Cin == 192 || Cin == 384 || Cin == 128 || Cin == 256 || Cin == 512)) {
        const int ntw = pam_conv3x3_slab(H, W, Cin, Cout) / 16;
        C3Args c;
        c.in = a.in; c.wimg = (const uint16_t*)w_img; c.bias = bias; c.res = a.res; c.out = a.out;
        c.N = N; c.H = H; c.W = W; c.Cout = Cout; c.relu = relu; c.inv_pw = 1.0f / (float)(W + 2);
        int cfg = 0;
        pick_rows(N, H, W, Cout, ntw, c.TH, cfg);
#ifdef PAM_DIAG
        c.dbg = 0; c.stamps = g_c3_stamps;
#endif
        if (tile_cfg >= 100) {                           // tuning hook: 1000 + TH*100 + cfg (diagnostic build also: 100 + dbg bits)
            if (tile_cfg >= 1000) { c.TH = (tile_cfg - 1000) / 100; cfg = (tile_cfg - 1000) % 100; }
#ifdef PAM_DIAG
            else c.dbg = tile_cfg - 100;
#endif
        }
        if (c.TH * (W + 2) > 16 * (cfg / 10) * (cfg % 10)) return PAM_E_ARG;
        c.tiles_y = (H + c.TH - 1) / c.TH;
        hipStream_t s = (hipStream_t)stream;
        // rows too wide for the patch-in-LDS kernel (e.g. the detector's 208-wide layers): the generic kernel takes them
        const int npatch = (c.TH + 2) * (W + 2), pmax = (cfg == 44 || cfg == 54) ? 416 : (cfg == 43 ? 352 : 288);
        const bool fits = npatch <= pmax && c3_lds_bytes(Cin, ntw, npatch) <= 150 * 1024 && Cout % (16 * ntw) == 0 &&
                          (relu <= 1 || c3_general_act(Cin));
        if (!fits && tile_cfg >= 100) return PAM_E_ARG;
        if (fits) CONV_KIND(PAM_CONV_KERNEL_3X3);
        if (fits) switch (Cin * 10 + ntw) {
            case 483: return launch_c3<48, 3>(s, c, cfg);
            case 643: return launch_c3<64, 3>(s, c, cfg);
            case 963: return launch_c3<96, 3>(s, c, cfg);
            case 1923: return launch_c3<192, 3>(s, c, cfg);
            case 3843: return launch_c3<384, 3>(s, c, cfg);
            case 484: return launch_c3<48, 4>(s, c, cfg);
            case 644: return launch_c3<64, 4>(s, c, cfg);
            case 964: return launch_c3<96, 4>(s, c, cfg);
            case 1924: return launch_c3<192, 4>(s, c, cfg);
            case 3844: return launch_c3<384, 4>(s, c, cfg);
            case 1922: return launch_c3<192, 2>(s, c, cfg);
            case 3842: return launch_c3<384, 2>(s, c, cfg);
            case 1921: return launch_c3<192, 1>(s, c, cfg);
            case 3841: return launch_c3<384, 1>(s, c, cfg);
            case 1284: return launch_c3<128, 4>(s, c, cfg);     // Darknet-53 widths
            case 2564: return launch_c3<256, 4>(s, c, cfg);
            case 2563: return launch_c3<256, 3>(s, c, cfg);     // HRNet transition1: 256 -> 48 at 96 x 72
            case 2562: return launch_c3<256, 2>(s, c, cfg);
            case 5124: return launch_c3<512, 4>(s, c, cfg);
            case 5122: return launch_c3<512, 2>(s, c, cfg);
            default: if (tile_cfg >= 100) return PAM_E_ARG;     // no instantiation for this (Cin, slab): generic kernel below
        }
    }
    if (tile_cfg >= 100) tile_cfg = -1;
    // streamed implicit GEMM (k_conv_gs): codes 0 / 1, taps in a 32-bit mask, whole 16-byte pieces per tap (Cin % 8 == 0)
    // (round 5: code 2 -- leaky, no residual -- too: the detector's 1x1 and strided layers, which the classic implicit GEMM ran at 20 us each)
    const bool leaky_gs = relu == 2 && !a.res && classic && relu_from == 0;
    const bool gs_ok = (relu <= 1 || leaky_gs) && KH * KW <= 9 && Cin % 8 == 0 && (size_t)N * H * W * in_cstride * 2 < (1u << 31);
    if (leaky_gs && gs_ok && tile_cfg == -1 && conv_gs_auto(a)) {
        CONV_KIND(PAM_CONV_KERNEL_GS);
        if (Cout % 48 == 0) return launch_conv_gs<3>((hipStream_t)stream, a);
        // Darknet's widths are multiples of 64: the smallest pixel tile that still is ONE round of workgroups, as for HRNet's small fuse convolutions
        const int nslab = Cout / 64;
        if (((a.M + 63) / 64) * nslab <= 256) return launch_conv_gs_r<4, false, 64, 3>((hipStream_t)stream, a);
        if (((a.M + 127) / 128) * nslab <= 256) return launch_conv_gs_r<4, false, 128, 3>((hipStream_t)stream, a);
        return launch_conv_gs<4>((hipStream_t)stream, a);
    }
    if (tile_cfg >= 8 && tile_cfg <= 12 && !gs_ok) return PAM_E_ARG;
    if (tile_cfg == 12) {                               // 64-pixel tiles: the smallest images (12 x 9) as a few hundred short workgroups
        if (a.res || Cout % 48 != 0) return PAM_E_ARG;
        CONV_KIND(PAM_CONV_KERNEL_GS);
        return launch_conv_gs_r<3, false, 64, 3>((hipStream_t)stream, a);
    }
    if (tile_cfg == 10 || tile_cfg == 11) {             // 128-pixel tiles, ring of 5 (10) / 3 (11) chunks: twice the workgroups, deeper prefetch
        if (a.res || Cout % 48 != 0) return PAM_E_ARG;
        CONV_KIND(PAM_CONV_KERNEL_GS);
        return tile_cfg == 10 ? launch_conv_gs_r<3, false, 128, 5>((hipStream_t)stream, a) : launch_conv_gs_r<3, false, 128, 3>((hipStream_t)stream, a);
    }
    if (tile_cfg == 9)                                  // streamed implicit GEMM with 96-channel slabs: the gathered pixel tile feeds twice the MFMAs
        return Cout % 96 == 0 ? (CONV_KIND(PAM_CONV_KERNEL_GS), launch_conv_gs<6>((hipStream_t)stream, a)) : PAM_E_ARG;
    // large-M strided layers with whole 96-channel slabs and no residual (merged fuse heads 48 -> 96 / 192 at 96 x 72, transition1's
    // 256 -> 96): the 96-channel-slab form gathers every pixel tile half as often (26.3 -> 23.3, 16.2 -> 13.2, 46.2 -> 31.4 us at 20 crops;
    // slower below ~100 pixel tiles, where the layer is a latency chain whatever its tile)
#ifndef PAM_GS_OLDTILES                                  /* A/B hook (tools/ab_build.sh): round-2 tile choice */
    if (gs_ok && tile_cfg == -1 && !classic && !a.res && Cout % 96 == 0 && KH == 3 && stride == 2 && a.M >= 100 * 256)
        return CONV_KIND(PAM_CONV_KERNEL_GS), launch_conv_gs<6>((hipStream_t)stream, a);
#endif
    if (gs_ok && (tile_cfg == 8 || (tile_cfg == -1 && !classic && conv_gs_auto(a)))) {
        CONV_KIND(PAM_CONV_KERNEL_GS);
#ifndef PAM_GS_OLDTILES
        if (tile_cfg == -1 && !a.res && Cout % 48 == 0) {
            // the small fuse-layer convolutions are latency chains of a few workgroups: the smallest pixel tile that still is ONE round of
            // workgroups (<= 256) -- at 20 crops 192 -> 384 at 12 x 9 21.3 -> 16.9 us, 48 -> 48 at 24 x 18 10.0 -> 6.1 us, 384 -> 336 1x1
            // 8.8 -> 5.8 us; two rounds lose (tools/bench_conv.py --fuse --tiles=-1,11,12)
            const int nslab = Cout / 48;
            if (((a.M + 63) / 64) * nslab <= 256) return launch_conv_gs_r<3, false, 64, 3>((hipStream_t)stream, a);
            if (((a.M + 127) / 128) * nslab <= 256) return launch_conv_gs_r<3, false, 128, 3>((hipStream_t)stream, a);
        }
#endif
        return (Cout % 48 == 0) ? launch_conv_gs<3>((hipStream_t)stream, a) : launch_conv_gs<4>((hipStream_t)stream, a);
    }
    CONV_KIND(PAM_CONV_KERNEL_IGEMM);
    return (Cout % 48 == 0) ? dispatch_conv<3>((hipStream_t)stream, a, tile_cfg) : dispatch_conv<4>((hipStream_t)stream, a, tile_cfg);
}

// out[n,y,x,c] = [relu](base[n,y,x,c] + sum_t term_t[n, y >> sh_t, x >> sh_t, c]); 8 channels (16 B) per thread
struct UpArgs { const uint16_t* base; const uint16_t* term[3]; int sh[3]; int tcs[3]; int nterms; uint16_t* out; int N, H, W, C, relu; };
__global__ __launch_bounds__(256) void k_upsample_add(UpArgs a) {
    const unsigned C8 = (unsigned)a.C >> 3, total = (unsigned)a.N * a.H * a.W * C8;       // host checks total < 2^31: 32-bit index math
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned pix = e / C8, c8 = e - pix * C8;
        const unsigned t2 = pix / (unsigned)a.W, x = pix - t2 * a.W;
        const unsigned n = t2 / (unsigned)a.H, y = t2 - n * a.H;
        const bf16x8 b = *(const bf16x8*)(a.base + (size_t)pix * a.C + c8 * 8);
        bf16x8 q[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {                   // all term loads issued together (independent addresses)
            q[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (t < a.nterms) {
                const unsigned hs = (unsigned)a.H >> a.sh[t], ws = (unsigned)a.W >> a.sh[t];
                q[t] = *(const bf16x8*)(a.term[t] + ((size_t)(n * hs + (y >> a.sh[t])) * ws + (x >> a.sh[t])) * a.tcs[t] + c8 * 8);
            }
        }
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32((uint16_t)b[k]);
#pragma unroll
        for (int t = 0; t < 3; ++t)                     // same summation order as before: base, then terms in order
            if (t < a.nterms) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += bf16_to_f32((uint16_t)q[t][k]);
            }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (short)f32_to_bf16_rne(a.relu ? fmaxf(v[k], 0.0f) : v[k]);
        *(bf16x8*)(a.out + (size_t)pix * a.C + c8 * 8) = o;
    }
}

extern "C" int pam_upsample_add_nhwc_bf16_ex(void* stream, const void* base, int n_terms, const void* const* terms,
                                             const int32_t* shifts, const int32_t* term_cstrides, void* out, int N, int H, int W, int C, int relu);
extern "C" int pam_upsample_add_nhwc_bf16(void* stream, const void* base, int n_terms, const void* const* terms,
                                          const int32_t* shifts, void* out, int N, int H, int W, int C, int relu) {
    return pam_upsample_add_nhwc_bf16_ex(stream, base, n_terms, terms, shifts, nullptr, out, N, H, W, C, relu);
}
extern "C" int pam_upsample_add_nhwc_bf16_ex(void* stream, const void* base, int n_terms, const void* const* terms,
                                             const int32_t* shifts, const int32_t* term_cstrides, void* out, int N, int H, int W, int C, int relu) {
    if (!base || !out || n_terms < 0 || n_terms > 3 || C % 8 != 0 || (size_t)N * H * W * (C / 8) >= (1ull << 31)) return PAM_E_ARG;
    UpArgs a;
    a.base = (const uint16_t*)base; a.out = (uint16_t*)out; a.nterms = n_terms;
    for (int t = 0; t < 3; ++t) {
        a.term[t] = t < n_terms ? (const uint16_t*)terms[t] : nullptr; a.sh[t] = t < n_terms ? shifts[t] : 0;
        a.tcs[t] = (t < n_terms && term_cstrides && term_cstrides[t] > 0) ? term_cstrides[t] : C;
        if (a.tcs[t] < C || a.tcs[t] % 8 != 0) return PAM_E_ARG;
    }
    a.N = N; a.H = H; a.W = W; a.C = C; a.relu = relu;
    const size_t total = (size_t)N * H * W * (C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    pam_launch(k_upsample_add, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
