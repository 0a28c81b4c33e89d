// libpam_hip.so, fused-block part of a1: one HRNet BasicBlock
//     out = ReLU( conv3x3( ReLU(conv3x3(x) + b1) ) + b2 + x )            (3x3, stride 1, pad 1, C -> C -> C, BN folded)
// per work item, with the intermediate kept in LDS, and the blocks of up to four branches (C = 48 / 96 / 192 at
// 96x72 / 48x36 / 24x18 for HRNet-W48) in ONE launch.  Call site this stands for: the absent HRNet backend inside
// HRNetPose.predict, /root/reference/src/ivclabpose.py:210.
//
// Work item = TH output rows (full width) of one image, all C output channels, one workgroup of NW waves per item:
//   1. X: input rows ty0-2 .. ty0+TH+1 of the zero-PADDED grid (PW = W + 2 columns) -> LDS, one slot of PA bytes per
//      grid position (PA = 32 (mod 64) bytes: conflict-free ds_read_b128 of 16 consecutive slots).
//   2. conv1 on the TH+2 rows the second conv needs.  Output "slot" p <-> window corner at X slot p, so the MFMA B fragment
//      of tap (ky,kx) is ONE ds_read_b128 at (p + ky*PW + kx)*PA -- linear in p.  Every wave keeps ALL its accumulators
//      (MW M-tiles x 3 N-tiles of 16x16) in registers and walks K = 9*C outermost; the weights stream through a ring of
//      host-packed chunk images in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass)
//      D - 1 chunks ahead of the MFMAs, ONE barrier per chunk (raw s_barrier + counted vmcnt: nothing is drained).
//   3. the residual (centre rows of X) moves to registers, then the intermediate (bias, ReLU, bf16; zero on the padding
//      columns / rows outside the image) overwrites X in place, shifted by one slot so that conv2 uses conv1's addressing.
//   4. conv2 the same way out of the intermediate; epilogue straight from the accumulators (+ residual, ReLU, 16-B stores).
// Weights are the MFMA A operand with their rows permuted inside each 48-channel slab (row j*16 + q = channel
// 12*(q >> 2) + 4*j + (q & 3)), so a lane ends with 12 contiguous channels of one pixel (as in k_conv3x3).
// Wave w: slab = w % (C/48), M part = w / (C/48); M tile i of a wave is tile  part + i * parts.
//
// Two shapes of workgroup (measured, tools/stamp_block.py + tools/micro/barrier_mfma.cpp: the MFMA blocks between two barriers are
// short, so whatever sits between them -- waits, DMA issue, address arithmetic, the barrier itself -- idles the matrix pipe unless
// another wave of the SIMD is multiplying meanwhile):
//   NW = 4 (256 threads, <= 80 KB of LDS): TWO workgroups per CU.  The SIMD partner of every wave belongs to the other
//          workgroup, which has its own barriers and its own phase -- loads, the intermediate's write pass and the epilogue of one
//          item run beside the other item's MFMAs.  Needs small tiles (C = 48: 4 rows, C = 96: 3 rows).
//   NW = 8 (512 threads, one workgroup per CU): for C = 192, whose tile does not fit twice.  The two waves of a SIMD
//          (w, w + 4) run half a phase apart (see conv_pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/pam.h"
#include "pam_launch.hpp"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
#define OOB_OFFSET 0x80000000u

// Diagnostic build only (tools/stamp_block.py compiles this file with -DPAM_DIAG): per-workgroup s_memtime stamps into a buffer
// of their own that no kernel reads; the shipped library contains no stamp code.
#ifdef PAM_DIAG
static unsigned long long* g_bb_stamps = nullptr;
extern "C" int pam_block_debug_stamps(void* dev_buf) { g_bb_stamps = (unsigned long long*)dev_buf; return PAM_OK; }
#define BB_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_STAMP(k) do { } while (0)
#endif
#define BB_WAITVM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// barrier between conv1 and conv2 of an item.  __syncthreads() also waits vmcnt(0), i.e. drains the weight DMAs of conv2's first chunks;
// a raw s_barrier behind lgkmcnt(0) keeps them in flight -- measured identical (21.9 vs 21.6 us per C = 48 block, 2.55-2.60 ms per forward
// either way; tools/ab_build.sh PAM_BB_RAWBARRIER), so the plain form stays
#ifdef PAM_BB_RAWBARRIER
#define BB_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("s_barrier" ::: "memory"); } while (0)
#else
#define BB_BARRIER() __syncthreads()
#endif
#define BB_LGKM0 0xC07F                                  // s_waitcnt immediate: lgkmcnt(0), vmcnt / expcnt untouched

namespace {

struct BBranch {
    const uint16_t* in; const uint16_t* wimg; const float* bias; uint16_t* out;
    int N, H, W, C, TH, tiles_y, item0, nitems, sh;
    float inv_pw;
#ifdef PAM_DIAG
    unsigned long long* stamps;
#endif
};
struct BBArgs { BBranch br[PAM_BLOCK_MAX_BRANCHES]; int nbr; };

// per-width constants (host and device).  NW = waves per workgroup.
// SH ("short" items, NW = 8, C = 48 / 96 only): 4-row tiles with their own M-tile counts -- twice the items at roughly half the cost each,
// for grouped launches whose items would otherwise pack badly on 256 CUs (360 items of 22-29 us = two rounds for 104 of them)
template <int C, int NW, bool SH = false> struct BCfg {
    static constexpr int NSLAB = C / 48, MPARTS = NW / NSLAB;
    static constexpr int PA = (C == 48) ? 96 : 2 * C + 32;                 // bytes per X slot
    static constexpr int KSTEPS = (9 * C + 31) / 32;                       // 32-deep k-steps that carry weights: K = flattened (tap, cin), zero tail
    static constexpr int NSTEP = (KSTEPS + 1) / 2 * 2;                     // ... padded to an even count (C = 96: one all-zero k-step)
    static constexpr int KS = (NW == 4 && C == 96) ? 1 : 2;                // k-steps per weight chunk = per barrier
    static constexpr int NCH = NSTEP / KS;                                  // chunks per convolution
    static constexpr int D = (NW == 4 || C == 192) ? 3 : 4;                // ring slots: chunk c + D - 1 is issued while chunk c is multiplied
    static constexpr int PWT = 64;                                          // bytes per weight row of a k-step image: 4 pieces of 16 B, no padding --
                                                                            // row q of a 16-row tile keeps piece g at g ^ sigma[q >> 2], sigma = (0,2,3,1):
                                                                            // every 16-lane group of a ds_read_b128 then covers all 64 banks once
    static constexpr int SUB = C * PWT;                                     // bytes of one k-step's image [C rows][64 B]
    static constexpr int CHB = KS * SUB;                                    // bytes per chunk (whole 1-KiB DMA pieces)
    static_assert(CHB % 1024 == 0, "chunk = whole DMA pieces");
    // M tiles per wave, conv1 / conv2
    static constexpr int MW1 = SH ? 4 : (NW == 8 ? (C == 192 ? 5 : 6) : (C == 48 ? 7 : 6));
    static constexpr int MW2 = SH ? 3 : (NW == 8 ? (C == 192 ? 4 : 5) : (C == 48 ? 5 : 4));
    static constexpr int PWMAX = (C == 48) ? 76 : (C == 96 ? 40 : 24);      // widest padded row this width is instantiated for
    static constexpr int XMAX = 16 * MW1 * MPARTS + 2 * PWMAX;              // bound on X slots (patch pieces per thread)
    static constexpr bool SUPPORTED = !(NW == 4 && C == 192) && !(SH && (NW != 8 || C == 192));   // the 192-wide tile does not fit twice per CU
};

__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }   // exact for x < 2^16
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {           // bf16 is sign-magnitude: max(int16, 0) clears the negatives
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}

template <int C, int NW, bool SH = false>
__device__ __forceinline__ void bblock_item(const BBranch& a, int v, char* smem) {
    typedef BCfg<C, NW, SH> K;
    constexpr int NSLAB = K::NSLAB, MPARTS = K::MPARTS, PA = K::PA, NCH = K::NCH, PWT = K::PWT, CHB = K::CHB, KS = K::KS;
    constexpr int MW1 = K::MW1, MW2 = K::MW2, PC8 = C / 8, D = K::D, T = 64 * NW;
    constexpr int NPX = (K::XMAX * PC8 + T - 1) / T;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform (scalar) from here on
    const int slab = wave % NSLAB, mpart = wave / NSLAB;
    // XCD-aware item order: workgroups b, b+8, ... share an XCD and its L2 -> give each XCD a contiguous run of tiles
    // (vertically adjacent tiles re-read each other's halo rows)
    const int ntiles = a.nitems;
    int bx;
    { const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, loc = v >> 3; bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    const int n = bx / a.tiles_y, ty0 = (bx - n * a.tiles_y) * a.TH;
    const int th = min(a.TH, a.H - ty0);
    const int PW = a.W + 2, XS = (a.TH + 4) * PW;
    const int n1 = (th + 2) * PW, n2 = th * PW;

    char* Xb = smem;
    char* zero_slot = smem + ((((size_t)XS + 2) * PA + 15) & ~(size_t)15);               // 64 zero bytes: K tail / padding k-step
    char* Wb0 = zero_slot + 64;
    const char* wsrc = (const char*)a.wimg;
    auto issue_chunk = [&](int c) {                     // chunk c of the 2*NCH stream -> ring slot c % D (LDS-DMA, 1 KiB per wave-instruction)
        char* dst = Wb0 + (size_t)((unsigned)c % (unsigned)D) * CHB;
        const char* src = wsrc + (size_t)c * CHB + lane * 16;
#pragma unroll
        for (int piece = wave; piece < CHB / 1024; piece += NW)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + piece * 1024), (lds_void*)(dst + piece * 1024), 16, 0, 0);
    };
    BB_STAMP(0);
#pragma unroll
    for (int c = 0; c < D - 2; ++c) issue_chunk(c);
    if (tid < 4) *(u32x4*)(zero_slot + tid * 16) = (u32x4){0, 0, 0, 0};

    // ---- X patch: global -> registers -> LDS (zero fill outside the image by the descriptor's range check) ----------------
    {
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * C * 2), 0x00020000);
        u32x4 ra[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int q = tid + i * T;
            unsigned off = OOB_OFFSET;
            if (q < XS * PC8) {
                const int pp = q / PC8, c8 = q - pp * PC8;
                const int row = fdiv_small(pp, a.inv_pw), col = pp - row * PW;
                const int iy = ty0 - 2 + row, ix = col - 1;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    off = (unsigned)((((size_t)n * a.H + iy) * a.W + ix) * C * 2 + c8 * 16);
            }
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int q = tid + i * T;
            if (q < XS * PC8) { const int pp = q / PC8, c8 = q - pp * PC8; *(u32x4*)(Xb + (size_t)pp * PA + c8 * 16) = ra[i]; }
        }
    }

    BB_STAMP(1);
    const float* bias = a.bias + 48 * slab + 12 * g;
    const char* xl = Xb + (size_t)(mpart * 16 + l15) * PA;               // this lane's slot in M tile 0 of the wave
    const char* wl = Wb0 + (size_t)(slab * 48 + l15) * PWT + ((g ^ ((0x78 >> ((l15 >> 2) * 2)) & 3)) * 16);      // 0x78 = sigma packed, 2 bits each

    // Weight ring protocol.  top(c), run by every wave before the first k-step of chunk c: wait until this wave's pieces of chunk
    // c + 1 have landed (counted vmcnt: the D - 3 younger chunks stay in flight), barrier (-> chunk c + 1 is visible to all, and
    // everyone is done reading chunk c - 1), then refill that slot with chunk c + D - 1.  A raw s_barrier: fragment reads already in
    // flight may cross it, nothing is drained.
    constexpr int TOT = 2 * NCH, PCS = CHB / 1024, P0 = PCS / NW, REM = PCS % NW;
    auto top_wait_barrier = [&](int c) {
        if (c + 1 < TOT) {
            if (D > 3 && c + D - 2 < TOT) { if (REM != 0 && wave < REM) BB_WAITVM((D - 3) * (P0 + 1)); else BB_WAITVM((D - 3) * P0); }
            else BB_WAITVM(0);
        }
        asm volatile("s_barrier" ::: "memory");
    };
    auto top_issue = [&](int c) { if (c + D - 1 < TOT) issue_chunk(c + D - 1); };
    auto top = [&](int c) { top_wait_barrier(c); top_issue(c); };
    __syncthreads();                                    // X (and the zero slot) visible
    top(-1);

    // one convolution: K outermost, all accumulators of the wave live in registers; the fragments of k-step st + 1 are read
    // while the MFMAs of k-step st issue (two register sets)
    auto conv_pass = [&](auto MWC, int cv, auto& acc) {
        constexpr int MW = decltype(MWC)::value;
        const int c0 = cv * NCH;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 b4 = *(const f32x4*)(bias + cv * C + 4 * j);
#pragma unroll
            for (int i = 0; i < MW; ++i) acc[i][j] = b4;
        }
        auto ldfrag = [&](int st, bf16x8* af, bf16x8* bf) {
            const int cc = st / KS, ks = st - cc * KS;
            const char* wc = wl + (size_t)((unsigned)(c0 + cc) % (unsigned)D) * CHB + ks * K::SUB;
            unsigned aoff; bool zero = false;
            if constexpr (C == 48) {
                const int k0 = 32 * st + 8 * g;                          // flattened (tap, c); an 8-slice never straddles taps
                const int t = k0 / 48, ch = k0 - t * 48;
                const int ky = t / 3, kx = t - ky * 3;
                zero = k0 >= 432;
                aoff = (unsigned)((ky * PW + kx) * PA + ch * 2);
            } else {
                constexpr int SPT = C / 32;                              // k-steps per tap
                zero = st >= K::KSTEPS;                                  // the padding k-step: zero weights, and zero (not stale) activations
                const int t = st / SPT, kc = st - t * SPT;
                const int ky = t / 3, kx = t - ky * 3;
                aoff = (unsigned)((ky * PW + kx) * PA + kc * 64 + g * 16);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) af[j] = *(const bf16x8*)(wc + (size_t)j * 16 * PWT);
#pragma unroll
            for (int i = 0; i < MW; ++i) bf[i] = *(const bf16x8*)(zero ? zero_slot : xl + (size_t)i * MPARTS * 16 * PA + aoff);
        };
        auto mfmas = [&](const bf16x8* af, const bf16x8* bf) {
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[j]), __builtin_bit_cast(bf16x8_t, bf[i]), acc[i][j], 0, 0, 0);
        };
        bf16x8 a0[3], b0[MW], a1[3], b1[MW];
        ldfrag(0, a0, b0);
        int st = 0;
        // One loop iteration = two k-steps.  NW = 8: the two waves of a SIMD (w and w + 4) run half a phase apart so that one's
        // issue / read block runs beside the other's MFMAs instead of both idling the matrix pipe after every barrier:
        //   waves 4-7: [barrier, DMA issue, reads of st + 1 | MFMAs of st | reads of st + 2 | MFMAs of st + 1]
        //   waves 0-3: [barrier | MFMAs of st - 1 | DMA issue, reads of st + 1 | MFMAs of st | reads of st + 2]   (one k-step behind)
        // The older half (waves 0-3) wins the issue arbitration, so it is the one that multiplies first.  Only MFMAs (registers) lag;
        // every LDS access keeps its place relative to the barriers, so the ring protocol is the same for both halves.
        // NW = 4: all waves take the first form; the SIMD partner is a wave of the CU's other workgroup.
        if (NW == 4 || wave >= 4) {
#pragma unroll 1
            for (; st < K::NSTEP; st += 2) {
                // the fragments of this k-step were read one k-step ago: tell the compiler's wait-count pass they are here, so it does
                // not wait (lgkmcnt(0)) AFTER the next k-step's reads have been issued -- those stay in flight under the MFMAs
                __builtin_amdgcn_s_waitcnt(BB_LGKM0);
                top(c0 + st / KS);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(st + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(BB_LGKM0);
                if constexpr (KS == 1) top(c0 + st + 1);
                if (st + 2 < K::NSTEP) ldfrag(st + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (NW == 8) {
            static_assert(NW != 8 || KS == 2, "the lagging half assumes one barrier per two k-steps");
            // counted waits: the set about to be multiplied was read a whole k-step ago; the newest reads (3 + MW ds_read_b128 of the
            // other set, issued just before) stay in flight across the barrier
            constexpr int LGKM_KEEP = BB_LGKM0 | ((3 + MW) << 8);
            __builtin_amdgcn_s_waitcnt(BB_LGKM0);
            top(c0);                                                     // chunk 0's barrier: nothing to multiply yet
            ldfrag(1, a1, b1);
            st = 1;
#pragma unroll 1
            for (; st + 1 < K::NSTEP; st += 2) {                         // st odd: set 0 holds k-step st - 1, set 1 holds st
                __builtin_amdgcn_s_waitcnt(LGKM_KEEP);
                top_wait_barrier(c0 + ((st + 1) >> 1));
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0);                                           // k-step st - 1
                __builtin_amdgcn_sched_barrier(0);
                top_issue(c0 + ((st + 1) >> 1));
                ldfrag(st + 1, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(LGKM_KEEP);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a1, b1);                                           // k-step st
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < K::NSTEP) ldfrag(st + 2, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // st = NSTEP - 1: set 0 holds k-step NSTEP - 2, set 1 holds NSTEP - 1
            __builtin_amdgcn_s_waitcnt(LGKM_KEEP);
            mfmas(a0, b0);
            mfmas(a1, b1);
        }
    };

    // ---- conv1 -------------------------------------------------------------------------------------------------------------
    uint32_t rres[MW2][6];
    {
        f32x4 acc[MW1][3];
        conv_pass(std::integral_constant<int, MW1>{}, 0, acc);
        BB_STAMP(2);
        // residual of this wave's conv2 tiles: centre slot o + 2*PW + 1, channels 48*slab + 12*g .. + 11 (24 bytes)
#pragma unroll
        for (int i = 0; i < MW2; ++i) {
            const int o = (mpart + MPARTS * i) * 16 + l15;
            const char* p = Xb + (size_t)(o + 2 * PW + 1) * PA + (48 * slab + 12 * g) * 2;
#pragma unroll
            for (int h = 0; h < 3; ++h) { const u32x2 q2 = *(const u32x2*)(p + 8 * h); rres[i][2 * h] = q2[0]; rres[i][2 * h + 1] = q2[1]; }
        }
        BB_BARRIER();                                                    // every wave is done reading X
        // intermediate = ReLU(conv1 + b1) as bf16 at slot p + 1; zero where the grid position is padding / outside the image
#pragma unroll
        for (int i = 0; i < MW1; ++i) {
            const int p = (mpart + MPARTS * i) * 16 + l15;
            const int row = fdiv_small(p, a.inv_pw), col = p - row * PW;
            const int iy = ty0 - 1 + row;
            const bool ok = p < n1 && col < a.W && (unsigned)iy < (unsigned)a.H;
            if (p > n1) continue;                                        // junk tiles past the last row: nothing reads them, and they may lie outside X
            char* d = Xb + (size_t)(p + 1) * PA + (48 * slab + 12 * g) * 2;
            uint32_t ov[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ov[2 * j] = ok ? relu_bf16x2(pack_bf16x2(acc[i][j][0], acc[i][j][1])) : 0u;
                ov[2 * j + 1] = ok ? relu_bf16x2(pack_bf16x2(acc[i][j][2], acc[i][j][3])) : 0u;
            }
            // 24 bytes at a 24-byte step: one 16-byte + one 8-byte write (16 + 8 for even g, 8 + 16 for odd g, as the epilogue's stores)
            // instead of three 8-byte writes -- at a slot pitch of 96 B a ds_write_b64 of 16 consecutive slots is 4-way bank-conflicted
            // (24.5 % of the kernel's LDS cycles were conflict cycles), the 8-lane groups of a ds_write_b128 only 2-way
#ifdef PAM_BB_WRITE64                                                         // A/B hook (tools/ab_build.sh): the old three 8-byte writes
#pragma unroll
            for (int j = 0; j < 3; ++j) *(u32x2*)(d + 8 * j) = (u32x2){ov[2 * j], ov[2 * j + 1]};
#else
            const bool odd = g & 1;
            *(u32x4*)(d + (odd ? 8 : 0)) = odd ? (u32x4){ov[2], ov[3], ov[4], ov[5]} : (u32x4){ov[0], ov[1], ov[2], ov[3]};
            *(u32x2*)(d + (odd ? 0 : 16)) = odd ? (u32x2){ov[0], ov[1]} : (u32x2){ov[4], ov[5]};
#endif
        }
        if (tid < PA / 16) *(u32x4*)(Xb + tid * 16) = (u32x4){0, 0, 0, 0};      // slot 0 = the padding column left of the first row
    }
    BB_BARRIER();                                       // the intermediate is visible
    BB_STAMP(3);

    // ---- conv2 + epilogue ----------------------------------------------------------------------------------------------------
    {
        f32x4 acc[MW2][3];
        conv_pass(std::integral_constant<int, MW2>{}, 1, acc);
        BB_STAMP(4);
#pragma unroll
        for (int i = 0; i < MW2; ++i) {
            const int o = (mpart + MPARTS * i) * 16 + l15;
            const int row = fdiv_small(o, a.inv_pw), col = o - row * PW;
            if (o < n2 && col < a.W) {
                uint32_t ov[6];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float r0 = __builtin_bit_cast(float, rres[i][2 * j] << 16), r1 = __builtin_bit_cast(float, rres[i][2 * j] & 0xffff0000u);
                    const float r2 = __builtin_bit_cast(float, rres[i][2 * j + 1] << 16), r3 = __builtin_bit_cast(float, rres[i][2 * j + 1] & 0xffff0000u);
                    ov[2 * j] = relu_bf16x2(pack_bf16x2(acc[i][j][0] + r0, acc[i][j][1] + r1));
                    ov[2 * j + 1] = relu_bf16x2(pack_bf16x2(acc[i][j][2] + r2, acc[i][j][3] + r3));
                }
                uint16_t* d = a.out + (((size_t)n * a.H + ty0 + row) * a.W + col) * C + 48 * slab + 12 * g;
                const bool odd = g & 1;                                  // 24 bytes, 8-byte aligned: 16 + 8 (even g) or 8 + 16 (odd g)
                *(u32x4*)(d + (odd ? 4 : 0)) = odd ? (u32x4){ov[2], ov[3], ov[4], ov[5]} : (u32x4){ov[0], ov[1], ov[2], ov[3]};
                *(u32x2*)(d + (odd ? 0 : 8)) = odd ? (u32x2){ov[0], ov[1]} : (u32x2){ov[4], ov[5]};
            }
        }
    }
    BB_STAMP(5);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_bblock(BBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int b = 0;
    const int item = blockIdx.x;
#pragma unroll
    for (int q = 1; q < PAM_BLOCK_MAX_BRANCHES; ++q) if (q < a.nbr && item >= a.br[q].item0) b = q;
    const BBranch& br = a.br[b];
    switch (br.C) {
        case 48: if constexpr (NW == 8) { if (br.sh) { bblock_item<48, NW, true>(br, item - br.item0, smem); break; } }
                 bblock_item<48, NW>(br, item - br.item0, smem); break;
        case 96: if constexpr (NW == 8) { if (br.sh) { bblock_item<96, NW, true>(br, item - br.item0, smem); break; } }
                 bblock_item<96, NW>(br, item - br.item0, smem); break;
        case 192: if constexpr (BCfg<192, NW>::SUPPORTED) bblock_item<192, NW>(br, item - br.item0, smem); break;
    }
}

template <int C, int NW, bool SH = false> static size_t lds_for(int TH, int W) {
    typedef BCfg<C, NW, SH> K;
    const int PW = W + 2, XS = (TH + 4) * PW;
    size_t x = ((size_t)(XS + 2) * K::PA + 15) & ~(size_t)15;
    const size_t reach = (size_t)(16 * K::MW1 * K::MPARTS + 2 * PW + 2 + 1) * K::PA;    // junk M tiles read past the patch: keep it inside the allocation
    size_t total = x + 64 + (size_t)K::D * K::CHB;
    if (reach > total) total = reach;
    return total;
}
template <int C, int NW, bool SH = false> static int rows_for(int H, int W) {
    typedef BCfg<C, NW, SH> K;
    if (!K::SUPPORTED || H < 1 || W < 1 || W + 2 > K::PWMAX) return 0;
    const int PW = W + 2;
    const size_t cap = NW == 4 ? 80 * 1024 : 160 * 1024;               // NW = 4 is only worth it with two workgroups per CU
    // tallest tile whose conv1 / conv2 slots fit the wave tiling and whose LDS fits, preferring a divisor of H near the cap
    int th = 16 * K::MW1 * K::MPARTS / PW - 2;
    const int th2 = 16 * K::MW2 * K::MPARTS / PW;
    if (th2 < th) th = th2;
    if (th > H) th = H;
    while (th >= 1 && lds_for<C, NW, SH>(th, W) > cap) --th;
    if (th < 1) return 0;
    for (int t = th; t >= 1 && t * 100 >= th * 75; --t) if (H % t == 0) return t;
    return th;
}
static int rows_dispatch(int C, int H, int W, int nw, int sh = 0) {
    if (sh) return nw != 8 ? 0 : (C == 48 ? rows_for<48, 8, true>(H, W) : (C == 96 ? rows_for<96, 8, true>(H, W) : 0));
    switch (C * 10 + nw) {
        case 484: return rows_for<48, 4>(H, W);
        case 964: return rows_for<96, 4>(H, W);
        case 488: return rows_for<48, 8>(H, W);
        case 968: return rows_for<96, 8>(H, W);
        case 1928: return rows_for<192, 8>(H, W);
    }
    return 0;
}
static size_t lds_dispatch(int C, int th, int W, int nw, int sh = 0) {
    if (sh) return nw != 8 ? ~(size_t)0 : (C == 48 ? lds_for<48, 8, true>(th, W) : (C == 96 ? lds_for<96, 8, true>(th, W) : ~(size_t)0));
    switch (C * 10 + nw) {
        case 484: return lds_for<48, 4>(th, W);
        case 964: return lds_for<96, 4>(th, W);
        case 488: return lds_for<48, 8>(th, W);
        case 968: return lds_for<96, 8>(th, W);
        case 1928: return lds_for<192, 8>(th, W);
    }
    return ~(size_t)0;
}

}  // namespace

extern "C" int pam_basic_block_rows(int C, int H, int W, int waves) {
    if (waves == 4 || waves == 8) return rows_dispatch(C, H, W, waves);
    const int r8 = rows_dispatch(C, H, W, 8);
    return r8 > 0 ? r8 : rows_dispatch(C, H, W, 4);
}

extern "C" int pam_basic_block_chunk_layout(int C, int32_t* out5) {
    if (!out5) return PAM_E_ARG;
    // the global image is one [C][64 B] sub-image per k-step, conv1's NSTEP then conv2's: independent of the workgroup shape
    switch (C) {
        case 48: out5[1] = BCfg<48, 8>::NSTEP; break;
        case 96: out5[1] = BCfg<96, 8>::NSTEP; break;
        case 192: out5[1] = BCfg<192, 8>::NSTEP; break;
        default: return PAM_E_ARG;
    }
    out5[0] = 1; out5[2] = 64; out5[3] = C * 64; out5[4] = out5[1];
    return PAM_OK;
}

extern "C" int pam_basic_block_nhwc_bf16_ex(void* stream, int n_branches, const PamBlockDesc* d, int waves) {
    if (n_branches < 1 || n_branches > PAM_BLOCK_MAX_BRANCHES || !d) return PAM_E_ARG;
    const int short_mask = (waves >> 4) & 0xf;          // bit k: branch k of the call runs as "short" 4-row items (8-wave form, C = 48 / 96)
    waves &= 0xf;
    if (short_mask && waves != 8) return PAM_E_ARG;
    if (waves != 4 && waves != 8) {                                    // auto: two workgroups per CU whenever every branch fits that way
        waves = 4;
        for (int k = 0; k < n_branches; ++k) if (rows_dispatch(d[k].C, d[k].H, d[k].W, 4) < 1) waves = 8;
    }
    BBArgs a;
    a.nbr = n_branches;
    // widest first: the long items are dispatched first and the short ones fill in behind them
    int order[PAM_BLOCK_MAX_BRANCHES];
    for (int i = 0; i < n_branches; ++i) order[i] = i;
    auto cost = [&](int k) { return d[k].C * (((short_mask >> k) & 1) ? 1 : 2); };      // per-item cost rank: width x tile height class
    for (int i = 0; i < n_branches; ++i)
        for (int j = i + 1; j < n_branches; ++j)
            if (cost(order[j]) > cost(order[i])) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int items = 0;
    size_t lds = 0;
    for (int k = 0; k < n_branches; ++k) {
        const PamBlockDesc& s = d[order[k]];
        if (!s.in || !s.w_img || !s.bias || !s.out || s.N < 1 || s.in == s.out) return PAM_E_ARG;
        const int sh = (short_mask >> order[k]) & 1;
        const int th = rows_dispatch(s.C, s.H, s.W, waves, sh);
        if (th < 1) return PAM_E_ARG;
        BBranch& b = a.br[k];
        b.sh = sh;
        b.in = (const uint16_t*)s.in; b.wimg = (const uint16_t*)s.w_img; b.bias = s.bias; b.out = (uint16_t*)s.out;
        b.N = s.N; b.H = s.H; b.W = s.W; b.C = s.C; b.TH = th; b.tiles_y = (s.H + th - 1) / th;
        b.item0 = items; b.nitems = b.tiles_y * s.N; b.inv_pw = 1.0f / (float)(s.W + 2);
#ifdef PAM_DIAG
        b.stamps = g_bb_stamps;
#endif
        if ((size_t)s.N * s.H * s.W * s.C * 2 >= (1ull << 31)) return PAM_E_ARG;
        items += b.nitems;
        const size_t l = lds_dispatch(s.C, th, s.W, waves, sh);
        if (l > lds) lds = l;
    }
    for (int k = n_branches; k < PAM_BLOCK_MAX_BRANCHES; ++k) a.br[k] = a.br[0];
    if (lds > 160 * 1024) return PAM_E_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_bblock<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_bblock<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (waves == 4) pam_launch(k_bblock<4>, dim3(items), dim3(256), lds, (hipStream_t)stream, a);
    else pam_launch(k_bblock<8>, dim3(items), dim3(512), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_basic_block_nhwc_bf16(void* stream, int n_branches, const PamBlockDesc* d) {
    return pam_basic_block_nhwc_bf16_ex(stream, n_branches, d, 0);
}
