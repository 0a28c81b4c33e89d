// libpam_hip.so, image part of a1 (HRNetPose.predict pre/post-processing; call site /root/reference/src/ivclabpose.py:210).
// Crop / resize / normalise in front of the conv stack (csrc/pam_conv.hip, csrc/pam_block.hip), the network's final 1x1
// convolution and the arg-max decode behind it.  All HBM-bound streaming kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"

#define J PAM_J

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}

// One thread per output pixel: bilinear sample (half-pixel centres, border replicate) of the person box from the BGR
// uint8 frame, BGR->RGB, /255, ImageNet mean/std, bf16 NHWC store (6 B per thread, contiguous across the wave).
// antialias: the resize of upstream simple-HRNet (a PIL image through torchvision's Resize) filters with a triangle whose support grows
// with the down-scaling factor; plain bilinear interpolation (the default here, = cv2.resize INTER_LINEAR / F.interpolate(bilinear)) only
// ever reads 2 x 2 pixels.  The two agree for boxes no larger than the network input; HD boxes taller than 384 pixels are down-scaled.
// With antialias the taps are the frame pixels i whose centres lie within max(scale, 1) of the output pixel's centre, inside the box
// rounded outwards (the crop upstream cuts), weights 1 - |i + 0.5 - centre| / max(scale, 1), normalised (PIL's ImagingResample formula,
// in float32: PIL's own uint8 rounding between its two passes is not reproduced).  Crops n_src .. (grid) repeat crop n_src - 1: a replay
// bucket larger than the call (HRNetPose pads a batch to a multiple of graph_bucket) needs no padded box table.
constexpr int AA_MAXT = 24;             // taps per axis at most (down-scaling by up to 11.5)
__global__ __launch_bounds__(256) void k_preprocess_crops(int n_src, const uint8_t* const* __restrict__ frames, int H, int W,
                                                          const int* __restrict__ view_of, const float* __restrict__ boxes,
                                                          int oh, int ow, int oc, uint16_t* __restrict__ out, int antialias) {
    const int crop = blockIdx.y, src = min(crop, n_src - 1);
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= oh * ow) return;
    const int oy = px / ow, ox = px % ow;
    const uint8_t* __restrict__ img = frames[view_of[src]];
    const float bx = boxes[src * 4 + 0], by = boxes[src * 4 + 1], bw = boxes[src * 4 + 2], bh = boxes[src * 4 + 3];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};
    uint16_t o[3];
    if (!antialias) {
        float sx = bx + (ox + 0.5f) * (bw / (float)ow) - 0.5f;
        float sy = by + (oy + 0.5f) * (bh / (float)oh) - 0.5f;
        sx = fminf(fmaxf(sx, 0.0f), (float)(W - 1));
        sy = fminf(fmaxf(sy, 0.0f), (float)(H - 1));
        const int x0 = (int)sx, y0 = (int)sy;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float fx = sx - (float)x0, fy = sy - (float)y0;
        const uint8_t* r0 = img + ((size_t)y0 * W) * 3;
        const uint8_t* r1 = img + ((size_t)y1 * W) * 3;
        // the two pixels of a row are 6 consecutive bytes: ONE unaligned 8-byte load per row instead of six byte loads (the kernel was bound by
        // the texture-address unit: 12 one-byte gathers per output pixel); not for the last two columns (x1 is clamped there / the load would
        // leave the row)
        uint8_t t0[6], t1[6];
        if (x0 + 2 < W) {
            struct __attribute__((packed)) U8 { uint32_t x, y; };              // alignment 1: the backend emits one dwordx2 load (unaligned access is on)
            const U8 q0 = *(const U8*)(r0 + x0 * 3), q1 = *(const U8*)(r1 + x0 * 3);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                t0[k] = (uint8_t)((k < 4 ? q0.x >> (8 * k) : q0.y >> (8 * (k - 4))) & 0xff);
                t1[k] = (uint8_t)((k < 4 ? q1.x >> (8 * k) : q1.y >> (8 * (k - 4))) & 0xff);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { t0[k] = r0[x0 * 3 + k]; t0[3 + k] = r0[x1 * 3 + k]; t1[k] = r1[x0 * 3 + k]; t1[3 + k] = r1[x1 * 3 + k]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {          // c indexes RGB; source is BGR
            const int sc = 2 - c;
            const float a = (float)t0[sc], b = (float)t0[3 + sc];
            const float cc = (float)t1[sc], d = (float)t1[3 + sc];
            const float top = a + (b - a) * fx, bot = cc + (d - cc) * fx;
            const float v = (top + (bot - top) * fy) * (1.0f / 255.0f);
            o[c] = f32_to_bf16((v - mean[c]) * istd[c]);
        }
    } else {
        const float scx = bw / (float)ow, scy = bh / (float)oh, supx = fmaxf(scx, 1.0f), supy = fmaxf(scy, 1.0f);
        const float cx = bx + (ox + 0.5f) * scx, cy = by + (oy + 0.5f) * scy;
        const int xlo = max(0, (int)floorf(bx)), xhi = max(xlo + 1, min(W, (int)ceilf(bx + bw)));
        const int ylo = max(0, (int)floorf(by)), yhi = max(ylo + 1, min(H, (int)ceilf(by + bh)));
        int x0 = max(xlo, (int)(cx - supx + 0.5f)), x1 = min(xhi, (int)(cx + supx + 0.5f));
        int y0 = max(ylo, (int)(cy - supy + 0.5f)), y1 = min(yhi, (int)(cy + supy + 0.5f));
        if (x1 <= x0) { x0 = min(max((int)cx, xlo), xhi - 1); x1 = x0 + 1; }
        if (y1 <= y0) { y0 = min(max((int)cy, ylo), yhi - 1); y1 = y0 + 1; }
        x1 = min(x1, x0 + AA_MAXT); y1 = min(y1, y0 + AA_MAXT);
        const float isx = 1.0f / supx, isy = 1.0f / supy;
        float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f;
        for (int y = y0; y < y1; ++y) {
            const float wy = fmaxf(0.0f, 1.0f - fabsf(((float)y + 0.5f - cy) * isy));
            const uint8_t* r = img + ((size_t)y * W) * 3;
            float row[3] = {0.f, 0.f, 0.f}, wr = 0.f;
            for (int x = x0; x < x1; ++x) {
                const float wx = fmaxf(0.0f, 1.0f - fabsf(((float)x + 0.5f - cx) * isx));
                row[0] += wx * (float)r[x * 3 + 2]; row[1] += wx * (float)r[x * 3 + 1]; row[2] += wx * (float)r[x * 3 + 0];
                wr += wx;
            }
            acc[0] += wy * row[0]; acc[1] += wy * row[1]; acc[2] += wy * row[2];
            wsum += wy * wr;
        }
        const float inv = wsum > 0.f ? 1.0f / (wsum * 255.0f) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = f32_to_bf16((acc[c] * inv - mean[c]) * istd[c]);
    }
    if (oc == 3) {
        uint16_t* dst = out + (((size_t)crop * oh + oy) * ow + ox) * 3;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    } else {            // 8-channel form for the MFMA conv kernel (Cin % 8 == 0): RGB + 5 zero channels, one 16-B store
        uint4 v;
        v.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); v.y = (uint32_t)o[2]; v.z = 0; v.w = 0;
        *(uint4*)(out + (((size_t)crop * oh + oy) * ow + ox) * 8) = v;
    }
}

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {    // larger value wins; ties -> smaller flat index (np.argmax)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ Best wave_argmax(Best x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Best o; o.v = __shfl_xor(x.v, off, 64); o.i = __shfl_xor(x.i, off, 64);
        x = better(x, o);
    }
    return x;
}
__device__ __forceinline__ void write_keypoint_at(int j, double py, double pxx, float score, int hm_h, int hm_w, const float* box,
                                                  double* det_row, float* kp_row) {
    // upstream SimpleHRNet.predict: pt / (res // 4) * box extent + box origin, stored float32
    const float y = (float)(py / (double)hm_h * (double)box[3] + (double)box[1]);
    const float x = (float)(pxx / (double)hm_w * (double)box[2] + (double)box[0]);
    det_row[j * 3 + 0] = (double)y; det_row[j * 3 + 1] = (double)x; det_row[j * 3 + 2] = (double)score;
    if (kp_row) { kp_row[j * 3 + 0] = x; kp_row[j * 3 + 1] = y; kp_row[j * 3 + 2] = score; }
}
__device__ __forceinline__ void write_keypoint(int j, Best b, int hm_h, int hm_w, const float* box, double* det_row, float* kp_row) {
    write_keypoint_at(j, (double)(b.i / hm_w), (double)(b.i % hm_w), b.v, hm_h, hm_w, box, det_row, kp_row);
}
// soft-arg-max partial of one joint over a set of pixels: m = max, s = sum exp(beta (v - m)), sx / sy = the same sum weighted with the
// pixel's column / row.  Two partials merge by rescaling both to the larger maximum (the usual streaming softmax).
struct Soft { float m, s, sx, sy; };
__device__ __forceinline__ Soft soft_merge(Soft a, Soft b, float beta) {
    if (b.s == 0.0f) return a;
    if (a.s == 0.0f) return b;
    const float m = fmaxf(a.m, b.m), fa = __expf(beta * (a.m - m)), fb = __expf(beta * (b.m - m));
    Soft r; r.m = m; r.s = a.s * fa + b.s * fb; r.sx = a.sx * fa + b.sx * fb; r.sy = a.sy * fa + b.sy * fb;
    return r;
}

// NHWC heat-maps (n, H, W, 17) float32: one workgroup per person.  Pixel tiles of 256 x 17 floats are staged through LDS
// with coalesced 16-byte loads; each lane then owns one pixel and reads its 17 values at stride 17 words (odd stride:
// conflict-free), keeping 17 running (max, index) pairs in registers.
#define TILE_PX 1024
#define DEC_T 1024            // 16 waves per person: only n (~20) workgroups exist, so each one is as wide as it can be
__global__ __launch_bounds__(DEC_T) void k_decode_nhwc(int n, const float* __restrict__ hm, int hm_h, int hm_w,
                                                     const int* __restrict__ view_of, const int* __restrict__ slot_of,
                                                     const float* __restrict__ boxes, int max_dets, double* __restrict__ det,
                                                     float* __restrict__ kp) {
    __shared__ __attribute__((aligned(16))) float tile[TILE_PX * J];
    __shared__ Best red[DEC_T / 64][J];
    const int crop = blockIdx.x, tid = threadIdx.x;
    const int HW = hm_h * hm_w;
    const float* src = hm + (size_t)crop * HW * J;
    Best best[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { best[j].v = -__builtin_huge_valf(); best[j].i = 0x7fffffff; }
    for (int base = 0; base < HW; base += TILE_PX) {
        const int npx = min(TILE_PX, HW - base);
        const int nfl = npx * J;
        const float* g = src + (size_t)base * J;
        if ((nfl & 3) == 0 && (((uintptr_t)g & 15) == 0)) {
            const float4* g4 = (const float4*)g; float4* t4 = (float4*)tile;
            for (int e = tid; e < nfl / 4; e += DEC_T) t4[e] = g4[e];
        } else {
            for (int e = tid; e < nfl; e += DEC_T) tile[e] = g[e];
        }
        __syncthreads();
        if (tid < npx) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float v = tile[tid * J + j];
                if (v > best[j].v) { best[j].v = v; best[j].i = base + tid; }   // strictly greater: earlier index wins ties
            }
        }
        __syncthreads();
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        Best b = wave_argmax(best[j]);
        if (lane == 0) red[wave][j] = b;
    }
    __syncthreads();
    if (tid < J) {
        Best b = red[0][tid];
#pragma unroll
        for (int w = 1; w < DEC_T / 64; ++w) b = better(b, red[w][tid]);
        double* row = det + ((size_t)view_of[crop] * max_dets + slot_of[crop]) * J * 3;
        write_keypoint(tid, b, hm_h, hm_w, boxes + crop * 4, row, kp ? kp + (size_t)crop * J * 3 : nullptr);
    }
}

// NCHW heat-maps (n, 17, H, W): one workgroup per (person, joint), plain coalesced scan.
__global__ __launch_bounds__(256) void k_decode_nchw(int n, const float* __restrict__ hm, int hm_h, int hm_w,
                                                     const int* __restrict__ view_of, const int* __restrict__ slot_of,
                                                     const float* __restrict__ boxes, int max_dets, double* __restrict__ det,
                                                     float* __restrict__ kp) {
    __shared__ Best red[4];
    const int crop = blockIdx.x / J, j = blockIdx.x % J, tid = threadIdx.x;
    const int HW = hm_h * hm_w;
    const float* src = hm + ((size_t)crop * J + j) * HW;
    Best b; b.v = -__builtin_huge_valf(); b.i = 0x7fffffff;
    for (int e = tid; e < HW; e += 256) { const float v = src[e]; if (v > b.v) { b.v = v; b.i = e; } }
    b = wave_argmax(b);
    if ((tid & 63) == 0) red[tid >> 6] = b;
    __syncthreads();
    if (tid == 0) {
        b = better(better(red[0], red[1]), better(red[2], red[3]));
        double* row = det + ((size_t)view_of[crop] * max_dets + slot_of[crop]) * J * 3;
        write_keypoint(j, b, hm_h, hm_w, boxes + crop * 4, row, kp ? kp + (size_t)crop * J * 3 : nullptr);
    }
}

// ---- final 1x1 convolution of the pose network: features NHWC bf16 (C channels) -> heat-maps NHWC float32 (J maps) ----------------
// One thread per pixel: 16-byte feature loads, the J x C float32 weights broadcast from LDS, J running sums in registers
// (float32 FMA chain over the channels in order, bias first); the workgroup's 256 x J outputs are contiguous in memory and
// leave through LDS as full 16-byte stores.  Streaming: C*2 bytes in, J*4 bytes out per pixel.
#define HEAD_T 256
template <int JN>
__global__ __launch_bounds__(HEAD_T) void k_head(int npix, const uint16_t* __restrict__ feat, int C, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];       // [JN*C] weights, then [HEAD_T*JN] output staging
    float* ws = hsm; float* os = hsm + JN * C;
    for (int i = threadIdx.x; i < JN * C; i += HEAD_T) ws[i] = w[i];
    __syncthreads();
    const int px = blockIdx.x * HEAD_T + threadIdx.x;
    float acc[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j] = bias ? bias[j] : 0.0f;
    if (px < npix) {
        const uint16_t* f = feat + (size_t)px * C;
        for (int c8 = 0; c8 < C; c8 += 8) {
            const uint4 v = *(const uint4*)(f + c8);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            float x[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[2 * k] = __uint_as_float(d[k] << 16); x[2 * k + 1] = __uint_as_float(d[k] & 0xffff0000u); }
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[j] = fmaf(x[k], ws[j * C + c8 + k], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) os[threadIdx.x * JN + j] = acc[j];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * HEAD_T * JN;             // this workgroup's outputs: [base, base + HEAD_T*JN) floats
    const size_t total = (size_t)npix * JN;
    for (int i = threadIdx.x * 4; i < HEAD_T * JN; i += HEAD_T * 4) {
        if (base + i + 3 < total) *(float4*)(out + base + i) = *(const float4*)(os + i);
        else for (int k = 0; k < 4; ++k) if (base + i + k < total) out[base + i + k] = os[i + k];
    }
}

extern "C" int pam_head_heatmaps(void* stream, int n_pix, const void* feat_bf16, int C, const float* w, const float* bias,
                                 int J_, float* out) {
    if (n_pix < 0 || !feat_bf16 || !w || !out || C <= 0 || C % 8 != 0 || J_ != PAM_J || (HEAD_T * J_) % 4 != 0) return PAM_E_ARG;
    if (n_pix == 0) return PAM_OK;
    const size_t lds = ((size_t)J_ * C + (size_t)HEAD_T * J_) * sizeof(float);
    hipLaunchKernelGGL((k_head<PAM_J>), dim3((n_pix + HEAD_T - 1) / HEAD_T), dim3(HEAD_T), lds, (hipStream_t)stream, n_pix,
                       (const uint16_t*)feat_bf16, C, w, bias, out);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// ---- head + arg-max in one pass: the heat-maps never reach memory ----------------------------------------------------------------
// k_head_argmax: workgroup = 256 consecutive pixels of ONE crop; every thread has its pixel's J values in registers (same FMA
// chain as k_head, so the values are bit-identical), the workgroup reduces (max, first index) per joint -- wave shuffles, then
// LDS staging, 8 lanes per joint -- and writes J candidates to cand[crop][tile][J].  k_argmax_finish: one wave per crop folds the
// tiles in order (strictly greater replaces: the lowest flat index wins ties, np.argmax's rule) and writes the keypoint rows.
template <int JN, bool SOFT>
__global__ __launch_bounds__(HEAD_T) void k_head_argmax(int HW, int tiles, const uint16_t* __restrict__ feat, int C,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ heat /* optional */, Best* __restrict__ cand,
                                                        int hm_w, float beta, Soft* __restrict__ scand /* SOFT only */) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];       // [JN*C] (unused: see ws), then Best[4][JN]
    // The weights are read through the kernel argument with wave-uniform indices: scalar loads (3.3 KB, resident in the scalar cache) feeding
    // the FMAs as SGPR operands, instead of an LDS copy broadcast-read 204 times per thread.  Head + finish 17.0 -> 15.5 us at 20 crops
    // (13 us of it is there at 8 crops too: two launches and their serial chains); the FMA chain and its order are unchanged.
    const float* __restrict__ ws = w; Best* red = (Best*)(hsm + JN * C);
    const int crop = blockIdx.x / tiles, tile = blockIdx.x - crop * tiles;
    const int lp = tile * HEAD_T + threadIdx.x;                       // pixel inside the crop = flat heat-map index
    const bool ok = lp < HW;
    float acc[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j] = bias ? bias[j] : 0.0f;
    if (ok) {
        const uint16_t* f = feat + ((size_t)crop * HW + lp) * C;
        auto fma8 = [&](const uint4 v, int c8) {
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            float x[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[2 * k] = __uint_as_float(d[k] << 16); x[2 * k + 1] = __uint_as_float(d[k] & 0xffff0000u); }
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[j] = fmaf(x[k], ws[j * C + c8 + k], acc[j]);
        };
        if (C == 48) {                                   // HRNet-W48: all six 16-byte pieces of the pixel in flight at once (same FMA order)
            uint4 v[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) v[q] = *(const uint4*)(f + 8 * q);
#pragma unroll
            for (int q = 0; q < 6; ++q) fma8(v[q], 8 * q);
        } else {
            for (int c8 = 0; c8 < C; c8 += 8) fma8(*(const uint4*)(f + c8), c8);
        }
        if (heat) {
            float* o = heat + ((size_t)crop * HW + lp) * JN;
#pragma unroll
            for (int j = 0; j < JN; ++j) o[j] = acc[j];
        }
    }
    // per-joint (max, first index) over the workgroup's 256 pixels: the values go through LDS pixel-major (pitch JN + 1 words), then
    // 8 lanes per joint scan 32 pixels each -- lane `part` takes pixels part, part + 8, ...: increasing index inside a lane, and the 8
    // lanes of a joint read 8 consecutive rows (different banks; blocks of 32 rows would all fall on one) -- and fold by 3 shuffles
    float* vs = (float*)(red + (HEAD_T / 64) * JN);                   // [HEAD_T][JN + 1]
#pragma unroll
    for (int j = 0; j < JN; ++j) vs[threadIdx.x * (JN + 1) + j] = ok ? acc[j] : -__builtin_huge_valf();
    __syncthreads();
    if (threadIdx.x < JN * 8) {
        const int j = threadIdx.x >> 3, part = threadIdx.x & 7;
        Best b; b.v = -__builtin_huge_valf(); b.i = 0x7fffffff;
#pragma unroll 8
        for (int q = 0; q < HEAD_T / 8; ++q) {
            const int px = q * 8 + part;
            const float v = vs[px * (JN + 1) + j];
            if (v > b.v) { b.v = v; b.i = tile * HEAD_T + px; }            // strictly greater: the first maximum of the part
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            Best o; o.v = __shfl_xor(b.v, off, 64); o.i = __shfl_xor(b.i, off, 64);
            b = better(b, o);
        }
        if (part == 0) cand[((size_t)crop * tiles + tile) * JN + j] = b;
        if constexpr (SOFT) {
            // after the butterfly every lane of the joint holds the tile's maximum: second pass over the same LDS values, weights
            // relative to it, then plain sums over the 8 lanes (all on the same maximum)
            float s0 = 0.f, sx = 0.f, sy = 0.f;
            if (b.v > -__builtin_huge_valf()) {
#pragma unroll 8
                for (int q = 0; q < HEAD_T / 8; ++q) {
                    const int px = q * 8 + part, lp2 = tile * HEAD_T + px;
                    const float e = __expf(beta * (vs[px * (JN + 1) + j] - b.v));       // pixels past the crop hold -inf: weight 0
                    const int yy = lp2 / hm_w, xx = lp2 - yy * hm_w;
                    s0 += e; sx += e * (float)xx; sy += e * (float)yy;
                }
            }
#pragma unroll
            for (int off = 4; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off, 64); sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); }
            if (part == 0) { Soft r; r.m = b.v; r.s = s0; r.sx = sx; r.sy = sy; scand[((size_t)crop * tiles + tile) * JN + j] = r; }
        }
    }
}
// soft-arg-max finish: one lane per joint merges the tiles' partials in order and writes the expected (column, row) through the box;
// the score is the maximum, as in the hard decode
__global__ __launch_bounds__(64) void k_softmax_finish(int tiles, const Soft* __restrict__ scand, float beta, int hm_h, int hm_w,
                                                       const int* __restrict__ view_of, const int* __restrict__ slot_of,
                                                       const float* __restrict__ boxes, int max_dets, double* __restrict__ det,
                                                       float* __restrict__ kp) {
    const int crop = blockIdx.x, j = threadIdx.x;
    if (j >= J) return;
    Soft a; a.m = -__builtin_huge_valf(); a.s = 0.f; a.sx = 0.f; a.sy = 0.f;
    for (int t = 0; t < tiles; ++t) a = soft_merge(a, scand[((size_t)crop * tiles + t) * J + j], beta);
    double* row = det + ((size_t)view_of[crop] * max_dets + slot_of[crop]) * J * 3;
    const double inv = a.s > 0.f ? 1.0 / (double)a.s : 0.0;
    write_keypoint_at(j, (double)a.sy * inv, (double)a.sx * inv, a.m, hm_h, hm_w, boxes + crop * 4, row, kp ? kp + (size_t)crop * J * 3 : nullptr);
}
__global__ __launch_bounds__(64) void k_argmax_finish(int tiles, const Best* __restrict__ cand, int hm_h, int hm_w,
                                                      const int* __restrict__ view_of, const int* __restrict__ slot_of,
                                                      const float* __restrict__ boxes, int max_dets, double* __restrict__ det,
                                                      float* __restrict__ kp) {
    const int crop = blockIdx.x, j = threadIdx.x;
    if (j >= J) return;
    Best b; b.v = -__builtin_huge_valf(); b.i = 0x7fffffff;
    for (int t0 = 0; t0 < tiles; t0 += 8) {                          // 8 candidates in flight at a time (a serial chain of 27 loads took 11 us)
        Best c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = min(t0 + q, tiles - 1);
            c[q] = cand[((size_t)crop * tiles + t) * J + j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (t0 + q < tiles && c[q].v > b.v) b = c[q];            // tiles are in index order: strictly greater keeps the first maximum
    }
    double* row = det + ((size_t)view_of[crop] * max_dets + slot_of[crop]) * J * 3;
    write_keypoint(j, b, hm_h, hm_w, boxes + crop * 4, row, kp ? kp + (size_t)crop * J * 3 : nullptr);
}

extern "C" long long pam_head_decode_scratch_bytes(int n, int hm_h, int hm_w) {
    if (n < 0 || hm_h <= 0 || hm_w <= 0) return -1;
    const long long tiles = ((long long)hm_h * hm_w + HEAD_T - 1) / HEAD_T;
    return (long long)n * tiles * J * (long long)sizeof(Best);
}

extern "C" int pam_head_decode(void* stream, int n, int hm_h, int hm_w, const void* feat_bf16, int C, const float* w, const float* bias,
                               int J_, float* dev_heatmaps_or_null, const int32_t* dev_view_of, const int32_t* dev_slot_of,
                               const float* dev_boxes, int max_dets, double* dev_det, float* dev_kp_xyc, void* dev_scratch) {
    if (n < 0 || !feat_bf16 || !w || !dev_view_of || !dev_slot_of || !dev_boxes || !dev_det || !dev_scratch || hm_h <= 0 || hm_w <= 0 ||
        C <= 0 || C % 8 != 0 || J_ != PAM_J)
        return PAM_E_ARG;
    if (n == 0) return PAM_OK;
    const int HW = hm_h * hm_w, tiles = (HW + HEAD_T - 1) / HEAD_T;
    const size_t lds = (size_t)J_ * C * sizeof(float) + (size_t)(HEAD_T / 64) * J_ * sizeof(Best) + (size_t)HEAD_T * (J_ + 1) * sizeof(float);
    hipLaunchKernelGGL((k_head_argmax<PAM_J, false>), dim3(n * tiles), dim3(HEAD_T), lds, (hipStream_t)stream, HW, tiles,
                       (const uint16_t*)feat_bf16, C, w, bias, dev_heatmaps_or_null, (Best*)dev_scratch, hm_w, 0.0f, (Soft*)nullptr);
    hipLaunchKernelGGL(k_argmax_finish, dim3(n), dim3(64), 0, (hipStream_t)stream, tiles, (const Best*)dev_scratch, hm_h, hm_w,
                       dev_view_of, dev_slot_of, dev_boxes, max_dets, dev_det, dev_kp_xyc);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// the same pass with a soft-arg-max decode: keypoint = sum_p softmax(beta * heat)_p * (column, row)_p per joint (sub-pixel), score = max
extern "C" long long pam_head_decode_soft_scratch_bytes(int n, int hm_h, int hm_w) {
    const long long b = pam_head_decode_scratch_bytes(n, hm_h, hm_w);
    return b < 0 ? b : b + b / (long long)sizeof(Best) * (long long)sizeof(Soft);
}
extern "C" int pam_head_decode_soft(void* stream, int n, int hm_h, int hm_w, const void* feat_bf16, int C, const float* w, const float* bias,
                                    int J_, float beta, float* dev_heatmaps_or_null, const int32_t* dev_view_of, const int32_t* dev_slot_of,
                                    const float* dev_boxes, int max_dets, double* dev_det, float* dev_kp_xyc, void* dev_scratch) {
    if (n < 0 || !feat_bf16 || !w || !dev_view_of || !dev_slot_of || !dev_boxes || !dev_det || !dev_scratch || hm_h <= 0 || hm_w <= 0 ||
        C <= 0 || C % 8 != 0 || J_ != PAM_J || !(beta > 0.0f))
        return PAM_E_ARG;
    if (n == 0) return PAM_OK;
    const int HW = hm_h * hm_w, tiles = (HW + HEAD_T - 1) / HEAD_T;
    const size_t lds = (size_t)J_ * C * sizeof(float) + (size_t)(HEAD_T / 64) * J_ * sizeof(Best) + (size_t)HEAD_T * (J_ + 1) * sizeof(float);
    Best* cand = (Best*)dev_scratch;
    Soft* scand = (Soft*)(cand + (size_t)n * tiles * J_);                 // 16-byte records behind the 8-byte ones (n * tiles * 17 * 8 is a multiple of 8)
    hipLaunchKernelGGL((k_head_argmax<PAM_J, true>), dim3(n * tiles), dim3(HEAD_T), lds, (hipStream_t)stream, HW, tiles,
                       (const uint16_t*)feat_bf16, C, w, bias, dev_heatmaps_or_null, cand, hm_w, beta, scand);
    hipLaunchKernelGGL(k_softmax_finish, dim3(n), dim3(64), 0, (hipStream_t)stream, tiles, (const Soft*)scand, beta, hm_h, hm_w,
                       dev_view_of, dev_slot_of, dev_boxes, max_dets, dev_det, dev_kp_xyc);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_preprocess_crops_ex(void* stream, int n, int n_total, const void* const* dev_frames, int frame_h, int frame_w,
                                       const int32_t* dev_view_of, const float* dev_boxes, int out_h, int out_w,
                                       int out_c, void* dev_out_bf16, int antialias) {
    if (n < 0 || n_total < n || !dev_frames || !dev_view_of || !dev_boxes || !dev_out_bf16 || out_h <= 0 || out_w <= 0 || (out_c != 3 && out_c != 8)) return PAM_E_ARG;
    if (n == 0) return PAM_OK;
    dim3 grid((out_h * out_w + 255) / 256, n_total);
    hipLaunchKernelGGL(k_preprocess_crops, grid, dim3(256), 0, (hipStream_t)stream, n, (const uint8_t* const*)dev_frames,
                       frame_h, frame_w, dev_view_of, dev_boxes, out_h, out_w, out_c, (uint16_t*)dev_out_bf16, antialias ? 1 : 0);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
extern "C" int pam_preprocess_crops(void* stream, int n, const void* const* dev_frames, int frame_h, int frame_w,
                                    const int32_t* dev_view_of, const float* dev_boxes, int out_h, int out_w,
                                    int out_c, void* dev_out_bf16) {
    return pam_preprocess_crops_ex(stream, n, n, dev_frames, frame_h, frame_w, dev_view_of, dev_boxes, out_h, out_w, out_c, dev_out_bf16, 0);
}

extern "C" int pam_decode_heatmaps(void* stream, int n, const float* dev_heatmaps, int nchw, int hm_h, int hm_w,
                                   const int32_t* dev_view_of, const int32_t* dev_slot_of, const float* dev_boxes,
                                   int max_dets, double* dev_det, float* dev_kp_xyc) {
    if (n < 0 || !dev_heatmaps || !dev_view_of || !dev_slot_of || !dev_boxes || !dev_det || hm_h <= 0 || hm_w <= 0) return PAM_E_ARG;
    if (n == 0) return PAM_OK;
    if (nchw)
        hipLaunchKernelGGL(k_decode_nchw, dim3(n * J), dim3(256), 0, (hipStream_t)stream, n, dev_heatmaps, hm_h, hm_w,
                           dev_view_of, dev_slot_of, dev_boxes, max_dets, dev_det, dev_kp_xyc);
    else
        hipLaunchKernelGGL(k_decode_nhwc, dim3(n), dim3(DEC_T), 0, (hipStream_t)stream, n, dev_heatmaps, hm_h, hm_w,
                           dev_view_of, dev_slot_of, dev_boxes, max_dets, dev_det, dev_kp_xyc);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// ---- measurement aid (bench.py): the shader clock the chip holds at this point of a stream ---------------------------------
// One wave spins for `ticks` periods of the 100 MHz reference counter and reports (delta s_memtime, delta s_memrealtime): shader
// clock [MHz] = 100 * d_memtime / d_memrealtime (MI355X guide, 'DVFS give-back' item 6).  Enqueued right before and right after the
// timed region it shows whether a run sat in a lower power state; it carries no data of the path.
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0; out[1] = r1 - r0;
}
extern "C" int pam_clock_probe(void* stream, unsigned long long* dev_out2, int microseconds) {
    if (!dev_out2 || microseconds < 1 || microseconds > 100000) return PAM_E_ARG;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_out2, (unsigned long long)microseconds * 100ull);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
