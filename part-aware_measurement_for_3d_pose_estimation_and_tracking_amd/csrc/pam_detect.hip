// libpam_hip.so, person-detector side: what `backend.YOLOv3` does around its conv stack for
// ivclabpose.PersonDetect (/root/reference/src/ivclabpose.py:116-120,183-204; the backend itself is not in the reference
// tree, so these follow the public Darknet YOLOv3 definition -- parity unpinned).  The Darknet-53 convolutions run on
// pam_conv.hip; the kernels here are the HBM-bound streaming pieces: frame resize, route(upsample, skip) and the
// three-scale box decode + greedy NMS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"

__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// ---- whole-frame resize -------------------------------------------------------------------------------------------
// One thread per output pixel: bilinear (half-pixel centres, border replicate = cv2.resize INTER_LINEAR) from the BGR
// uint8 frame, BGR->RGB, /255, stored as 8-channel bf16 NHWC (RGB + 5 zero channels: the conv kernels want Cin % 8 == 0).
__global__ __launch_bounds__(256) void k_resize_frames(int n, const uint8_t* const* __restrict__ frames, int H, int W,
                                                       int oh, int ow, uint16_t* __restrict__ out) {
    const int img_i = blockIdx.y;
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (img_i >= n || px >= oh * ow) return;
    const int oy = px / ow, ox = px - oy * ow;
    const uint8_t* __restrict__ img = frames[img_i];
    float sx = (ox + 0.5f) * ((float)W / (float)ow) - 0.5f;
    float sy = (oy + 0.5f) * ((float)H / (float)oh) - 0.5f;
    sx = fminf(fmaxf(sx, 0.0f), (float)(W - 1));
    sy = fminf(fmaxf(sy, 0.0f), (float)(H - 1));
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = sx - (float)x0, fy = sy - (float)y0;
    const uint8_t* r0 = img + ((size_t)y0 * W) * 3;
    const uint8_t* r1 = img + ((size_t)y1 * W) * 3;
    uint16_t o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {          // c indexes RGB; source is BGR
        const int sc = 2 - c;
        const float a = (float)r0[x0 * 3 + sc], b = (float)r0[x1 * 3 + sc];
        const float cc = (float)r1[x0 * 3 + sc], d = (float)r1[x1 * 3 + sc];
        const float top = a + (b - a) * fx, bot = cc + (d - cc) * fx;
        o[c] = f32_to_bf16((top + (bot - top) * fy) * (1.0f / 255.0f));
    }
    uint4 v;
    v.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); v.y = (uint32_t)o[2]; v.z = 0; v.w = 0;
    *(uint4*)(out + (((size_t)img_i * oh + oy) * ow + ox) * 8) = v;
}

extern "C" int pam_resize_frames(void* stream, int n, const void* const* dev_frames, int frame_h, int frame_w,
                                 int out_h, int out_w, void* dev_out_bf16) {
    if (n < 0 || !dev_frames || !dev_out_bf16 || frame_h <= 0 || frame_w <= 0 || out_h <= 0 || out_w <= 0) return PAM_E_ARG;
    if (n == 0) return PAM_OK;
    dim3 grid((out_h * out_w + 255) / 256, n);
    hipLaunchKernelGGL(k_resize_frames, grid, dim3(256), 0, (hipStream_t)stream, n, (const uint8_t* const*)dev_frames,
                       frame_h, frame_w, out_h, out_w, (uint16_t*)dev_out_bf16);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// ---- route(upsample x2, skip): out[n,y,x,:] = concat(a[n, y/2, x/2, :Ca], b[n, y, x, :Cb]); 16 B per thread -----------
__global__ __launch_bounds__(256) void k_upsample_concat(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                         uint16_t* __restrict__ out, int N, int H, int W, int Ca, int Cb) {
    const int C8 = (Ca + Cb) >> 3;
    const size_t total = (size_t)N * H * W * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int x = (int)(p % W);
        const size_t q = p / W;
        const int y = (int)(q % H), n = (int)(q / H);
        const int c = c8 * 8;
        uint4 v;
        if (c < Ca) v = *(const uint4*)(a + (((size_t)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * Ca + c);
        else        v = *(const uint4*)(b + p * Cb + (c - Ca));
        *(uint4*)(out + p * (Ca + Cb) + c) = v;
    }
}

extern "C" int pam_upsample_concat_nhwc_bf16(void* stream, const void* a, const void* b, void* out, int N, int H, int W,
                                             int Ca, int Cb) {
    if (!a || !b || !out || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Ca % 8 != 0 || Cb % 8 != 0 || Ca <= 0 || Cb <= 0)
        return PAM_E_ARG;
    const size_t total = (size_t)N * H * W * ((Ca + Cb) / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_upsample_concat, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)out, N, H, W, Ca, Cb);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

// ---- YOLO head decode + greedy NMS for one class --------------------------------------------------------------------
// One workgroup per image.  Candidates are numbered head-major, then cell (row-major), then anchor.  Pass 1 keeps, in
// that order, every candidate with sigmoid(objectness) * sigmoid(class logit) > score_thresh (at most PAM_YOLO_MAX_CAND of them enter
// NMS): a thread takes DET_K consecutive candidates per round and has all their score loads in flight at once (round 5: one candidate
// per thread and three barriers per 1 024 candidates was 11 serial memory round trips = 35 us of the 87 us launch), the kept ones are
// compacted in order through a block scan of the per-thread counts.  Pass 2 is the classic greedy NMS -- repeatedly take the best
// surviving score (ties: lowest candidate number), emit it, drop every survivor whose IoU with it exceeds nms_thresh -- run by ONE wave
// with the candidates in registers (NJ per lane), so a round is a wave reduction and NJ IoU tests without a workgroup barrier (two
// barriers per emitted box before: 1.5 us x 36 boxes of a random-weight network).
#define DET_T 1024
#define DET_K 8
struct YoloArgs {
    const uint16_t* head[3];
    int gh[3], gw[3], cs[3];
    float anchors[18];          // [head][anchor][w, h] in network-input pixels
    int net_w, net_h, nc, cls;
    float score_thresh, nms_thresh;
    int frame_w, frame_h, max_det;
    float* out; int* count;
    int n_img;
};
struct DBest { float v; int i; };
__device__ __forceinline__ DBest dbetter(DBest a, DBest b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct DetLds {
    float x1[PAM_YOLO_MAX_CAND], y1[PAM_YOLO_MAX_CAND], x2[PAM_YOLO_MAX_CAND], y2[PAM_YOLO_MAX_CAND], sc[PAM_YOLO_MAX_CAND];
    int wtot[DET_T / 64];
    int running;
};

// greedy NMS of candidates 0 .. ncand - 1 (ncand <= 64 NJ) by one wave: candidate j * 64 + lane lives in slot j of lane `lane`
template <int NJ, bool REGS>                                 // REGS: the boxes live in registers too (NJ <= 4); else they are re-read from LDS every round
__device__ __forceinline__ int det_nms_wave(const YoloArgs& a, const DetLds& L, int img, int ncand, int lane) {
    float mx1[REGS ? NJ : 1], my1[REGS ? NJ : 1], mx2[REGS ? NJ : 1], my2[REGS ? NJ : 1], msc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = j * 64 + lane;
        const bool ok = c < ncand;
        if (REGS) { mx1[j] = ok ? L.x1[c] : 0.f; my1[j] = ok ? L.y1[c] : 0.f; mx2[j] = ok ? L.x2[c] : 0.f; my2[j] = ok ? L.y2[c] : 0.f; }
        msc[j] = ok ? L.sc[c] : -1.0f;                                  // a dead or absent candidate has score -1
    }
    int kept = 0;
    while (kept < a.max_det) {
        DBest b; b.v = msc[0]; b.i = lane;
#pragma unroll
        for (int j = 1; j < NJ; ++j) { DBest t; t.v = msc[j]; t.i = j * 64 + lane; b = dbetter(b, t); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { DBest t; t.v = __shfl_xor(b.v, o, 64); t.i = __shfl_xor(b.i, o, 64); b = dbetter(b, t); }
        if (b.v < 0.0f) break;                                          // uniform
        const float kx1 = L.x1[b.i], ky1 = L.y1[b.i], kx2 = L.x2[b.i], ky2 = L.y2[b.i];
        if (lane == 0) {
            float* row = a.out + ((size_t)img * a.max_det + kept) * 5;
            row[0] = kx1; row[1] = ky1; row[2] = kx2; row[3] = ky2; row[4] = b.v;
        }
        const float karea = (kx2 - kx1) * (ky2 - ky1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = min(j * 64 + lane, PAM_YOLO_MAX_CAND - 1);
            const float x1 = REGS ? mx1[REGS ? j : 0] : L.x1[c], y1 = REGS ? my1[REGS ? j : 0] : L.y1[c];
            const float x2 = REGS ? mx2[REGS ? j : 0] : L.x2[c], y2 = REGS ? my2[REGS ? j : 0] : L.y2[c];
            const float iw = fminf(x2, kx2) - fmaxf(x1, kx1), ih = fminf(y2, ky2) - fmaxf(y1, ky1);
            const float inter = (iw > 0.0f && ih > 0.0f) ? iw * ih : 0.0f;
            const float uni = (x2 - x1) * (y2 - y1) + karea - inter;
            if (msc[j] >= 0.0f && (j * 64 + lane == b.i || inter > a.nms_thresh * uni)) msc[j] = -1.0f;
        }
        ++kept;
    }
    return kept;
}

__global__ __launch_bounds__(DET_T) void k_yolo_detect(YoloArgs a) {
    __shared__ DetLds L;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per0 = a.gh[0] * a.gw[0] * 3, per1 = a.gh[1] * a.gw[1] * 3, per2 = a.gh[2] * a.gw[2] * 3;
    const int total = per0 + per1 + per2;
    const int stride_a = 5 + a.nc;
    if (tid == 0) L.running = 0;
    __syncthreads();
    for (int base = 0; base < total; base += DET_T * DET_K) {
        // candidate q -> address of its 5 + nc values; the loads of all DET_K candidates are issued before the first is used
        const uint16_t* ptr[DET_K];
        uint16_t vo[DET_K], vc[DET_K];
#pragma unroll
        for (int k = 0; k < DET_K; ++k) {
            const int q = min(base + tid * DET_K + k, total - 1);      // clamped: a candidate past the end is loaded (again) and dropped
            int h = 0, r = q;
            if (r >= per0) { r -= per0; h = 1; if (r >= per1) { r -= per1; h = 2; } }
            const int cell = r / 3, an = r - cell * 3;
            const int gw = h == 0 ? a.gw[0] : (h == 1 ? a.gw[1] : a.gw[2]), gh = h == 0 ? a.gh[0] : (h == 1 ? a.gh[1] : a.gh[2]);
            const int cs = h == 0 ? a.cs[0] : (h == 1 ? a.cs[1] : a.cs[2]);
            const uint16_t* hp = h == 0 ? a.head[0] : (h == 1 ? a.head[1] : a.head[2]);
            ptr[k] = hp + ((size_t)img * gh * gw + cell) * cs + an * stride_a;
            vo[k] = ptr[k][4]; vc[k] = ptr[k][5 + a.cls];
        }
        float sc[DET_K];
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < DET_K; ++k) {
            sc[k] = sigmoidf_(bf16_to_f32(vo[k])) * sigmoidf_(bf16_to_f32(vc[k]));
            if (base + tid * DET_K + k < total && sc[k] > a.score_thresh) keep |= 1u << k;
        }
        // ordered compaction: exclusive scan of the per-thread counts over the workgroup (wave scan, wave totals, running base)
        const int cnt = __popc(keep);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) L.wtot[wave] = incl;
        __syncthreads();
        int off = L.running + incl - cnt;
        for (int w = 0; w < wave; ++w) off += L.wtot[w];
        if (keep) {
#pragma unroll
            for (int k = 0; k < DET_K; ++k) {
                if (!((keep >> k) & 1u)) continue;
                if (off < PAM_YOLO_MAX_CAND) {
                    const int q = base + tid * DET_K + k;
                    int h = 0, r = q;
                    if (r >= per0) { r -= per0; h = 1; if (r >= per1) { r -= per1; h = 2; } }
                    const int cell = r / 3, an = r - cell * 3;
                    const int gw = h == 0 ? a.gw[0] : (h == 1 ? a.gw[1] : a.gw[2]), gh = h == 0 ? a.gh[0] : (h == 1 ? a.gh[1] : a.gh[2]);
                    const int gy = cell / gw, gx = cell - gy * gw;
                    const uint16_t* p = ptr[k];
                    const float bx = (sigmoidf_(bf16_to_f32(p[0])) + (float)gx) / (float)gw;
                    const float by = (sigmoidf_(bf16_to_f32(p[1])) + (float)gy) / (float)gh;
                    const float bw = expf(bf16_to_f32(p[2])) * a.anchors[(h * 3 + an) * 2 + 0] / (float)a.net_w;
                    const float bh = expf(bf16_to_f32(p[3])) * a.anchors[(h * 3 + an) * 2 + 1] / (float)a.net_h;
                    L.x1[off] = (bx - 0.5f * bw) * (float)a.frame_w; L.x2[off] = (bx + 0.5f * bw) * (float)a.frame_w;
                    L.y1[off] = (by - 0.5f * bh) * (float)a.frame_h; L.y2[off] = (by + 0.5f * bh) * (float)a.frame_h;
                    L.sc[off] = sc[k];
                }
                ++off;
            }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < DET_T / 64; ++w) t += L.wtot[w]; L.running += t; }
        __syncthreads();
    }
    if (wave != 0) return;
    const int nfound = L.running;
    const int ncand = nfound < PAM_YOLO_MAX_CAND ? nfound : PAM_YOLO_MAX_CAND;
    static_assert(PAM_YOLO_MAX_CAND == 64 * 16, "NMS wave holds 16 candidates per lane");
    int kept;
    if (ncand <= 64) kept = det_nms_wave<1, true>(a, L, img, ncand, lane);
    else if (ncand <= 256) kept = det_nms_wave<4, true>(a, L, img, ncand, lane);
    else kept = det_nms_wave<16, false>(a, L, img, ncand, lane);
    if (lane == 0) { a.count[img] = kept; a.count[a.n_img + img] = nfound; }
}

// ---- the same decode + NMS with pass 1 spread over several workgroups per image (round 5) ----------------------------------------
// Pass 1 of k_yolo_detect is one CU's rate for divergent 2-byte loads (10 647 candidates x 2 loads per view: ~35 us of its 56 us).  Here a
// workgroup of 256 threads scores ONE run of 256 x DET_K consecutive candidates and writes its kept ones, in order, to its segment of
// a workspace in device memory; the workgroup that takes an image's last ticket (agent-scope release / acquire around an atomic counter:
// the segments come from other XCDs' L2s) concatenates the segments in run order -- the same candidate order as the one-workgroup
// kernel -- and runs the same one-wave NMS, with every candidate in registers (four waves per workgroup: 512 VGPRs per lane are there).
// workspace: [n_img] tickets (zero before the FIRST use; the kernel leaves them zero) | [n_img][runs] counts | [n_img][runs] DetSeg
struct DetSeg { float x1[PAM_YOLO_MAX_CAND], y1[PAM_YOLO_MAX_CAND], x2[PAM_YOLO_MAX_CAND], y2[PAM_YOLO_MAX_CAND], sc[PAM_YOLO_MAX_CAND]; };
#define DET_TS 256
static inline size_t det_ws_head_bytes(int n_img, int runs) { return (((size_t)n_img * (1 + runs) * sizeof(int)) + 255) & ~(size_t)255; }

__global__ __launch_bounds__(DET_TS) void k_yolo_detect_split(YoloArgs a, int* __restrict__ tickets, int* __restrict__ counts, DetSeg* __restrict__ segs) {
    __shared__ DetLds L;
    __shared__ int s_last;
    const int run = blockIdx.x, runs = gridDim.x, img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per0 = a.gh[0] * a.gw[0] * 3, per1 = a.gh[1] * a.gw[1] * 3, per2 = a.gh[2] * a.gw[2] * 3;
    const int total = per0 + per1 + per2;
    const int stride_a = 5 + a.nc;
    const int base = run * DET_TS * DET_K;
    DetSeg& seg = segs[(size_t)img * runs + run];
    {
        const uint16_t* ptr[DET_K];
        uint16_t vo[DET_K], vc[DET_K];
#pragma unroll
        for (int k = 0; k < DET_K; ++k) {
            const int q = min(base + tid * DET_K + k, total - 1);
            int h = 0, r = q;
            if (r >= per0) { r -= per0; h = 1; if (r >= per1) { r -= per1; h = 2; } }
            const int cell = r / 3, an = r - cell * 3;
            const int gw = h == 0 ? a.gw[0] : (h == 1 ? a.gw[1] : a.gw[2]), gh = h == 0 ? a.gh[0] : (h == 1 ? a.gh[1] : a.gh[2]);
            const int cs = h == 0 ? a.cs[0] : (h == 1 ? a.cs[1] : a.cs[2]);
            const uint16_t* hp = h == 0 ? a.head[0] : (h == 1 ? a.head[1] : a.head[2]);
            ptr[k] = hp + ((size_t)img * gh * gw + cell) * cs + an * stride_a;
            vo[k] = ptr[k][4]; vc[k] = ptr[k][5 + a.cls];
        }
        float sc[DET_K];
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < DET_K; ++k) {
            sc[k] = sigmoidf_(bf16_to_f32(vo[k])) * sigmoidf_(bf16_to_f32(vc[k]));
            if (base + tid * DET_K + k < total && sc[k] > a.score_thresh) keep |= 1u << k;
        }
        const int cnt = __popc(keep);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) L.wtot[wave] = incl;
        __syncthreads();
        int off = incl - cnt;
        for (int w = 0; w < wave; ++w) off += L.wtot[w];
        if (keep) {
#pragma unroll
            for (int k = 0; k < DET_K; ++k) {
                if (!((keep >> k) & 1u)) continue;
                if (off < PAM_YOLO_MAX_CAND) {
                    const int q = base + tid * DET_K + k;
                    int h = 0, r = q;
                    if (r >= per0) { r -= per0; h = 1; if (r >= per1) { r -= per1; h = 2; } }
                    const int cell = r / 3, an = r - cell * 3;
                    const int gw = h == 0 ? a.gw[0] : (h == 1 ? a.gw[1] : a.gw[2]), gh = h == 0 ? a.gh[0] : (h == 1 ? a.gh[1] : a.gh[2]);
                    const int gy = cell / gw, gx = cell - gy * gw;
                    const uint16_t* p = ptr[k];
                    const float bx = (sigmoidf_(bf16_to_f32(p[0])) + (float)gx) / (float)gw;
                    const float by = (sigmoidf_(bf16_to_f32(p[1])) + (float)gy) / (float)gh;
                    const float bw = expf(bf16_to_f32(p[2])) * a.anchors[(h * 3 + an) * 2 + 0] / (float)a.net_w;
                    const float bh = expf(bf16_to_f32(p[3])) * a.anchors[(h * 3 + an) * 2 + 1] / (float)a.net_h;
                    seg.x1[off] = (bx - 0.5f * bw) * (float)a.frame_w; seg.x2[off] = (bx + 0.5f * bw) * (float)a.frame_w;
                    seg.y1[off] = (by - 0.5f * bh) * (float)a.frame_h; seg.y2[off] = (by + 0.5f * bh) * (float)a.frame_h;
                    seg.sc[off] = sc[k];
                }
                ++off;
            }
        }
    }
    // every thread releases its segment stores; the last thread to get there publishes the run's count and takes a ticket
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < DET_TS / 64; ++w) t += L.wtot[w];
        __hip_atomic_store(&counts[(size_t)img * runs + run], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = __hip_atomic_fetch_add(&tickets[img], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == runs - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) __hip_atomic_store(&tickets[img], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    int nfound = 0, off = 0;
    for (int r = 0; r < runs; ++r) {
        const int c = __hip_atomic_load(&counts[(size_t)img * runs + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int take = min(min(c, PAM_YOLO_MAX_CAND), PAM_YOLO_MAX_CAND - off);
        const DetSeg& g = segs[(size_t)img * runs + r];
        for (int i = tid; i < take; i += DET_TS) {
            L.x1[off + i] = g.x1[i]; L.y1[off + i] = g.y1[i]; L.x2[off + i] = g.x2[i]; L.y2[off + i] = g.y2[i]; L.sc[off + i] = g.sc[i];
        }
        nfound += c; off += take;
    }
    __syncthreads();
    if (wave != 0) return;
    const int ncand = off;
    int kept;
    if (ncand <= 64) kept = det_nms_wave<1, true>(a, L, img, ncand, lane);
    else if (ncand <= 256) kept = det_nms_wave<4, true>(a, L, img, ncand, lane);
    else kept = det_nms_wave<16, true>(a, L, img, ncand, lane);
    if (lane == 0) { a.count[img] = kept; a.count[a.n_img + img] = nfound; }
}

static int det_fill_args(YoloArgs& a, int n_img, const void* const* heads, const int32_t* grid_h, const int32_t* grid_w, const int32_t* chan_stride,
                         const float* anchors, int net_w, int net_h, int num_classes, int class_id, float score_thresh, float nms_thresh,
                         int frame_w, int frame_h, int max_det, float* dev_out, int32_t* dev_count) {
    if (n_img < 0 || !heads || !grid_h || !grid_w || !chan_stride || !anchors || !dev_out || !dev_count || num_classes <= 0 ||
        class_id < 0 || class_id >= num_classes || max_det <= 0 || net_w <= 0 || net_h <= 0)
        return PAM_E_ARG;
    for (int h = 0; h < 3; ++h) {
        if (!heads[h] || grid_h[h] <= 0 || grid_w[h] <= 0 || chan_stride[h] < 3 * (5 + num_classes)) return PAM_E_ARG;
        a.head[h] = (const uint16_t*)heads[h]; a.gh[h] = grid_h[h]; a.gw[h] = grid_w[h]; a.cs[h] = chan_stride[h];
    }
    for (int i = 0; i < 18; ++i) a.anchors[i] = anchors[i];
    a.net_w = net_w; a.net_h = net_h; a.nc = num_classes; a.cls = class_id; a.score_thresh = score_thresh; a.nms_thresh = nms_thresh;
    a.frame_w = frame_w; a.frame_h = frame_h; a.max_det = max_det; a.out = dev_out; a.count = dev_count; a.n_img = n_img;
    return PAM_OK;
}
static int det_runs(const int32_t* grid_h, const int32_t* grid_w) {
    long long total = 0;
    for (int h = 0; h < 3; ++h) total += (long long)grid_h[h] * grid_w[h] * 3;
    return (int)((total + DET_TS * DET_K - 1) / (DET_TS * DET_K));
}
extern "C" long long pam_yolo_detect_workspace_bytes(int n_img, const int32_t* grid_h, const int32_t* grid_w) {
    if (n_img < 0 || !grid_h || !grid_w) return -1;
    for (int h = 0; h < 3; ++h) if (grid_h[h] <= 0 || grid_w[h] <= 0) return -1;
    const int runs = det_runs(grid_h, grid_w);
    return (long long)(det_ws_head_bytes(n_img, runs) + (size_t)n_img * runs * sizeof(DetSeg));
}
extern "C" int pam_yolo_detect_ws(void* stream, int n_img, const void* const* heads, const int32_t* grid_h, const int32_t* grid_w,
                                  const int32_t* chan_stride, const float* anchors, int net_w, int net_h, int num_classes,
                                  int class_id, float score_thresh, float nms_thresh, int frame_w, int frame_h, int max_det,
                                  float* dev_out, int32_t* dev_count, void* dev_workspace, long long workspace_bytes) {
    YoloArgs a;
    const int rc = det_fill_args(a, n_img, heads, grid_h, grid_w, chan_stride, anchors, net_w, net_h, num_classes, class_id, score_thresh,
                                 nms_thresh, frame_w, frame_h, max_det, dev_out, dev_count);
    if (rc != PAM_OK) return rc;
    if (n_img == 0) return PAM_OK;
    if (!dev_workspace || workspace_bytes < pam_yolo_detect_workspace_bytes(n_img, grid_h, grid_w) || ((uintptr_t)dev_workspace & 15)) return PAM_E_ARG;
    const int runs = det_runs(grid_h, grid_w);
    int* tickets = (int*)dev_workspace;
    int* counts = tickets + n_img;
    DetSeg* segs = (DetSeg*)((char*)dev_workspace + det_ws_head_bytes(n_img, runs));
    hipLaunchKernelGGL(k_yolo_detect_split, dim3(runs, n_img), dim3(DET_TS), 0, (hipStream_t)stream, a, tickets, counts, segs);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_yolo_detect(void* stream, int n_img, const void* const* heads, const int32_t* grid_h, const int32_t* grid_w,
                               const int32_t* chan_stride, const float* anchors, int net_w, int net_h, int num_classes,
                               int class_id, float score_thresh, float nms_thresh, int frame_w, int frame_h, int max_det,
                               float* dev_out, int32_t* dev_count) {
    YoloArgs a;
    const int rc = det_fill_args(a, n_img, heads, grid_h, grid_w, chan_stride, anchors, net_w, net_h, num_classes, class_id, score_thresh,
                                 nms_thresh, frame_w, frame_h, max_det, dev_out, dev_count);
    if (rc != PAM_OK) return rc;
    if (n_img == 0) return PAM_OK;
    hipLaunchKernelGGL(k_yolo_detect, dim3(n_img), dim3(DET_T), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PAM_OK : PAM_E_HIP;
}
