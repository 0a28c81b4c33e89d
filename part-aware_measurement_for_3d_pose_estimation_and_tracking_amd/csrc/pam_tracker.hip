// libpam_hip.so, tracker part: device-resident tracker state, the fused per-frame kernel (one workgroup per scene)
// and the per-operator test kernels, behind the C ABI of include/pam.h.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <limits.h>
#include <string>
#include "pam_device.hpp"

using namespace pam;

#define TENTATIVE 1
#define CONFIRMED 2
#define DELETED 3
#define T_NONE (INT_MIN / 2)
#define ST_TRACK_OVERFLOW 1
#define ST_HYP_OVERFLOW 2
#define ST_LSAP_INFEASIBLE 4
#define ST_NDET_CLAMPED 8       /* a per-view detection count outside [0, max_dets] was clamped (pam_frame_dev takes device counts unchecked) */
#define ST_INPUT_VOID 16        /* the producer of this frame's keypoints declared them void (pam_set_input_guard / the records' void flag): the
                                   frame was NOT applied -- state untouched, record without tracks, out_i[3] = first void frame of the run of void frames */
#define BLOCK 256

struct Dims { int C, MAXP, MAXT, HCAP, MAXH, S, N1, N2; };

struct Carver {
    char* p;
    template <typename T> __host__ __device__ T* take(size_t n) {
        uintptr_t a = (uintptr_t)p;
        const uintptr_t al = sizeof(T) < 8 ? sizeof(T) : 8;
        a = (a + al - 1) & ~(al - 1);
        T* r = (T*)a;
        p = (char*)(a + n * sizeof(T));
        return r;
    }
    __host__ __device__ char* bytes(size_t n) { uintptr_t a = ((uintptr_t)p + 7) & ~(uintptr_t)7; p = (char*)(a + n); return (char*)a; }
};

// ---- per-scene tracker state (HBM resident across frames) -------------------------------------------------------
struct SceneState {
    int* hdr;        // [0] n_tracks [1] next_id [2] status of this frame [3] used_lo [4] used_hi [5] OR of every frame's status since pam_reset
    int* order;      // list position -> slot (the reference's self.tracks order)
    int *track_id, *hits, *age, *tsu, *already, *state, *p2d_n, *h_head, *h_len, *jv_V;
    int* p2d_order;  // [slot][k]   camera ids in dict-insertion order (Appendix A-5)
    int* p2d_time;   // [slot][cid] frame of the stored 2D pose or T_NONE
    int* cur_det;    // [slot][cid] detection index matched in the current frame or -1
    int* hist_time;  // [slot][HCAP] ring
    int* jv_count;   // [slot][17]  kept views per joint of the newest pose
    float* vel;      // [slot][51]  float32 velocity (IterativeTracker.py:391)
    double* p2d_pose;  // [slot][cid][51] rows (y,x,score)
    double* hist;      // [slot][HCAP][51] ring of smoothed poses
};
__host__ __device__ inline char* carve_state(char* base, const Dims& d, SceneState& s) {
    Carver c{base};
    s.hdr = c.take<int>(8);
    s.order = c.take<int>(d.MAXT);
    s.track_id = c.take<int>(d.MAXT); s.hits = c.take<int>(d.MAXT); s.age = c.take<int>(d.MAXT);
    s.tsu = c.take<int>(d.MAXT); s.already = c.take<int>(d.MAXT); s.state = c.take<int>(d.MAXT);
    s.p2d_n = c.take<int>(d.MAXT); s.h_head = c.take<int>(d.MAXT); s.h_len = c.take<int>(d.MAXT);
    s.jv_V = c.take<int>(d.MAXT);
    s.p2d_order = c.take<int>((size_t)d.MAXT * d.C);
    s.p2d_time = c.take<int>((size_t)d.MAXT * d.C);
    s.cur_det = c.take<int>((size_t)d.MAXT * d.C);
    s.hist_time = c.take<int>((size_t)d.MAXT * d.HCAP);
    s.jv_count = c.take<int>((size_t)d.MAXT * J);
    s.vel = c.take<float>((size_t)d.MAXT * J3);
    s.p2d_pose = c.take<double>((size_t)d.MAXT * d.C * J3);
    s.hist = c.take<double>((size_t)d.MAXT * d.HCAP * J3);
    return c.bytes(0);
}

// ---- per-scene scratch (rewritten every frame) -------------------------------------------------------------------
struct Scratch {
    int* misc;       // [0] hyp_n [1] n_tracks at frame start
    int* dt;         // [MAXT]
    double* aff;     // [C][MAXT][MAXP]
    char* lsap1;     // [C] x lsap_scratch_bytes(N1)
    int *as_rows, *as_cols;  // [C][N1]
    int* match_det;  // [MAXT][C] by OLD list position
    unsigned char* taken;  // [C][MAXP]
    int *um_n, *um_idx;    // [C], [C][MAXP]
    int *hyp_size, *hyp_view, *hyp_det;   // [MAXH], [MAXH][C], [MAXH][C]
    double* hc_cost; unsigned char* hc_veto;   // [MAXH][MAXP]
    char* lsap2; int *l2_rows, *l2_cols;
    int *sel_n, *sel_cid, *sel_T;   // [MAXT], [MAXT][C]
    double *pred, *raw3d;           // [MAXT][51]
    uint32_t* conf; double* rayd;   // [MAXT][17][C]
    uint32_t* keep; int* nview; int* ok;   // [MAXT][17], [MAXT][17], [MAXT]
    uint32_t* h_conf; float* h_sum;        // [MAXH][17][C]
    uint32_t* h_keep; double* h_pose; int* h_slot;   // [MAXH][17], [MAXH][51], [MAXH]
    double* rpart;                  // [MAXT][17][4][10] partial triangular factors of the split DLT
};
// small, latency-critical arrays first (they are carved out of LDS when they fit: the serial LSAP walks, atomically built
// conflict masks and every index-chasing read then cost an LDS access instead of an L2 round trip), bulk arrays after
__host__ __device__ inline char* carve_ws_hot(char* base, const Dims& d, Scratch& w) {
    Carver c{base};
    w.aff = c.take<double>((size_t)d.C * d.MAXT * d.MAXP);
    w.lsap1 = c.bytes((size_t)d.C * ((lsap_scratch_bytes(d.N1) + 7) & ~(size_t)7));
    w.misc = c.take<int>(8);
    w.dt = c.take<int>(d.MAXT);
    w.as_rows = c.take<int>((size_t)d.C * d.N1); w.as_cols = c.take<int>((size_t)d.C * d.N1);
    w.match_det = c.take<int>((size_t)d.MAXT * d.C);
    w.um_n = c.take<int>(d.C); w.um_idx = c.take<int>((size_t)d.C * d.MAXP);
    w.sel_n = c.take<int>(d.MAXT); w.sel_cid = c.take<int>((size_t)d.MAXT * d.C); w.sel_T = c.take<int>((size_t)d.MAXT * d.C);
    w.conf = c.take<uint32_t>((size_t)d.MAXT * J * d.C);
    w.keep = c.take<uint32_t>((size_t)d.MAXT * J); w.nview = c.take<int>((size_t)d.MAXT * J); w.ok = c.take<int>(d.MAXT);
    w.taken = c.take<unsigned char>((size_t)d.C * d.MAXP);
    return c.bytes(0);
}
__host__ __device__ inline char* carve_ws_bulk(char* base, const Dims& d, Scratch& w) {
    Carver c{base};
    w.hyp_size = c.take<int>(d.MAXH); w.hyp_view = c.take<int>((size_t)d.MAXH * d.C);
    w.hyp_det = c.take<int>((size_t)d.MAXH * d.C);
    w.hc_cost = c.take<double>((size_t)d.MAXH * d.MAXP);
    w.hc_veto = c.take<unsigned char>((size_t)d.MAXH * d.MAXP);
    w.lsap2 = c.bytes(lsap_scratch_bytes(d.N2));
    w.l2_rows = c.take<int>(d.N2); w.l2_cols = c.take<int>(d.N2);
    w.pred = c.take<double>((size_t)d.MAXT * J3); w.raw3d = c.take<double>((size_t)d.MAXT * J3);
    w.rayd = c.take<double>((size_t)d.MAXT * J * d.C);
    w.h_conf = c.take<uint32_t>((size_t)d.MAXH * J * d.C); w.h_sum = c.take<float>((size_t)d.MAXH * J * d.C);
    w.h_keep = c.take<uint32_t>((size_t)d.MAXH * J); w.h_pose = c.take<double>((size_t)d.MAXH * J3);
    w.h_slot = c.take<int>(d.MAXH);
    w.rpart = c.take<double>((size_t)d.MAXT * J * 4 * 10);
    return c.bytes(0);
}
__host__ __device__ inline char* carve_ws(char* base, const Dims& d, Scratch& w) {
    return carve_ws_bulk(carve_ws_hot(base, d, w), d, w);
}
__host__ __device__ inline size_t hot_bytes(const Dims& d) {
    Scratch w;
    return ((size_t)(uintptr_t)carve_ws_hot((char*)0, d, w) + 15) & ~(size_t)15;      // offset from a zero base
}
// integer part of the scene state (everything before `vel`): contiguous, copied to LDS for the duration of a frame
__host__ __device__ inline size_t state_int_bytes(const Dims& d) {
    SceneState s;
    carve_state((char*)0, d, s);
    return (size_t)(uintptr_t)(char*)s.vel;
}

struct FrameArgs {
    Dims d;
    CamSet cs;
    const PamParams* prm;
    char* state; size_t state_stride;
    char* ws; size_t ws_stride;
    const int* n_det; const double* det;
    const int* view_row;   // view-sharded input (pam_frame_dev_views): det = the all-gathered records, view v's record is row view_row[v];
                           // record = (MAXP + 1) x 51 doubles: MAXP detection rows, then the view's detection count in the first double
    int* out_i; double* out_d;
    PamOutLayout ol;
    int frame_id;
    int hot_in_lds;
    const int* guard;      // pam_set_input_guard: a device word the producer of the keypoints raises when they are void (pam_flag_gate's dev_void)
};

__device__ const int g_zeroT[PAM_MAX_VIEWS] = {0};

__device__ __forceinline__ double now_s() { return (double)__builtin_amdgcn_s_memrealtime() * 1e-8; }

// =====================================================================================================================
// The per-frame step: IterativeTracker.tracking (IterativeTracker.py:115-180) + output collection
// (ivclabpose.py:259-287).  One 256-thread workgroup per scene; phases separated by workgroup barriers.
// =====================================================================================================================
// PART: 0 = the whole step in one launch.  Rigs with more than 8 cameras run it as THREE launches (round 6): 1 = P0-P4a (association,
// view selection), 2 = P4b (the epipolar conflict sets: nT x V(V-1)/2 x 17 independent items -- 55 000 on the 31-camera rig, which one
// workgroup walked in 54 passes of dependent L2 loads, 136 of its 297 us) spread over MANY workgroups, 3 = P4c-P7.  The split forms keep
// all scratch and the integer state in global memory (hot_in_lds = 0); one scene per handle.
template <int NTHREADS, int PART = 0>
__global__ __launch_bounds__(NTHREADS, 4) void k_frame(FrameArgs A) {      // 4 waves per SIMD: four 256-thread scenes per CU (<= 128 VGPRs)
    const Dims d = A.d;
    const int C = d.C, MAXP = d.MAXP, MAXT = d.MAXT, HCAP = d.HCAP, MAXH = d.MAXH;
    const int sidx = PART == 2 ? 0 : blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    // P4b's items are dealt over the whole grid in the split form
    const int it0 = PART == 2 ? (int)blockIdx.x * (int)blockDim.x + tid : tid, itS = PART == 2 ? (int)(gridDim.x * blockDim.x) : NT;
    SceneState st; Scratch ws;
    carve_state(A.state + (size_t)sidx * A.state_stride, d, st);
    carve_ws(A.ws + (size_t)sidx * A.ws_stride, d, ws);
    extern __shared__ __attribute__((aligned(16))) char hot_lds[];
    char* const st_glob = A.state + (size_t)sidx * A.state_stride;
    const size_t st_ints = state_int_bytes(d);
    // ---- input guard: keypoints whose producer gave up (a device-side gate of the captured HRNet forward timed out, csrc/pam_sync.hip)
    //      must never reach the tracker state -- the reference's PersonPoseDetect (/root/reference/src/ivclabpose.py:208-212) cannot hand
    //      out a frame it later disowns.  The word is raised on THIS device (A.guard) or travels with a view-sharded record (second double
    //      of its count row, set by the rank that produced it).  A void frame leaves the state as it is; hdr[6] remembers the first frame
    //      of the current run of void frames (+ 1) so that the host knows where to resume.
    if (PART >= 2) {                                     // the split step decides ONCE, in its first launch (the word may rise between the launches)
        if (ws.misc[2]) return;
    } else {
        int voided = A.guard ? __hip_atomic_load(A.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        if (A.view_row)
            for (int v = 0; v < d.C; ++v) voided |= (A.det[((size_t)A.view_row[v] * (d.MAXP + 1) + d.MAXP) * J3 + 1] != 0.0);
        if (PART == 1 && threadIdx.x == 0) ws.misc[2] = voided;
        if (voided) {
            if (threadIdx.x == 0) {
                int* oi = A.out_i + (size_t)sidx * A.ol.int_words;
                double* od = A.out_d + (size_t)sidx * A.ol.dbl_words;
                if (st.hdr[6] == 0) st.hdr[6] = A.frame_id + 1;
                oi[0] = 0; oi[1] = ST_INPUT_VOID | (st.hdr[5] << 16); oi[2] = A.frame_id; oi[3] = st.hdr[6] - 1;
                const double t = now_s();
                for (int k = 0; k < 12; ++k) od[k] = t;
            }
            return;
        }
    }
    char* const ws_glob = A.ws + (size_t)sidx * A.ws_stride;         // the hot scratch's home in global memory (between the launches of a split step)
    if (A.hot_in_lds) {
        carve_ws_hot(hot_lds, d, ws);                                   // small scratch in LDS
        char* st_lds = hot_lds + hot_bytes(d);                          // integer scene state in LDS for this frame
        for (size_t o = (size_t)threadIdx.x * 4; o < st_ints; o += (size_t)blockDim.x * 4) *(int*)(st_lds + o) = *(const int*)(st_glob + o);
        if (PART == 3)                                                  // ... and what the first two launches left in the hot scratch
            for (size_t o = (size_t)threadIdx.x * 4; o < hot_bytes(d); o += (size_t)blockDim.x * 4) *(int*)(hot_lds + o) = *(const int*)(ws_glob + o);
        const ptrdiff_t sh = st_lds - st_glob;
#define REBASE(p) p = (int*)((char*)(p) + sh)
        REBASE(st.hdr); REBASE(st.order); REBASE(st.track_id); REBASE(st.hits); REBASE(st.age); REBASE(st.tsu); REBASE(st.already);
        REBASE(st.state); REBASE(st.p2d_n); REBASE(st.h_head); REBASE(st.h_len); REBASE(st.jv_V); REBASE(st.p2d_order);
        REBASE(st.p2d_time); REBASE(st.cur_det); REBASE(st.hist_time); REBASE(st.jv_count);
#undef REBASE
        if (PART == 1 && threadIdx.x == 0) ws.misc[2] = 0;               // (the LDS copy goes home at the end of this launch: not void, see above)
        __syncthreads();
    }
    const PamParams& prm = *A.prm;
    const CamSet cs = A.cs;
    const int frame = A.frame_id;
    const int* vrow = A.view_row;                                       // non-null: records gathered from the ranks (one scene)
    const int* n_det = vrow ? nullptr : A.n_det + (size_t)sidx * C;
    const double* det = vrow ? A.det : A.det + (size_t)sidx * C * MAXP * J3;
    const size_t recd = vrow ? (size_t)(MAXP + 1) * J3 : (size_t)MAXP * J3;
    int* out_i = A.out_i + (size_t)sidx * A.ol.int_words;
    double* out_d = A.out_d + (size_t)sidx * A.ol.dbl_words;
#define DET(v, k) (det + (size_t)(vrow ? vrow[v] : (v)) * recd + (size_t)(k) * J3)
    // view-sharded records carry the count as a double: a non-finite or out-of-range value is clamped explicitly (never cast: the
    // conversion of a NaN is undefined) and raises ST_NDET_CLAMPED below
    auto rec_count = [&](int v) -> double { return det[(size_t)vrow[v] * recd + (size_t)MAXP * J3]; };
    auto count_of = [&](int v) -> int {
        if (!vrow) return n_det[v];
        const double c = rec_count(v);
        return (c >= 0.0 && c <= (double)MAXP) ? (int)c : (c > (double)MAXP ? MAXP + 1 : -1);
    };
#define NDET_RAW(v) count_of(v)
    const int nT = st.hdr[0];
    // device-side counts are not validated by the host (pam_frame_dev): clamp them so that no index leaves det / ws.taken
#define NDET(v) min(max(NDET_RAW(v), 0), MAXP)

    if (PART <= 1) {
    // ---- P0: add_age, time gaps (IterativeTracker.py:126-129) -------------------------------------------------------
    if (tid == 0) {
        out_d[0] = now_s(); ws.misc[1] = nT;
        int bad = 0;
        for (int v = 0; v < C; ++v) { const int nv = NDET_RAW(v); bad |= (nv < 0 || nv > MAXP); }
        st.hdr[2] = bad ? ST_NDET_CLAMPED : 0;                       // the status word is per frame (out_i[1]); nothing sticks
    }
    __syncthreads();
    for (int i = tid; i < nT; i += NT) {
        const int s = st.order[i];
        st.already[s] = 0; st.age[s] += 1; st.tsu[s] += 1;
        const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
        ws.dt[i] = frame - st.hist_time[s * HCAP + newest];
        for (int v = 0; v < C; ++v) st.cur_det[s * C + v] = -1;
    }
    for (int k = tid; k < MAXT * C; k += NT) ws.match_det[k] = -1;
    for (int k = tid; k < C * MAXP; k += NT) ws.taken[k] = 0;
    __syncthreads();

    // ---- P1: re-projection affinity for every (view, track, detection) (:137-149) ----------------------------------
    if (MAXP == 8 || MAXP == 16 || MAXP == 32) {
        // the MAXP slots of a (view, track) pair sit in MAXP neighbouring lanes: the track's 17 re-projections are computed once per
        // group and shared through the wave (track_det_affinity_group) instead of once per detection; whole waves iterate together
        const int total = C * nT * MAXP;
        for (int it = tid; it - (tid & 63) < total; it += NT) {
            const bool in = it < total;
            const int ic = in ? it : 0;
            const int v = ic / (nT * MAXP), r = ic % (nT * MAXP), i = r / MAXP, k = r % MAXP;
            const bool live = in && k < NDET(v);
            const int s = st.order[i];
            const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
            const int dt = ws.dt[i];
            const double e = (dt >= 0 && dt < PAM_EXP_TABLE) ? prm.exp_lambda_a[dt] : exp(prm.lambda_a * (double)dt);
            const float* Pv = cs.P + v * 12;
            const double* X3 = st.hist + ((size_t)s * HCAP + newest) * J3;
            const double* dk = DET(v, live ? k : 0);
            const double a = MAXP == 8 ? track_det_affinity_group<8>(Pv, X3, dk, prm.alpha2d * (double)dt, e, prm.count_gate, live)
                           : MAXP == 16 ? track_det_affinity_group<16>(Pv, X3, dk, prm.alpha2d * (double)dt, e, prm.count_gate, live)
                                        : track_det_affinity_group<32>(Pv, X3, dk, prm.alpha2d * (double)dt, e, prm.count_gate, live);
            if (live) ws.aff[((size_t)v * MAXT + i) * MAXP + k] = a;
        }
    } else
    for (int it = tid; it < C * nT * MAXP; it += NT) {
        const int v = it / (nT * MAXP), r = it % (nT * MAXP), i = r / MAXP, k = r % MAXP;
        if (k >= NDET(v)) continue;
        const int s = st.order[i];
        const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
        const int dt = ws.dt[i];
        const double e = (dt >= 0 && dt < PAM_EXP_TABLE) ? prm.exp_lambda_a[dt] : exp(prm.lambda_a * (double)dt);
        ws.aff[((size_t)v * MAXT + i) * MAXP + k] =
            track_det_affinity(cs.P + v * 12, st.hist + ((size_t)s * HCAP + newest) * J3, DET(v, k),
                               prm.alpha2d * (double)dt, e, prm.count_gate);
    }
    __syncthreads();
#ifdef PAM_FINE_STAMPS
    if (tid == 0) out_d[12] = now_s();
#endif

    // ---- P2: one assignment problem per view (:150-160) ---------------------------------------------------------
    // one WAVE per view (lsap_solve_wave: a lane per column / row), views dealt over the workgroup's waves
    for (int v = tid >> 6; v < C; v += NT >> 6) {
        const int m = NDET(v);
        if (nT > 0 && m > 0) {
            int* rows = ws.as_rows + v * d.N1; int* cols = ws.as_cols + v * d.N1;
            const double* aff = ws.aff + (size_t)v * MAXT * MAXP;
            int row, col;
            const int np = lsap_solve_wave(nT, m, aff, MAXP, -1.0, rows, cols, row, col);
            if (np < 0 && (tid & 63) == 0) atomicOr(&st.hdr[2], ST_LSAP_INFEASIBLE);
            if ((tid & 63) < np && aff[row * MAXP + col] > 0.0) {
                ws.match_det[row * C + v] = col;
                ws.taken[v * MAXP + col] = 1;
            }
        }
    }
    __syncthreads();
#ifdef PAM_FINE_STAMPS
    if (tid == 0) out_d[13] = now_s();
#endif

    // ---- P3: add_pose (:155-160,289-298), unmatched lists + confidence filter (:56-61,163-167) ----------------
    for (int i = tid; i < nT; i += NT) {
        const int s = st.order[i];
        for (int v = 0; v < C; ++v) {
            const int k = ws.match_det[i * C + v];
            if (k < 0) continue;
            st.already[s] = 1;
            if (st.p2d_time[s * C + v] == T_NONE) { st.p2d_order[s * C + st.p2d_n[s]] = v; st.p2d_n[s] += 1; }
            st.p2d_time[s * C + v] = frame;
            st.cur_det[s * C + v] = k;
        }
    }
    for (int it = tid; it < nT * C * J3; it += NT) {
        const int i = it / (C * J3), r = it % (C * J3), v = r / J3, e = r % J3;
        const int k = ws.match_det[i * C + v];
        if (k >= 0) st.p2d_pose[((size_t)st.order[i] * C + v) * J3 + e] = DET(v, k)[e];
    }
    for (int v = tid; v < C; v += NT) {
        int n = 0;
        for (int k = 0; k < NDET(v); ++k)
            if (!ws.taken[v * MAXP + k] && believe(DET(v, k)) > prm.conf_threshold) ws.um_idx[v * MAXP + n++] = k;
        ws.um_n[v] = n;
    }
    __syncthreads();
    if (tid == 0) out_d[1] = now_s();

    // ---- P4a: views offered to each track (:310-325) and constant-velocity prediction (:341) ------------------
    for (int i = tid; i < nT; i += NT) {
        const int s = st.order[i];
        int n = 0;
        if (st.already[s])
            for (int k = 0; k < st.p2d_n[s]; ++k) {
                const int cid = st.p2d_order[s * C + k];
                const int T = frame - st.p2d_time[s * C + cid];
                if (T <= 3) { ws.sel_cid[i * C + n] = cid; ws.sel_T[i * C + n] = T; ++n; }
            }
        ws.sel_n[i] = (st.already[s] && n >= 2) ? n : 0;
    }
    for (int it = tid; it < nT * J * C; it += NT) ws.conf[it] = 0u;
    for (int it = tid; it < nT * J3; it += NT) {
        const int i = it / J3, e = it % J3, s = st.order[i];
        const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
        const float step = st.vel[s * J3 + e] * (float)(frame - st.hist_time[s * HCAP + newest]);   // float32 product
        ws.pred[i * J3 + e] = st.hist[((size_t)s * HCAP + newest) * J3 + e] + (double)step;
    }
    __syncthreads();

    if (tid == 0) out_d[4] = now_s();
    }                                                    // PART <= 1
    if (PART == 1) {                                     // hand over: hot scratch and integer state back to global memory
        if (A.hot_in_lds) {
            const char* st_lds = hot_lds + hot_bytes(d);
            for (size_t o = (size_t)threadIdx.x * 4; o < hot_bytes(d); o += (size_t)blockDim.x * 4) *(int*)(ws_glob + o) = *(const int*)(hot_lds + o);
            for (size_t o = (size_t)threadIdx.x * 4; o < st_ints; o += (size_t)blockDim.x * 4) *(int*)(st_glob + o) = *(const int*)(st_lds + o);
        }
        if (tid == 0) out_d[15] = now_s();               // split step: end of launch 1 / [12], [13] launch 2's first workgroup / [14] start of launch 3
        return;
    }
    if (PART == 2 && blockIdx.x == 0 && tid == 0) out_d[12] = now_s();
    if (PART == 3 && tid == 0) out_d[14] = now_s();
    // ---- P4b: per-joint conflict sets of the part-aware filter (matching.py:115-151, IterativeTracker.py:345-346): one
    //           lane per (track, joint, view pair r < c), bits OR-ed into the row masks; back-projection ray distances
    //           (matching.py:254-270) one lane per (track, joint, view) ---------------------------------------------------
    int maxV = 0;
    for (int i = 0; i < nT; ++i) maxV = max(maxV, ws.sel_n[i]);
    if (PART != 3) {                                     // (launch 3 of the split step finds the conflict sets and ray distances done)
        const int npair = maxV * (maxV - 1) / 2;
        // joints fastest: the 17 lanes of one (track, view pair) share both fundamental matrices and read consecutive joints
        for (int it = it0; it < nT * npair * J; it += itS) {
            const int i = it / (npair * J), r2 = it % (npair * J), pq = r2 / J, j = r2 % J;
            const int V = ws.sel_n[i];
            // unrank pq -> (r, c), r < c < maxV (row-major over the strict upper triangle), closed form
            const double disc = (2.0 * maxV - 1.0) * (2.0 * maxV - 1.0) - 8.0 * (double)pq;
            int r = (int)((2.0 * maxV - 1.0 - sqrt(disc)) * 0.5);
            while (r > 0 && r * (2 * maxV - r - 1) / 2 > pq) --r;
            while ((r + 1) * (2 * maxV - r - 2) / 2 <= pq) ++r;
            const int c = r + 1 + (pq - r * (2 * maxV - r - 1) / 2);
            if (c >= V) continue;
            const int s = st.order[i];
            const int cr = ws.sel_cid[i * C + r], cc = ws.sel_cid[i * C + c];
            const double dsym = epi_sym(cs, cr, st.p2d_pose + ((size_t)s * C + cr) * J3 + j * 3,
                                        cc, st.p2d_pose + ((size_t)s * C + cc) * J3 + j * 3);
            if (1.0 - dsym / prm.joint_threshold < 0.0) atomicOr(&ws.conf[((size_t)i * J + j) * C + r], 1u << c);
        }
        for (int it = it0; it < nT * J * C; it += itS) {
            const int i = it / (J * C), r2 = it % (J * C), j = r2 / C, r = r2 % C;
            if (r >= ws.sel_n[i]) continue;
            const int s = st.order[i], cr = ws.sel_cid[i * C + r];
            const double* pr = st.p2d_pose + ((size_t)s * C + cr) * J3 + j * 3;
            ws.rayd[((size_t)i * J + j) * C + r] = ray_point_dist(cs.RKINV + cr * 9, cs.pos + cr * 3, pr[1], pr[0], ws.pred + i * J3 + j * 3);
        }
    }
    if (PART == 2) {
        if (blockIdx.x == 0 && tid == 0) out_d[13] = now_s();
        return;
    }
    __syncthreads();

    if (tid == 0) out_d[5] = now_s();
    // ---- P4c: greedy filter (matching.py:243-285) + weighted DLT (construction.py:89-114) per (track, joint); tracks that
    //           see many views split each joint's row folding over 4 lanes and merge the triangular factors ------------
    const int nsplit = maxV > 8 ? 4 : 1;
    for (int it = tid; it < nT * J * nsplit; it += NT) {
        const int ij = it / nsplit, q = it % nsplit, i = ij / J, j = ij % J;
        const int V = ws.sel_n[i];
        if (V == 0) continue;
        const int s = st.order[i];
        uint32_t km;
        if (nsplit == 1 || q == 0) {
            km = greedy_keep_update(V, ws.conf + ((size_t)i * J + j) * C, ws.rayd + ((size_t)i * J + j) * C);
            ws.keep[i * J + j] = km; ws.nview[i * J + j] = __popc(km);
        }
        if (nsplit == 1) {
            double X[3];
            if (__popc(km) >= 2) {
                const int* scid = ws.sel_cid + i * C;
                dlt_joint(cs, V, scid, ws.sel_T + i * C, prm.w_lambda_t, prm.lambda_t, km,
                          [&](int v) { return st.p2d_pose + ((size_t)s * C + scid[v]) * J3 + j * 3; }, X);
            } else {
                X[0] = ws.pred[i * J3 + j * 3]; X[1] = ws.pred[i * J3 + j * 3 + 1]; X[2] = ws.pred[i * J3 + j * 3 + 2];
            }
            ws.raw3d[i * J3 + j * 3] = X[0]; ws.raw3d[i * J3 + j * 3 + 1] = X[1]; ws.raw3d[i * J3 + j * 3 + 2] = X[2];
        }
    }
    if (nsplit > 1) {
        __syncthreads();
        for (int it = tid; it < nT * J * nsplit; it += NT) {
            const int ij = it / nsplit, q = it % nsplit, i = ij / J, j = ij % J;
            const int V = ws.sel_n[i];
            if (V == 0) continue;
            const int s = st.order[i];
            const int* scid = ws.sel_cid + i * C;
            double R[4][4];
            dlt_zero(R);
            dlt_fold(cs, V, q, nsplit, scid, ws.sel_T + i * C, prm.w_lambda_t, prm.lambda_t, ws.keep[i * J + j],
                     [&](int v) { return st.p2d_pose + ((size_t)s * C + scid[v]) * J3 + j * 3; }, R);
            dlt_pack(R, ws.rpart + ((size_t)ij * 4 + q) * 10);
        }
        __syncthreads();
        for (int it = tid; it < nT * J; it += NT) {
            const int i = it / J, j = it % J;
            if (ws.sel_n[i] == 0) continue;
            double X[3];
            if (ws.nview[i * J + j] >= 2) {
                double R[4][4];
                dlt_zero(R);
                for (int q = 0; q < 4; ++q) dlt_merge(R, ws.rpart + ((size_t)it * 4 + q) * 10);
                dlt_solve(R, X);
            } else {
                X[0] = ws.pred[i * J3 + j * 3]; X[1] = ws.pred[i * J3 + j * 3 + 1]; X[2] = ws.pred[i * J3 + j * 3 + 2];
            }
            ws.raw3d[i * J3 + j * 3] = X[0]; ws.raw3d[i * J3 + j * 3 + 1] = X[1]; ws.raw3d[i * J3 + j * 3 + 2] = X[2];
        }
    }
    __syncthreads();

    if (tid == 0) out_d[6] = now_s();
    // ---- P4d: success test (:369) ------------------------------------------------------------------------------
    for (int i = tid; i < nT; i += NT) {
        int okv = 0;
        if (ws.sel_n[i] > 0) {
            int fails = 0;
            for (int j = 0; j < J; ++j) fails += (ws.nview[i * J + j] < 2);
            okv = !(3 * fails > J);
        }
        ws.ok[i] = okv;
    }
    __syncthreads();

    if (tid == 0) out_d[7] = now_s();
    // ---- P4e: temporal smoothing into the next ring entry (:371-383) -------------------------------------------
    for (int it = tid; it < nT * J3; it += NT) {
        const int i = it / J3, e = it % J3;
        if (!ws.ok[i]) continue;
        const int s = st.order[i], j = e / 3;
        const int head = st.h_head[s], len = st.h_len[s];
        const bool arm = (j == 9 || j == 10);
        const double* hs = st.hist + (size_t)s * HCAP * J3;
        const double raw = ws.raw3d[i * J3 + e];
        const double val = smooth_last(len + 1, arm ? prm.n_taps_arm : prm.n_taps_body, arm ? prm.taps_arm : prm.taps_body,
                                       [&](int q) { return q < len ? hs[((head + q) % HCAP) * J3 + e] : raw; });
        st.hist[((size_t)s * HCAP + (head + len) % HCAP) * J3 + e] = val;
    }
    __syncthreads();
    if (tid == 0) out_d[8] = now_s();
    // ---- P4f: append + prune history (:330-332) ---------------------------------------------------------------
    for (int i = tid; i < nT; i += NT) {
        if (!ws.ok[i]) continue;
        const int s = st.order[i];
        int head = st.h_head[s], len = st.h_len[s];
        st.hist_time[s * HCAP + (head + len) % HCAP] = frame;
        ++len;
        for (int j = 0; j < J; ++j) st.jv_count[s * J + j] = ws.nview[i * J + j];
        st.jv_V[s] = ws.sel_n[i];
        if (frame - st.hist_time[s * HCAP + head] > prm.max_age) { head = (head + 1) % HCAP; --len; }
        st.h_head[s] = head; st.h_len[s] = len;
    }
    __syncthreads();
    if (tid == 0) out_d[9] = now_s();
    // ---- P4g: float32 velocity (:385-395) and life cycle (:253-274) -------------------------------------------
    for (int it = tid; it < nT * J3; it += NT) {
        const int i = it / J3, e = it % J3;
        if (!ws.ok[i]) continue;
        const int s = st.order[i];
        const int head = st.h_head[s], len = st.h_len[s];
        if (len < 2) continue;
        const double* hs = st.hist + (size_t)s * HCAP * J3;
        st.vel[s * J3 + e] = velocity_f32(len, [&](int q) { return hs[((head + q) % HCAP) * J3 + e]; });
    }
    for (int i = tid; i < nT; i += NT) {
        const int s = st.order[i];
        if (ws.ok[i]) {
            st.hits[s] += 1; st.tsu[s] = 0;
            if (st.state[s] == TENTATIVE && st.hits[s] >= prm.n_init) st.state[s] = CONFIRMED;
        } else {
            if (st.state[s] == TENTATIVE && !st.already[s]) st.state[s] = DELETED;
            else if (st.tsu[s] >= prm.max_age) st.state[s] = DELETED;
        }
    }
    __syncthreads();
    if (tid == 0) out_d[2] = now_s();

    // ---- P5: init_target_GD (:52-113): greedy cross-view hypotheses from the unmatched detections ------------
    if (tid == 0) ws.misc[0] = 0;
    __syncthreads();
    // (steady state: every detection was matched -- no hypothesis can form, and the block below is 2 C + 5 workgroup barriers: 62 on the
    //  31-camera rig, 9 of its 143 us)
    int um_total = 0;
    for (int v = 0; v < C; ++v) um_total += ws.um_n[v];
    if (C >= 2 && um_total > 0) {
        if (tid == 0) {
            int nh = 0;
            for (int k = 0; k < ws.um_n[0]; ++k) {
                if (nh >= MAXH) { atomicOr(&st.hdr[2], ST_HYP_OVERFLOW); break; }
                ws.hyp_size[nh] = 1; ws.hyp_view[nh * C] = 0; ws.hyp_det[nh * C] = ws.um_idx[k]; ++nh;
            }
            ws.misc[0] = nh;
        }
        __syncthreads();
        for (int v = 1; v < C; ++v) {
            const int nh = ws.misc[0], nd = ws.um_n[v];
            if (nh > 0 && nd > 0) {
                for (int it = tid; it < nh * nd; it += NT) {   // Hypothesis.calculate_cost (hypothesis.py:53-68)
                    const int h = it / nd, k = it % nd;
                    const double* po = DET(v, ws.um_idx[v * MAXP + k]);
                    const double bel = believe(po);
                    double total = 0.0; int veto = 0;
                    const int sz = ws.hyp_size[h];
                    for (int q = 0; q < sz; ++q) {
                        const int cm = ws.hyp_view[h * C + q];
                        const double p = hyp_member_cost(cs, cm, DET(cm, ws.hyp_det[h * C + q]), v, po, prm.epi_threshold);
                        total += p;
                        if (p > 1.0 && bel > 0.5) veto = 1;
                    }
                    ws.hc_cost[h * MAXP + k] = total / (double)sz;
                    ws.hc_veto[h * MAXP + k] = (unsigned char)veto;
                }
            }
            __syncthreads();
            if (tid == 0 && nd > 0) {
                int np = 0;
                if (nh > 0) {
                    LsapScratch sc = lsap_carve(ws.lsap2, d.N2);
                    np = lsap_solve(nh, nd, ws.hc_cost, MAXP, 1.0, sc, ws.l2_rows, ws.l2_cols);
                    if (np < 0) { atomicOr(&st.hdr[2], ST_LSAP_INFEASIBLE); np = 0; }
                }
                uint32_t handled = 0;
                int nh2 = nh;
                for (int q = 0; q < np; ++q) {
                    const int h = ws.l2_rows[q], k = ws.l2_cols[q];
                    handled |= 1u << k;
                    if (ws.hc_veto[h * MAXP + k]) {
                        if (nh2 >= MAXH) { atomicOr(&st.hdr[2], ST_HYP_OVERFLOW); continue; }
                        ws.hyp_size[nh2] = 1; ws.hyp_view[nh2 * C] = v; ws.hyp_det[nh2 * C] = ws.um_idx[v * MAXP + k]; ++nh2;
                    } else {
                        const int sz = ws.hyp_size[h];
                        ws.hyp_view[h * C + sz] = v; ws.hyp_det[h * C + sz] = ws.um_idx[v * MAXP + k]; ws.hyp_size[h] = sz + 1;
                    }
                }
                for (int k = 0; k < nd; ++k)
                    if (!((handled >> k) & 1u)) {
                        if (nh2 >= MAXH) { atomicOr(&st.hdr[2], ST_HYP_OVERFLOW); continue; }
                        ws.hyp_size[nh2] = 1; ws.hyp_view[nh2 * C] = v; ws.hyp_det[nh2 * C] = ws.um_idx[v * MAXP + k]; ++nh2;
                    }
                ws.misc[0] = nh2;
            }
            __syncthreads();
        }
        // -- Hypothesis.get_3dpose_jf (hypothesis.py:23-44): float32 loop-form distances, init-mode filter, DLT ----
        const int nh = ws.misc[0];
        const float thr32 = (float)prm.init_threshold;
        for (int it = tid; it < nh * J * C; it += NT) {
            const int h = it / (J * C), r2 = it % (J * C), j = r2 / C, r = r2 % C;
            const int V = ws.hyp_size[h];
            if (V < 2 || r >= V) continue;
            const int* hv = ws.hyp_view + h * C; const int* hd = ws.hyp_det + h * C;
            uint32_t bits = 0;
            const float sum = np_sum_f32(V, [&](int c) -> float {
                if (c == r) return 1.0f - 0.0f / thr32;
                const int lo = c < r ? c : r, hi = c < r ? r : c;
                const float d32 = epi_sym_init(cs, hv[lo], DET(hv[lo], hd[lo]) + j * 3, hv[hi], DET(hv[hi], hd[hi]) + j * 3);
                const float a = 1.0f - d32 / thr32;
                if (c > r && a < 0.0f) bits |= 1u << c;
                return a;
            });
            ws.h_conf[((size_t)h * J + j) * C + r] = bits;
            ws.h_sum[((size_t)h * J + j) * C + r] = sum;
        }
        __syncthreads();
        for (int it = tid; it < nh * J; it += NT) {
            const int h = it / J, j = it % J;
            const int V = ws.hyp_size[h];
            if (V < 2) continue;
            const uint32_t km = greedy_keep_init(V, ws.h_conf + ((size_t)h * J + j) * C, ws.h_sum + ((size_t)h * J + j) * C);
            ws.h_keep[h * J + j] = km;
            if (__popc(km) >= 2) {
                const int* hv = ws.hyp_view + h * C; const int* hd = ws.hyp_det + h * C;
                double X[3];   // ages are all 0 at initialisation (hypothesis.py:24-27)
                dlt_joint(cs, V, hv, g_zeroT, prm.w_lambda_t, prm.lambda_t, km,
                          [&](int v) { return DET(hv[v], hd[v]) + j * 3; }, X);
                ws.h_pose[h * J3 + j * 3] = X[0]; ws.h_pose[h * J3 + j * 3 + 1] = X[1]; ws.h_pose[h * J3 + j * 3 + 2] = X[2];
            }
        }
        __syncthreads();
        if (tid == 0) {   // spawn tracks in hypothesis order (:102-113)
            int n = st.hdr[0];
            unsigned long long used = ((unsigned long long)(unsigned)st.hdr[4] << 32) | (unsigned)st.hdr[3];
            for (int h = 0; h < nh; ++h) {
                ws.h_slot[h] = -1;
                const int V = ws.hyp_size[h];
                if (V < 2) continue;
                bool good = true;
                for (int j = 0; j < J; ++j) if (__popc(ws.h_keep[h * J + j]) < 2) { good = false; break; }
                if (!good) continue;
                int s = -1;
                for (int q = 0; q < MAXT; ++q) if (!((used >> q) & 1ull)) { s = q; break; }
                if (s < 0) { atomicOr(&st.hdr[2], ST_TRACK_OVERFLOW); continue; }
                used |= 1ull << s;
                st.track_id[s] = st.hdr[1]; st.hdr[1] += 1;
                st.hits[s] = 1; st.age[s] = 1; st.tsu[s] = 0; st.already[s] = 0; st.state[s] = TENTATIVE;
                st.p2d_n[s] = V; st.h_head[s] = 0; st.h_len[s] = 1; st.hist_time[s * HCAP] = frame; st.jv_V[s] = V;
                for (int c = 0; c < C; ++c) { st.p2d_time[s * C + c] = T_NONE; st.cur_det[s * C + c] = -1; }
                for (int q = 0; q < V; ++q) {
                    const int cid = ws.hyp_view[h * C + q];
                    st.p2d_order[s * C + q] = cid; st.p2d_time[s * C + cid] = frame; st.cur_det[s * C + cid] = ws.hyp_det[h * C + q];
                }
                for (int j = 0; j < J; ++j) st.jv_count[s * J + j] = __popc(ws.h_keep[h * J + j]);
                st.order[n++] = s;
                ws.h_slot[h] = s;
            }
            st.hdr[0] = n; st.hdr[3] = (int)(unsigned)(used & 0xffffffffull); st.hdr[4] = (int)(unsigned)(used >> 32);
        }
        __syncthreads();
        for (int it = tid; it < nh * (C + 1) * J3; it += NT) {   // pose copies of the new tracks
            const int h = it / ((C + 1) * J3), r = it % ((C + 1) * J3), q = r / J3, e = r % J3;
            const int s = ws.h_slot[h];
            if (s < 0) continue;
            const int V = ws.hyp_size[h];
            if (q < V) {
                const int cid = ws.hyp_view[h * C + q];
                st.p2d_pose[((size_t)s * C + cid) * J3 + e] = DET(cid, ws.hyp_det[h * C + q])[e];
            } else if (q == C) {
                st.hist[(size_t)s * HCAP * J3 + e] = ws.h_pose[h * J3 + e];
                st.vel[s * J3 + e] = 0.0f;
            }
        }
        __syncthreads();
    }
    if (tid == 0) out_d[3] = now_s();

    if (tid == 0) out_d[10] = now_s();
    // ---- P6: drop deleted tracks, keep list order (:178) ------------------------------------------------------
    if (tid == 0) {
        const int n = st.hdr[0];
        unsigned long long used = ((unsigned long long)(unsigned)st.hdr[4] << 32) | (unsigned)st.hdr[3];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int s = st.order[i];
            if (st.state[s] == DELETED) used &= ~(1ull << s); else st.order[m++] = s;
        }
        st.hdr[0] = m; st.hdr[3] = (int)(unsigned)(used & 0xffffffffull); st.hdr[4] = (int)(unsigned)(used >> 32);
        // status word: bits 0-15 = this frame, bits 16-31 = OR over every frame since pam_create / pam_reset (a host that
        // only decodes the last record of a run still sees an overflow raised in any earlier frame)
        st.hdr[5] |= st.hdr[2];
        st.hdr[6] = 0;                                  // a frame was applied: the run of void frames (input guard above), if any, is over
        out_i[0] = m; out_i[1] = (st.hdr[2] & 0xffff) | (st.hdr[5] << 16); out_i[2] = frame; out_i[3] = ws.misc[0];
    }
    __syncthreads();

    // ---- P7: output record (ivclabpose.py:259-287 + state for parity tests) -----------------------------------
    const int nOut = st.hdr[0];
    const PamOutLayout ol = A.ol;
    for (int i = tid; i < nOut; i += NT) {
        const int s = st.order[i];
        int* o = out_i + ol.hdr_words + i * ol.trk_words;
        const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
        o[0] = st.track_id[s]; o[1] = st.state[s]; o[2] = st.hits[s]; o[3] = st.age[s]; o[4] = st.tsu[s];
        o[5] = (st.tsu[s] == 0 && st.state[s] == CONFIRMED) ? 1 : 0;
        o[6] = st.p2d_n[s]; o[7] = st.jv_V[s]; o[8] = st.h_len[s]; o[9] = st.hist_time[s * HCAP + newest];
    }
    for (int it = tid; it < nOut * (C + J); it += NT) {    // the per-view and per-joint fields: a lane per (track, field), not a lane per track
        const int i = it / (C + J), c = it % (C + J), s = st.order[i];
        int* o = out_i + ol.hdr_words + i * ol.trk_words;
        if (c < C) {
            o[ol.off_order + c] = (c < st.p2d_n[s]) ? st.p2d_order[s * C + c] : -1;
            o[ol.off_matched + c] = st.cur_det[s * C + c];
            o[ol.off_time2d + c] = st.p2d_time[s * C + c];
        } else {
            o[ol.off_nviews + (c - C)] = st.jv_count[s * J + (c - C)];
        }
    }
    for (int it = tid; it < nOut * J3; it += NT) {
        const int i = it / J3, e = it % J3, s = st.order[i];
        const int newest = (st.h_head[s] + st.h_len[s] - 1) % HCAP;
        double* o = out_d + ol.dbl_hdr_words + (size_t)i * ol.dbl_trk_words;
        o[e] = st.hist[((size_t)s * HCAP + newest) * J3 + e];
        o[J3 + e] = (double)st.vel[s * J3 + e];
    }
    if (tid == 0) out_d[11] = now_s();
    if (A.hot_in_lds) {                                                 // integer scene state back to HBM
        __syncthreads();
        const char* st_lds = hot_lds + hot_bytes(d);
        for (size_t o = (size_t)threadIdx.x * 4; o < st_ints; o += (size_t)blockDim.x * 4) *(int*)(st_glob + o) = *(const int*)(st_lds + o);
    }
#undef DET
#undef NDET
#undef NDET_RAW
}

// =====================================================================================================================
// Per-operator kernels (parity tests): thin wrappers over the same device functions.
// =====================================================================================================================
__global__ void k_op_project(CamSet cs, int cid, int n, const double* poses, double* out) {
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n * J; it += gridDim.x * blockDim.x) {
        double u, v;
        project_point(cs.P + cid * 12, poses[it * 3], poses[it * 3 + 1], poses[it * 3 + 2], u, v);
        out[it * 2] = v; out[it * 2 + 1] = u;
    }
}
__global__ void k_op_affinity(CamSet cs, const PamParams* prm, int cid, int n, int m, const double* tp, const int* dt,
                              const double* dets, double* aff) {
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n * m; it += gridDim.x * blockDim.x) {
        const int i = it / m, k = it % m, t = dt[i];
        const double e = (t >= 0 && t < PAM_EXP_TABLE) ? prm->exp_lambda_a[t] : exp(prm->lambda_a * (double)t);
        aff[it] = track_det_affinity(cs.P + cid * 12, tp + (size_t)i * J3, dets + (size_t)k * J3, prm->alpha2d * (double)t, e,
                                     prm->count_gate);
    }
}
__global__ void k_op_lsap(int nr, int nc, const double* cost, char* scratch, int N, int* rows, int* cols, int* np) {
    // one wave: the solver the frame kernel runs per view (N <= 64); beyond that the single-lane form the hypothesis step uses
    if (N <= 64) {
        int row, col;
        const int n = lsap_solve_wave(nr, nc, cost, nc, 1.0, rows, cols, row, col);
        if (threadIdx.x == 0) *np = n;
    } else if (threadIdx.x == 0) {
        LsapScratch sc = lsap_carve(scratch, N);
        *np = lsap_solve(nr, nc, cost, nc, 1.0, sc, rows, cols);
    }
}
__global__ void k_op_epi_dist(CamSet cs, int V, const int* cids, const double* pm, double* dist) {
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < V * V * J; it += gridDim.x * blockDim.x) {
        const int a = it / (V * J), r = it % (V * J), b = r / J, j = r % J;
        dist[it] = epi_sym(cs, cids[a], pm + ((size_t)a * J + j) * 3, cids[b], pm + ((size_t)b * J + j) * 3);
    }
}
__global__ void k_op_epi_pair(CamSet cs, int c1, const double* p1, int c2, const double* p2, double* out) {
    const int j = threadIdx.x;
    if (j < J) {
        double d1, d2;
        epi_pair_cv(cs.F + ((size_t)c1 * cs.C + c2) * 9, p1[j * 3 + 1], p1[j * 3], p2[j * 3 + 1], p2[j * 3], d1, d2);
        out[j * 2] = d1; out[j * 2 + 1] = d2;
    }
}
__global__ void k_op_epi_dist_init(CamSet cs, int V, const int* cids, const double* pm, float* dist) {
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < V * V * J; it += gridDim.x * blockDim.x) {
        const int a = it / (V * J), r = it % (V * J), b = r / J, j = r % J;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        dist[it] = (a == b) ? 0.0f : epi_sym_init(cs, cids[lo], pm + ((size_t)lo * J + j) * 3, cids[hi], pm + ((size_t)hi * J + j) * 3);
    }
}
__global__ void k_op_greedy(CamSet cs, int mode, int V, const int* cids, const void* aff, const double* pose_j,
                            const double* next_j, uint32_t* conf, double* ray, float* sums, uint32_t* keep) {
    const int r = threadIdx.x;
    if (r < V) {
        uint32_t bits = 0;
        if (mode == 0) {
            const double* a = (const double*)aff;
            for (int c = r + 1; c < V; ++c) if (a[r * V + c] < 0.0) bits |= 1u << c;
            ray[r] = ray_point_dist(cs.RKINV + cids[r] * 9, cs.pos + cids[r] * 3, pose_j[r * 3 + 1], pose_j[r * 3], next_j);
        } else {
            const float* a = (const float*)aff;
            for (int c = r + 1; c < V; ++c) if (a[r * V + c] < 0.0f) bits |= 1u << c;
            sums[r] = np_sum_f32(V, [&](int c) { return a[r * V + c]; });
        }
        conf[r] = bits;
    }
    __syncthreads();
    if (threadIdx.x == 0) *keep = (mode == 0) ? greedy_keep_update(V, conf, ray) : greedy_keep_init(V, conf, sums);
}
__global__ void k_op_dlt(CamSet cs, const PamParams* prm, int V, const int* cids, const int* Ts, const double* pm,
                         const uint32_t* keep, const double* next_pose, double* out) {
    const int j = threadIdx.x;
    if (j < J) {
        double X[3];
        if (__popc(keep[j]) >= 2)
            dlt_joint(cs, V, cids, Ts, prm->w_lambda_t, prm->lambda_t, keep[j],
                      [&](int v) { return pm + ((size_t)v * J + j) * 3; }, X);
        else { X[0] = next_pose[j * 3]; X[1] = next_pose[j * 3 + 1]; X[2] = next_pose[j * 3 + 2]; }
        out[j * 3] = X[0]; out[j * 3 + 1] = X[1]; out[j * 3 + 2] = X[2];
    }
}
__global__ void k_op_smooth(const PamParams* prm, int L, const double* hist, const double* raw, double* out) {
    const int e = threadIdx.x;
    if (e < J3) {
        const int j = e / 3;
        const bool arm = (j == 9 || j == 10);
        out[e] = smooth_last(L + 1, arm ? prm->n_taps_arm : prm->n_taps_body, arm ? prm->taps_arm : prm->taps_body,
                             [&](int q) { return q < L ? hist[(size_t)q * J3 + e] : raw[e]; });
    }
}
__global__ void k_op_velocity(int L, const double* hist, float* vel) {
    const int e = threadIdx.x;
    if (e < J3) vel[e] = velocity_f32(L, [&](int q) { return hist[(size_t)q * J3 + e]; });
}
__global__ void k_op_hyp_cost(CamSet cs, const PamParams* prm, int n, const int* cids, const double* poses, int o_cid,
                              const double* o_pose, double* cost, int* veto) {
    if (threadIdx.x == 0) {
        const double bel = believe(o_pose);
        double total = 0.0; int vt = 0;
        for (int q = 0; q < n; ++q) {
            const double p = hyp_member_cost(cs, cids[q], poses + (size_t)q * J3, o_cid, o_pose, prm->epi_threshold);
            total += p;
            if (p > 1.0 && bel > 0.5) vt = 1;
        }
        *cost = total / (double)n; *veto = vt;
    }
}

// =====================================================================================================================
// Host side of the C ABI
// =====================================================================================================================
struct PamHandle {
    int device = 0;
    Dims d{};
    PamParams prm{};
    PamParams* d_prm = nullptr;
    float *d_P = nullptr, *d_F = nullptr, *d_RK = nullptr; double* d_pos = nullptr;
    bool cams_set = false;
    char* d_state = nullptr; size_t state_stride = 0;
    char* d_ws = nullptr; size_t ws_stride = 0;
    int* d_ndet = nullptr; double* d_det = nullptr;
    int* d_out_i = nullptr; double* d_out_d = nullptr;
    PamOutLayout ol{};
    hipStream_t stream = nullptr;
    char* d_op = nullptr; size_t op_bytes = 0;     // staging for the per-operator entry points
    const int* d_guard = nullptr;                  // pam_set_input_guard
    std::string err;
};
static std::string g_create_err;

#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return PAM_E_HIP; } } while (0)
#define ARGCHK(h, cond, msg) do { if (!(cond)) { (h)->err = msg; return PAM_E_ARG; } } while (0)

static CamSet camset(const PamHandle* h) { return CamSet{h->d_P, h->d_F, h->d_RK, h->d_pos, h->d.C}; }

extern "C" const char* pam_version(void) { return "pam-hip 0.1 (gfx950)"; }
extern "C" const char* pam_last_error(const PamHandle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int pam_create(PamHandle** out, int device, int n_views, int max_dets, int max_tracks, int max_hyps,
                          int n_scenes, const PamParams* params) {
    if (!out || !params) { g_create_err = "null argument"; return PAM_E_ARG; }
    if (n_views < 1 || n_views > PAM_MAX_VIEWS || max_dets < 1 || max_dets > 32 || max_tracks < 1 || max_tracks > 64 ||
        n_scenes < 1 || params->max_age < 1 || params->max_age > 62 || params->n_taps_body < 1 ||
        params->n_taps_body > PAM_MAX_TAPS || params->n_taps_arm < 1 || params->n_taps_arm > PAM_MAX_TAPS) {
        g_create_err = "capacity out of range (views<=32, dets<=32, tracks<=64, max_age<=62, taps<=16)";
        return PAM_E_ARG;
    }
    PamHandle* h = new PamHandle();
    h->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); delete h; return PAM_E_HIP; }
    Dims& d = h->d;
    d.C = n_views; d.MAXP = max_dets; d.MAXT = max_tracks; d.HCAP = params->max_age + 2;
    d.MAXH = max_hyps > 0 ? max_hyps : 4 * max_dets;
    if (d.MAXH > n_views * max_dets) d.MAXH = n_views * max_dets;
    if (d.MAXH < max_dets) d.MAXH = max_dets;
    d.S = n_scenes;
    d.N1 = max_tracks > max_dets ? max_tracks : max_dets;
    d.N2 = d.MAXH > max_dets ? d.MAXH : max_dets;
    h->prm = *params;
    SceneState st; Scratch ws;
    h->state_stride = ((size_t)(uintptr_t)carve_state((char*)0, d, st) + 255) & ~(size_t)255;
    h->ws_stride = ((size_t)(uintptr_t)carve_ws((char*)0, d, ws) + 255) & ~(size_t)255;
    PamOutLayout& ol = h->ol;
    ol.n_views = d.C; ol.max_dets = d.MAXP; ol.max_tracks = d.MAXT; ol.n_scenes = d.S;
    ol.hdr_words = 4;
    ol.off_order = 10; ol.off_matched = 10 + d.C; ol.off_time2d = 10 + 2 * d.C; ol.off_nviews = 10 + 3 * d.C;
    ol.trk_words = 10 + 3 * d.C + J;
    ol.int_words = ol.hdr_words + d.MAXT * ol.trk_words;
    ol.dbl_hdr_words = 16; ol.dbl_trk_words = 2 * J3;
    ol.dbl_words = ol.dbl_hdr_words + d.MAXT * ol.dbl_trk_words;
    h->op_bytes = 8u << 20;
#define CR(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(e_); \
    pam_destroy(h); return PAM_E_HIP; } } while (0)
    CR(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CR(hipMalloc(&h->d_prm, sizeof(PamParams)));
    CR(hipMemcpy(h->d_prm, params, sizeof(PamParams), hipMemcpyHostToDevice));
    CR(hipMalloc(&h->d_P, sizeof(float) * d.C * 12));
    CR(hipMalloc(&h->d_F, sizeof(float) * d.C * d.C * 9));
    CR(hipMalloc(&h->d_RK, sizeof(float) * d.C * 9));
    CR(hipMalloc(&h->d_pos, sizeof(double) * d.C * 3));
    CR(hipMalloc(&h->d_state, h->state_stride * d.S));
    CR(hipMemset(h->d_state, 0, h->state_stride * d.S));
    CR(hipMalloc(&h->d_ws, h->ws_stride * d.S));
    CR(hipMemset(h->d_ws, 0, h->ws_stride * d.S));
    CR(hipMalloc(&h->d_ndet, sizeof(int) * d.S * d.C));
    CR(hipMalloc(&h->d_det, sizeof(double) * (size_t)d.S * d.C * d.MAXP * J3));
    {   // the record's two sections in ONE allocation, the float64 section behind the int32 one (padded to 8 bytes): a host that lays its
        // pinned landing buffer out the same way gets the record in ONE copy (pam_fetch; Handle.pinned_record)
        const size_t ib = (sizeof(int) * (size_t)d.S * ol.int_words + 7) & ~(size_t)7, db = sizeof(double) * (size_t)d.S * ol.dbl_words;
        char* rec = nullptr;
        CR(hipMalloc(&rec, ib + db));
        CR(hipMemset(rec, 0, ib + db));
        h->d_out_i = (int*)rec; h->d_out_d = (double*)(rec + ib);
    }
    CR(hipMalloc(&h->d_op, h->op_bytes));
#undef CR
    *out = h;
    return PAM_OK;
}

extern "C" int pam_destroy(PamHandle* h) {
    if (!h) return PAM_OK;
    hipSetDevice(h->device);
    hipFree(h->d_prm); hipFree(h->d_P); hipFree(h->d_F); hipFree(h->d_RK); hipFree(h->d_pos);
    hipFree(h->d_state); hipFree(h->d_ws); hipFree(h->d_ndet); hipFree(h->d_det);
    hipFree(h->d_out_i); hipFree(h->d_op);                  // (d_out_d lives in d_out_i's allocation)
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return PAM_OK;
}

extern "C" int pam_set_cameras(PamHandle* h, const float* P, const float* F, const float* RK_INV, const double* position) {
    if (!h) return PAM_E_ARG;
    ARGCHK(h, P && F && RK_INV && position, "null camera array");
    const Dims& d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->d_P, P, sizeof(float) * d.C * 12, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_F, F, sizeof(float) * d.C * d.C * 9, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_RK, RK_INV, sizeof(float) * d.C * 9, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_pos, position, sizeof(double) * d.C * 3, hipMemcpyHostToDevice));
    h->cams_set = true;
    return PAM_OK;
}

extern "C" int pam_reset(PamHandle* h) {
    if (!h) return PAM_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemset(h->d_state, 0, h->state_stride * h->d.S));
    return PAM_OK;
}

extern "C" int pam_out_layout(const PamHandle* h, PamOutLayout* out) {
    if (!h || !out) return PAM_E_ARG;
    *out = h->ol;
    return PAM_OK;
}

// The frame kernel skips a frame (record status bit 16, state untouched) while *dev_word != 0: dev_word is raised by whoever produced
// the frame's keypoints when they are not valid (pam_flag_gate's dev_void).  NULL removes the guard.
extern "C" int pam_set_input_guard(PamHandle* h, const int32_t* dev_word) {
    if (!h) return PAM_E_ARG;
    h->d_guard = dev_word;
    return PAM_OK;
}

static int launch_frame(PamHandle* h, hipStream_t s, int frame_id, const int* d_ndet, const double* d_det, const int* d_view_row = nullptr) {
    if (!h->cams_set) { h->err = "pam_set_cameras has not been called"; return PAM_E_STATE; }
    FrameArgs A;
    A.d = h->d; A.cs = camset(h); A.prm = h->d_prm;
    A.state = h->d_state; A.state_stride = h->state_stride; A.ws = h->d_ws; A.ws_stride = h->ws_stride;
    A.guard = h->d_guard;
    A.n_det = d_ndet; A.det = d_det; A.view_row = d_view_row; A.out_i = h->d_out_i; A.out_d = h->d_out_d; A.ol = h->ol; A.frame_id = frame_id;
    const size_t hot = hot_bytes(h->d) + ((state_int_bytes(h->d) + 15) & ~(size_t)15);
    A.hot_in_lds = hot <= 128 * 1024 ? 1 : 0;
    // one workgroup per scene; many-camera rigs have ~10^4-10^5 independent (track, view pair, joint) items per frame, so
    // they get the largest workgroup (16 waves hide the L2 latency of the pose / fundamental-matrix reads)
    const int block = h->d.C > 8 ? 1024 : BLOCK;
    // each workgroup size is its own instantiation with its own __launch_bounds__: under a shared bound of 1024 the 256-thread
    // form was held to 128 VGPRs (occupancy 4) and spilled 211 SGPRs
#ifdef PAM_DIAG
    static const int dblock = getenv("PAM_FRAME_BLOCK") ? atoi(getenv("PAM_FRAME_BLOCK")) : 0;
    static const int dlds = getenv("PAM_FRAME_LDS") ? atoi(getenv("PAM_FRAME_LDS")) : 1;
    if (!dlds) A.hot_in_lds = 0;
    if (getenv("PAM_FRAME_VERBOSE")) fprintf(stderr, "k_frame: hot bytes %zu, block %d\n", hot, dblock ? dblock : block);
    if (dblock == 64) { hipLaunchKernelGGL(k_frame<64>, dim3(h->d.S), dim3(64), A.hot_in_lds ? hot : 0, s, A); HIPCHK(h, hipGetLastError()); return PAM_OK; }
    if (dblock == 128) { hipLaunchKernelGGL(k_frame<128>, dim3(h->d.S), dim3(128), A.hot_in_lds ? hot : 0, s, A); HIPCHK(h, hipGetLastError()); return PAM_OK; }
#endif
    // wide rigs, one scene: three launches, the conflict sets of P4b over the whole chip (PAM_FRAME_SPLIT=0: the single launch, for A/B runs)
    static const int split_ok = getenv("PAM_FRAME_SPLIT") ? atoi(getenv("PAM_FRAME_SPLIT")) : 1;
    if (block == 1024 && h->d.S == 1 && split_ok) {
        // launches 1 and 3 keep the hot scratch and the integer state in LDS (the serial LSAP walks live there) and hand them over through
        // their home in global memory; launch 2 works on the global copy
        const int in_lds = A.hot_in_lds;
        hipLaunchKernelGGL((k_frame<1024, 1>), dim3(1), dim3(1024), in_lds ? hot : 0, s, A);
        A.hot_in_lds = 0;
        hipLaunchKernelGGL((k_frame<256, 2>), dim3(256), dim3(256), 0, s, A);
        A.hot_in_lds = in_lds;
        hipLaunchKernelGGL((k_frame<1024, 3>), dim3(1), dim3(1024), in_lds ? hot : 0, s, A);
    }
    else if (block == 1024) hipLaunchKernelGGL(k_frame<1024>, dim3(h->d.S), dim3(1024), A.hot_in_lds ? hot : 0, s, A);
    else hipLaunchKernelGGL(k_frame<BLOCK>, dim3(h->d.S), dim3(BLOCK), A.hot_in_lds ? hot : 0, s, A);
    HIPCHK(h, hipGetLastError());
    return PAM_OK;
}

extern "C" int pam_frame_dev(PamHandle* h, void* stream, int frame_id, const int32_t* dev_n_det, const double* dev_det) {
    if (!h) return PAM_E_ARG;
    ARGCHK(h, dev_n_det && dev_det, "null device buffer");
    return launch_frame(h, (hipStream_t)stream, frame_id, dev_n_det, dev_det);
}

// The frame on view-sharded input: dev_records = the buffer pam_allgather_keypoints (or any all-gather of the ranks' send buffers)
// filled, dev_view_row[v] = which of its records holds view v.  The kernel reads the records in place: no unpack kernel, no copies.
extern "C" int pam_frame_dev_views(PamHandle* h, void* stream, int frame_id, const double* dev_records, const int32_t* dev_view_row) {
    if (!h) return PAM_E_ARG;
    ARGCHK(h, dev_records && dev_view_row, "null device buffer");
    ARGCHK(h, h->d.S == 1, "view-sharded input is one scene per handle");
    return launch_frame(h, (hipStream_t)stream, frame_id, nullptr, dev_records, dev_view_row);
}

extern "C" int pam_fetch(PamHandle* h, void* stream, int32_t* host_out_i, double* host_out_d) {
    if (!h) return PAM_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t ib = (size_t)((char*)h->d_out_d - (char*)h->d_out_i), db = sizeof(double) * (size_t)h->d.S * h->ol.dbl_words;
    if (host_out_i && host_out_d && (char*)host_out_d == (char*)host_out_i + ib) {                 // one landing buffer in the device layout: one copy
        HIPCHK(h, hipMemcpyAsync(host_out_i, h->d_out_i, ib + db, hipMemcpyDeviceToHost, s));
        return PAM_OK;
    }
    if (host_out_i) HIPCHK(h, hipMemcpyAsync(host_out_i, h->d_out_i, sizeof(int) * (size_t)h->d.S * h->ol.int_words, hipMemcpyDeviceToHost, s));
    if (host_out_d) HIPCHK(h, hipMemcpyAsync(host_out_d, h->d_out_d, db, hipMemcpyDeviceToHost, s));
    return PAM_OK;
}

extern "C" int pam_sync(PamHandle* h, void* stream) {
    if (!h) return PAM_E_ARG;
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    return PAM_OK;
}

extern "C" int pam_frame(PamHandle* h, int frame_id, const int32_t* n_det, const double* det, int32_t* out_i, double* out_d) {
    if (!h) return PAM_E_ARG;
    ARGCHK(h, n_det && det, "null input");
    const Dims& d = h->d;
    for (int i = 0; i < d.S * d.C; ++i) ARGCHK(h, n_det[i] >= 0 && n_det[i] <= d.MAXP, "n_det exceeds max_dets");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_ndet, n_det, sizeof(int) * d.S * d.C, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_det, det, sizeof(double) * (size_t)d.S * d.C * d.MAXP * J3, hipMemcpyHostToDevice, h->stream));
    int rc = launch_frame(h, h->stream, frame_id, h->d_ndet, h->d_det);
    if (rc) return rc;
    if (out_i) HIPCHK(h, hipMemcpyAsync(out_i, h->d_out_i, sizeof(int) * (size_t)d.S * h->ol.int_words, hipMemcpyDeviceToHost, h->stream));
    if (out_d) HIPCHK(h, hipMemcpyAsync(out_d, h->d_out_d, sizeof(double) * (size_t)d.S * h->ol.dbl_words, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out_i)
        for (int s = 0; s < d.S; ++s)
            if (out_i[(size_t)s * h->ol.int_words + 1] & (ST_TRACK_OVERFLOW | ST_HYP_OVERFLOW)) {
                h->err = "scene ran out of track or hypothesis slots (raise max_tracks / max_hyps)";
                return PAM_E_OVERFLOW;
            }
    return PAM_OK;
}

// ---- per-operator entry points: stage to the handle's op buffer, run, copy back -----------------------------------
struct OpStage {
    PamHandle* h; size_t off = 0; bool ok = true;
    explicit OpStage(PamHandle* h_) : h(h_) { hipSetDevice(h->device); }
    template <typename T> T* in(const T* src, size_t n) {
        T* p = out<T>(n);
        if (p && src && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return p;
    }
    template <typename T> T* out(size_t n) {
        off = (off + 255) & ~(size_t)255;
        if (off + n * sizeof(T) > h->op_bytes) { ok = false; return nullptr; }
        T* p = (T*)(h->d_op + off); off += n * sizeof(T);
        return p;
    }
    template <typename T> void back(T* dst, const T* dev, size_t n) {
        if (hipMemcpy(dst, dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    }
    int done(const char* what) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess || !ok) { h->err = std::string(what) + ": " + (ok ? hipGetErrorString(e) : "staging failed (op buffer too small or copy error)"); return ok ? PAM_E_HIP : PAM_E_ARG; }
        return PAM_OK;
    }
};
#define OP_PROLOG(h) if (!(h)) return PAM_E_ARG; if (!(h)->cams_set) { (h)->err = "pam_set_cameras has not been called"; return PAM_E_STATE; } OpStage S(h)
#define OP_CHECK_STAGE(name) if (!S.ok) { h->err = name ": staging failed"; return PAM_E_ARG; }

extern "C" int pam_op_project(PamHandle* h, int cid, int n, const double* poses3d, double* out_yx) {
    OP_PROLOG(h);
    ARGCHK(h, cid >= 0 && cid < h->d.C && n >= 0, "bad cid / n");
    if (n == 0) return PAM_OK;
    const double* p = S.in(poses3d, (size_t)n * J3); double* o = S.out<double>((size_t)n * J * 2);
    OP_CHECK_STAGE("pam_op_project");
    hipLaunchKernelGGL(k_op_project, dim3((n * J + 255) / 256), dim3(256), 0, 0, camset(h), cid, n, p, o);
    int rc = S.done("pam_op_project"); if (rc) return rc;
    S.back(out_yx, o, (size_t)n * J * 2);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_track_affinity(PamHandle* h, int cid, int n, int m, const double* tracks_pose, const int32_t* dt,
                                     const double* dets, double* aff) {
    OP_PROLOG(h);
    ARGCHK(h, cid >= 0 && cid < h->d.C && n > 0 && m > 0, "bad cid / n / m");
    const double* tp = S.in(tracks_pose, (size_t)n * J3); const int* t = S.in(dt, n);
    const double* dd = S.in(dets, (size_t)m * J3); double* o = S.out<double>((size_t)n * m);
    OP_CHECK_STAGE("pam_op_track_affinity");
    hipLaunchKernelGGL(k_op_affinity, dim3((n * m + 63) / 64), dim3(64), 0, 0, camset(h), h->d_prm, cid, n, m, tp, t, dd, o);
    int rc = S.done("pam_op_track_affinity"); if (rc) return rc;
    S.back(aff, o, (size_t)n * m);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_lsap(PamHandle* h, int nr, int nc, const double* cost, int32_t* rows, int32_t* cols, int32_t* n_pairs) {
    if (!h) return PAM_E_ARG;
    OpStage S(h);
    ARGCHK(h, nr >= 0 && nc >= 0 && nr <= 1024 && nc <= 1024, "bad shape");
    if (nr == 0 || nc == 0) { *n_pairs = 0; return PAM_OK; }
    const int N = nr > nc ? nr : nc;
    const double* c = S.in(cost, (size_t)nr * nc);
    char* sc = S.out<char>(lsap_scratch_bytes(N) + 8);
    int* r = S.out<int>(N); int* cc = S.out<int>(N); int* np = S.out<int>(1);
    OP_CHECK_STAGE("pam_op_lsap");
    hipLaunchKernelGGL(k_op_lsap, dim3(1), dim3(64), 0, 0, nr, nc, c, (char*)(((uintptr_t)sc + 7) & ~(uintptr_t)7), N, r, cc, np);
    int rc = S.done("pam_op_lsap"); if (rc) return rc;
    S.back(n_pairs, np, 1);
    if (*n_pairs < 0) { h->err = "infeasible cost matrix"; return PAM_E_ARG; }
    S.back(rows, r, *n_pairs); S.back(cols, cc, *n_pairs);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_epi_dist(PamHandle* h, int V, const int32_t* cids, const double* pose_mat, double* dist) {
    OP_PROLOG(h);
    ARGCHK(h, V >= 1 && V <= PAM_MAX_VIEWS, "bad V");
    const int* c = S.in(cids, V); const double* pm = S.in(pose_mat, (size_t)V * J3); double* o = S.out<double>((size_t)V * V * J);
    OP_CHECK_STAGE("pam_op_epi_dist");
    hipLaunchKernelGGL(k_op_epi_dist, dim3((V * V * J + 255) / 256), dim3(256), 0, 0, camset(h), V, c, pm, o);
    int rc = S.done("pam_op_epi_dist"); if (rc) return rc;
    S.back(dist, o, (size_t)V * V * J);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_epi_pair(PamHandle* h, int c1, const double* person1, int c2, const double* person2, double* out) {
    OP_PROLOG(h);
    ARGCHK(h, c1 >= 0 && c1 < h->d.C && c2 >= 0 && c2 < h->d.C, "bad camera id");
    const double* p1 = S.in(person1, J3); const double* p2 = S.in(person2, J3); double* o = S.out<double>(J * 2);
    OP_CHECK_STAGE("pam_op_epi_pair");
    hipLaunchKernelGGL(k_op_epi_pair, dim3(1), dim3(64), 0, 0, camset(h), c1, p1, c2, p2, o);
    int rc = S.done("pam_op_epi_pair"); if (rc) return rc;
    S.back(out, o, J * 2);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_epi_dist_init(PamHandle* h, int V, const int32_t* cids, const double* pose_mat, float* dist) {
    OP_PROLOG(h);
    ARGCHK(h, V >= 1 && V <= PAM_MAX_VIEWS, "bad V");
    const int* c = S.in(cids, V); const double* pm = S.in(pose_mat, (size_t)V * J3); float* o = S.out<float>((size_t)V * V * J);
    OP_CHECK_STAGE("pam_op_epi_dist_init");
    hipLaunchKernelGGL(k_op_epi_dist_init, dim3((V * V * J + 255) / 256), dim3(256), 0, 0, camset(h), V, c, pm, o);
    int rc = S.done("pam_op_epi_dist_init"); if (rc) return rc;
    S.back(dist, o, (size_t)V * V * J);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_greedy(PamHandle* h, int mode, int V, const int32_t* cids, const void* aff, const double* pose_j,
                             const double* next_pose_j, uint32_t* keep_mask) {
    OP_PROLOG(h);
    ARGCHK(h, V >= 1 && V <= PAM_MAX_VIEWS && (mode == 0 || mode == 1), "bad V / mode");
    const int* c = S.in(cids, V);
    const void* a = (mode == 0) ? (const void*)S.in((const double*)aff, (size_t)V * V) : (const void*)S.in((const float*)aff, (size_t)V * V);
    const double* pj = (mode == 0) ? S.in(pose_j, (size_t)V * 3) : nullptr;
    const double* nj = (mode == 0) ? S.in(next_pose_j, 3) : nullptr;
    uint32_t* conf = S.out<uint32_t>(V); double* ray = S.out<double>(V); float* sums = S.out<float>(V); uint32_t* k = S.out<uint32_t>(1);
    OP_CHECK_STAGE("pam_op_greedy");
    hipLaunchKernelGGL(k_op_greedy, dim3(1), dim3(64), 0, 0, camset(h), mode, V, c, a, pj, nj, conf, ray, sums, k);
    int rc = S.done("pam_op_greedy"); if (rc) return rc;
    S.back(keep_mask, k, 1);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_dlt(PamHandle* h, int V, const int32_t* cids, const int32_t* Ts, const double* pose_mat,
                          const uint32_t* keep_mask, const double* next_pose, double* out) {
    OP_PROLOG(h);
    ARGCHK(h, V >= 1 && V <= PAM_MAX_VIEWS, "bad V");
    const int* c = S.in(cids, V); const int* t = S.in(Ts, V); const double* pm = S.in(pose_mat, (size_t)V * J3);
    const uint32_t* k = S.in(keep_mask, J); const double* np = S.in(next_pose, J3); double* o = S.out<double>(J3);
    OP_CHECK_STAGE("pam_op_dlt");
    hipLaunchKernelGGL(k_op_dlt, dim3(1), dim3(64), 0, 0, camset(h), h->d_prm, V, c, t, pm, k, np, o);
    int rc = S.done("pam_op_dlt"); if (rc) return rc;
    S.back(out, o, J3);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_smooth(PamHandle* h, int L, const double* hist, const double* raw, double* out) {
    if (!h) return PAM_E_ARG;
    OpStage S(h);
    ARGCHK(h, L >= 0 && L <= 4096, "bad L");
    const double* hs = S.in(hist, (size_t)(L > 0 ? L : 1) * J3); const double* r = S.in(raw, J3); double* o = S.out<double>(J3);
    OP_CHECK_STAGE("pam_op_smooth");
    hipLaunchKernelGGL(k_op_smooth, dim3(1), dim3(64), 0, 0, h->d_prm, L, hs, r, o);
    int rc = S.done("pam_op_smooth"); if (rc) return rc;
    S.back(out, o, J3);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_velocity(PamHandle* h, int L, const double* hist, float* vel) {
    if (!h) return PAM_E_ARG;
    OpStage S(h);
    ARGCHK(h, L >= 2 && L <= 4096, "bad L");
    const double* hs = S.in(hist, (size_t)L * J3); float* o = S.out<float>(J3);
    OP_CHECK_STAGE("pam_op_velocity");
    hipLaunchKernelGGL(k_op_velocity, dim3(1), dim3(64), 0, 0, L, hs, o);
    int rc = S.done("pam_op_velocity"); if (rc) return rc;
    S.back(vel, o, J3);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

extern "C" int pam_op_hyp_cost(PamHandle* h, int n_members, const int32_t* cids, const double* poses, int o_cid,
                               const double* o_pose, double* cost, int32_t* veto) {
    OP_PROLOG(h);
    ARGCHK(h, n_members >= 1 && n_members <= PAM_MAX_VIEWS && o_cid >= 0 && o_cid < h->d.C, "bad members / camera");
    const int* c = S.in(cids, n_members); const double* ps = S.in(poses, (size_t)n_members * J3); const double* po = S.in(o_pose, J3);
    double* oc = S.out<double>(1); int* ov = S.out<int>(1);
    OP_CHECK_STAGE("pam_op_hyp_cost");
    hipLaunchKernelGGL(k_op_hyp_cost, dim3(1), dim3(64), 0, 0, camset(h), h->d_prm, n_members, c, ps, o_cid, po, oc, ov);
    int rc = S.done("pam_op_hyp_cost"); if (rc) return rc;
    S.back(cost, oc, 1); S.back(veto, ov, 1);
    return S.ok ? PAM_OK : PAM_E_HIP;
}

// ---- row e: the ONE exchange of the path, inside the C ABI --------------------------------------------------------------------
// pam_allgather_keypoints enqueues an RCCL all-gather of this rank's per-view keypoint records on the caller's (decode) stream, so a
// host that is not PyTorch can run the sharded path with nothing but this library.  RCCL is bound at run time (dlopen of
// librccl.so.1: the process may already carry PyTorch's copy, which then serves these calls too); single-GPU users never load it.
#include <dlfcn.h>
struct Id128 { char b[128]; };                          // ncclUniqueId (passed by value to ncclCommInitRank)
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace
static RcclApi g_rccl;
static std::string g_rccl_err;
static bool rccl_load() {
    if (g_rccl.lib) return true;
    void* l = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!l) { g_rccl_err = std::string("dlopen librccl: ") + dlerror(); return false; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(l, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(l, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(l, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(l, "ncclAllGather");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(l, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) { g_rccl_err = "librccl lacks the nccl* entry points"; dlclose(l); return false; }
    g_rccl.lib = l;
    return true;
}
#define RCCLCHK(call) do { int e_ = (call); if (e_ != 0) { g_rccl_err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "error"); \
    return PAM_E_HIP; } } while (0)

extern "C" const char* pam_comm_last_error(void) { return g_rccl_err.c_str(); }
extern "C" int pam_comm_unique_id(void* id128) {
    if (!id128) return PAM_E_ARG;
    if (!rccl_load()) return PAM_E_STATE;
    RCCLCHK(g_rccl.GetUniqueId(id128));
    return PAM_OK;
}
extern "C" int pam_comm_init(void** comm, int world, int rank, const void* id128, int device) {
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return PAM_E_ARG;
    if (!rccl_load()) return PAM_E_STATE;
    if (hipSetDevice(device) != hipSuccess) { g_rccl_err = "hipSetDevice failed"; return PAM_E_HIP; }
    Id128 id; memcpy(id.b, id128, 128);
    RCCLCHK(g_rccl.CommInitRank(comm, world, id, rank));
    return PAM_OK;
}
extern "C" int pam_comm_destroy(void* comm) {
    if (!comm) return PAM_OK;
    if (!rccl_load()) return PAM_E_STATE;
    RCCLCHK(g_rccl.CommDestroy(comm));
    return PAM_OK;
}
extern "C" int pam_allgather_keypoints(PamHandle* h, void* comm, void* stream, const double* dev_send, int rows_per_rank, double* dev_recv) {
    if (!h) return PAM_E_ARG;
    ARGCHK(h, comm && dev_send && dev_recv && rows_per_rank >= 1, "null communicator / buffer");
    if (!rccl_load()) { h->err = g_rccl_err; return PAM_E_STATE; }
    // one record per view: (max_dets + 1) rows of 51 float64 -- the detection rows, then the view's detection count in the first
    // double of the last row (ViewGather's layout, read in place by pam_frame_dev_views); ncclFloat64 = 8
    const size_t count = (size_t)rows_per_rank * ((size_t)h->d.MAXP + 1) * J3;
    const int e = g_rccl.AllGather(dev_send, dev_recv, count, 8 /* ncclFloat64 */, comm, (hipStream_t)stream);
    if (e != 0) { h->err = std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return PAM_E_HIP; }
    return PAM_OK;
}
