// Device functions of the matching / triangulation path (gfx950, wave64).  Shared by the fused per-frame kernel and
// by the per-operator test kernels, so what the parity tests exercise is what the frame step runs.
// All association math is float64 (the reference's NumPy dtype); float32 only where the reference stores float32.
// Citations: /root/reference/src/...
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pam.h"

#define J PAM_J
#define J3 (PAM_J * 3)

namespace pam {

struct CamSet {              // device-resident camera constants of a rig (a18)
    const float* P;          // C*12   float32 (ivclabpose.py:163)
    const float* F;          // C*C*9  float32, F[a][b]: x_a^T F x_b = 0 (ivclabpose.py:166-177)
    const float* RKINV;      // C*9    float32 (ivclabpose.py:41)
    const double* pos;       // C*3    float64 (ivclabpose.py:45-46)
    int C;
};

// ---- a3: pin-hole projection, ivclabpose.py:91-98 --------------------------------------------------------------
__device__ __forceinline__ void project_point(const float* __restrict__ P, double X, double Y, double Z,
                                              double& u, double& v) {
    double h0 = (double)P[0] * X + (double)P[1] * Y + (double)P[2] * Z + (double)P[3];
    double h1 = (double)P[4] * X + (double)P[5] * Y + (double)P[6] * Z + (double)P[7];
    double h2 = (double)P[8] * X + (double)P[9] * Y + (double)P[10] * Z + (double)P[11];
    const double ih = 1.0 / h2;            // one fp64 division instead of two (<= 1 ulp from numpy's h0/h2, h1/h2)
    u = h0 * ih;
    v = h1 * ih;
}

// ---- a3+a4: track <-> detection affinity for one (track, detection) pair, IterativeTracker.py:137-149 -----------
// pose3d 17x3 world; det 17x3 rows (y, x, score); alpha_dt = alpha2d*dt; exp_ldt = exp(lambda_a*dt)
__device__ inline double track_det_affinity(const float* __restrict__ P, const double* __restrict__ pose3d,
                                            const double* __restrict__ det, double alpha_dt, double exp_ldt, int gate) {
    double sum = 0.0;
    int cnt = 0;
    const double inv_alpha_dt = 1.0 / alpha_dt;
    for (int j = 0; j < J; ++j) {
        double u, v;
        project_point(P, pose3d[j * 3 + 0], pose3d[j * 3 + 1], pose3d[j * 3 + 2], u, v);
        double dy = v - det[j * 3 + 0];
        double dx = u - det[j * 3 + 1];
        double c = 1.0 - sqrt(dy * dy + dx * dx) * inv_alpha_dt;
        if (c > 0.0) { sum += c; ++cnt; }
    }
    double aff = (cnt > gate) ? sum / (double)cnt : 0.0;
    aff = aff / exp_ldt;
    if (aff != aff) aff = 0.0;
    return aff;
}

// The same value for the MP detection slots of ONE (view, track) pair held by MP neighbouring lanes (MP = 8, 16 or 32; round 6): the
// track's 17 re-projections -- each with its fp64 division -- are computed ONCE per group, joint j by the group's lane j % MP, and handed
// round through the wave; every lane then runs track_det_affinity's own loop (same expressions, same joint order) on the received
// (u, v).  All lanes of the wave must call it; `live` = this lane has a detection to score.
template <int MP>
__device__ inline double track_det_affinity_group(const float* __restrict__ P, const double* __restrict__ pose3d,
                                                  const double* __restrict__ det, double alpha_dt, double exp_ldt, int gate, bool live) {
    constexpr int Q = (J + MP - 1) / MP;
    const int lane = threadIdx.x & 63, k = lane & (MP - 1), base = lane - k;
    double pu[Q], pv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = k + q * MP;
        pu[q] = pv[q] = 0.0;
        if (j < J) project_point(P, pose3d[j * 3 + 0], pose3d[j * 3 + 1], pose3d[j * 3 + 2], pu[q], pv[q]);
    }
    double sum = 0.0;
    int cnt = 0;
    const double inv_alpha_dt = 1.0 / alpha_dt;
#pragma unroll
    for (int q = 0; q < Q; ++q) {                        // (static register index per q; the walk over a row's lanes stays a loop)
#pragma unroll 1
        for (int m = 0; m < MP && q * MP + m < J; ++m) {
            const int j = q * MP + m;
            const double u = __shfl(pu[q], base + m);
            const double v = __shfl(pv[q], base + m);
            if (live) {
                double dy = v - det[j * 3 + 0];
                double dx = u - det[j * 3 + 1];
                double c = 1.0 - sqrt(dy * dy + dx * dx) * inv_alpha_dt;
                if (c > 0.0) { sum += c; ++cnt; }
            }
        }
    }
    double aff = (cnt > gate) ? sum / (double)cnt : 0.0;
    aff = aff / exp_ldt;
    if (aff != aff) aff = 0.0;
    return aff;
}

// ---- a5: rectangular LSAP (SciPy linear_sum_assignment restated), call sites IterativeTracker.py:79,150 ---------
struct LsapScratch {         // N = max(nr, nc) entries each
    double* u; double* v; double* spc;
    int* path; int* col4row; int* row4col; int* remaining;
    unsigned char* SR; unsigned char* SC;
};
__host__ __device__ inline size_t lsap_scratch_bytes(int N) {
    return (size_t)N * (3 * sizeof(double) + 4 * sizeof(int)) + 2 * ((N + 7) & ~7);
}
__device__ inline LsapScratch lsap_carve(void* base, int N) {
    LsapScratch s;
    char* p = (char*)base;
    s.u = (double*)p; p += sizeof(double) * N;
    s.v = (double*)p; p += sizeof(double) * N;
    s.spc = (double*)p; p += sizeof(double) * N;
    s.path = (int*)p; p += sizeof(int) * N;
    s.col4row = (int*)p; p += sizeof(int) * N;
    s.row4col = (int*)p; p += sizeof(int) * N;
    s.remaining = (int*)p; p += sizeof(int) * N;
    s.SR = (unsigned char*)p; p += (N + 7) & ~7;
    s.SC = (unsigned char*)p;
    return s;
}
// minimise sum sign*cost[r*ld + c]; writes pairs sorted by row; returns number of pairs (min(nr,nc)) or -1 (infeasible)
__device__ inline int lsap_solve(int nr0, int nc0, const double* cost, int ld, double sign, LsapScratch s,
                                 int* rows_out, int* cols_out) {
    if (nr0 == 0 || nc0 == 0) return 0;
    const bool tr = nc0 < nr0;                       // tall matrices are solved transposed
    const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
#define PAM_COST(i, j) (sign * (tr ? cost[(j) * ld + (i)] : cost[(i) * ld + (j)]))
    for (int i = 0; i < nr; ++i) { s.u[i] = 0.0; s.col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { s.v[j] = 0.0; s.row4col[j] = -1; s.path[j] = -1; }
    const double INF = __builtin_huge_val();
    for (int cur = 0; cur < nr; ++cur) {
        for (int j = 0; j < nc; ++j) { s.spc[j] = INF; s.SC[j] = 0; s.remaining[j] = nc - 1 - j; }
        for (int i = 0; i < nr; ++i) s.SR[i] = 0;
        int num_rem = nc, i = cur, sink = -1;
        double min_val = 0.0;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            s.SR[i] = 1;
            const double ui = s.u[i];
            for (int it = 0; it < num_rem; ++it) {
                const int j = s.remaining[it];
                const double r = min_val + PAM_COST(i, j) - ui - s.v[j];
                double sp = s.spc[j];
                if (r < sp) { s.path[j] = i; s.spc[j] = r; sp = r; }
                if (sp < lowest || (sp == lowest && s.row4col[j] == -1)) { lowest = sp; index = it; }
            }
            min_val = lowest;
            if (min_val == INF) return -1;
            const int j = s.remaining[index];
            if (s.row4col[j] == -1) sink = j; else i = s.row4col[j];
            s.SC[j] = 1;
            s.remaining[index] = s.remaining[--num_rem];
        }
        s.u[cur] += min_val;
        for (int r = 0; r < nr; ++r)
            if (s.SR[r] && r != cur) s.u[r] += min_val - s.spc[s.col4row[r]];
        for (int c = 0; c < nc; ++c)
            if (s.SC[c]) s.v[c] -= min_val - s.spc[c];
        int j = sink;
        for (;;) {
            const int ii = s.path[j];
            s.row4col[j] = ii;
            const int t = s.col4row[ii]; s.col4row[ii] = j; j = t;
            if (ii == cur) break;
        }
    }
#undef PAM_COST
    int n = 0;
    if (tr) {   // solved rows are original columns; emit sorted by original row
        for (int r = 0; r < nc; ++r)
            if (s.row4col[r] != -1) { rows_out[n] = r; cols_out[n] = s.row4col[r]; ++n; }
    } else {
        for (int r = 0; r < nr; ++r) { rows_out[n] = r; cols_out[n] = s.col4row[r]; ++n; }
    }
    return n;
}

// The same algorithm run by ONE WAVE (round 6): lane j keeps column j's state (v, shortest path cost, predecessor, assigned row, position
// in the `remaining` list), lane i row i's (u, assigned column, visited flag); every step of the shortest-augmenting-path search is one
// parallel relaxation of all remaining columns + wave reductions instead of a lane walking the list through LDS (11.6 us per frame for
// the five 4 x 4 problems of a Shelf frame, 33 us for the 31-camera rig's, all in dependent LDS round trips).  Results are the serial
// solver's bit for bit: the relaxations are the same expressions, and its tie rule -- walking `remaining` in list order, a column
// replaces the running minimum when strictly lower, or equal AND unassigned -- means: the LAST unassigned minimal column in list order
// if there is one, else the FIRST minimal column; list order is kept per lane (`pos`; removal moves the last entry into the hole).
// All 64 lanes must call it (max(nr0, nc0) <= 64); lane L < returned n holds pair (row, col) L of the row-sorted result.
// cross-lane helpers of the wave solver: a value of a wave-UNIFORM lane is a v_readlane (no LDS-pipe round trip like ds_bpermute), and a
// reduction over the 16 lanes of a DPP row is four row rotations
__device__ __forceinline__ int lane_bcast_i(int x, int src) { return __builtin_amdgcn_readlane(x, src); }
__device__ __forceinline__ double lane_bcast_d(double x, int src) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#define PAM_DPP_ROR(x, n) __builtin_amdgcn_update_dpp(0, (x), 0x120 + (n), 0xF, 0xF, false)
__device__ __forceinline__ double row16_min_d(double v) {
#define PAM_STEP(n) { const unsigned long long b = __builtin_bit_cast(unsigned long long, v); \
        const unsigned lo = (unsigned)PAM_DPP_ROR((int)(unsigned)b, n), hi = (unsigned)PAM_DPP_ROR((int)(unsigned)(b >> 32), n); \
        const double o = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); v = o < v ? o : v; }
    PAM_STEP(8) PAM_STEP(4) PAM_STEP(2) PAM_STEP(1)
#undef PAM_STEP
    return v;
}
__device__ __forceinline__ int row16_max_i(int v) {
    int o;
    o = PAM_DPP_ROR(v, 8); v = o > v ? o : v; o = PAM_DPP_ROR(v, 4); v = o > v ? o : v;
    o = PAM_DPP_ROR(v, 2); v = o > v ? o : v; o = PAM_DPP_ROR(v, 1); v = o > v ? o : v;
    return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
    int o;
    o = PAM_DPP_ROR(v, 8); v = o < v ? o : v; o = PAM_DPP_ROR(v, 4); v = o < v ? o : v;
    o = PAM_DPP_ROR(v, 2); v = o < v ? o : v; o = PAM_DPP_ROR(v, 1); v = o < v ? o : v;
    return v;
}

__device__ inline int lsap_solve_wave(int nr0, int nc0, const double* cost, int ld, double sign, int* rows_out, int* cols_out,
                                      int& my_row, int& my_col) {
    const int lane = threadIdx.x & 63;
    my_row = my_col = -1;
    if (nr0 == 0 || nc0 == 0) return 0;
    const bool tr = nc0 < nr0;
    const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
    int width = 1;
    while (width < nc) width <<= 1;                      // the reductions run over the lanes that can hold a column
    const bool row16 = width <= 16;                      // ... one DPP row: rotations instead of ds_bpermute exchanges
    const double INF = __builtin_huge_val();
    double u = 0.0, v = 0.0, spc = INF;
    int col4row = -1, row4col = -1, path = -1, pos = -1;
    for (int cur = 0; cur < nr; ++cur) {
        spc = INF;
        pos = lane < nc ? nc - 1 - lane : -1;            // remaining[it] = nc - 1 - it
        bool SR = false, SC = false;
        int num_rem = nc, i = cur, sink = -1;
        double min_val = 0.0;
        while (sink == -1) {
            if (lane == i) SR = true;
            const double ui = lane_bcast_d(u, i);
            const bool active = pos >= 0;
            double sp = spc;
            if (active) {
                const double c = sign * (tr ? cost[(size_t)lane * ld + i] : cost[(size_t)i * ld + lane]);
                const double r = min_val + c - ui - v;
                if (r < sp) { path = i; spc = r; sp = r; }
            }
            double lowest = active ? sp : INF;
            if (row16) lowest = row16_min_d(lowest);
            else for (int m = 1; m < width; m <<= 1) { const double o = __shfl_xor(lowest, m); lowest = o < lowest ? o : lowest; }
            lowest = lane_bcast_d(lowest, 0);
            if (lowest == INF) return -1;
            const bool ismin = active && sp == lowest;
            const bool unas = ismin && row4col == -1;
            const bool anyun = __ballot(unas) != 0ull;
            // the chosen list position: max over the unassigned minima, else min over the minima
            int key = anyun ? (unas ? pos : -1) : (ismin ? pos : 0x7fffffff);
            if (row16) key = anyun ? row16_max_i(key) : row16_min_i(key);
            else for (int m = 1; m < width; m <<= 1) { const int o = __shfl_xor(key, m); key = anyun ? (o > key ? o : key) : (o < key ? o : key); }
            key = lane_bcast_i(key, 0);
            const int jsel = __ffsll((long long)__ballot(active && pos == key)) - 1;
            min_val = lowest;
            const int r4c = lane_bcast_i(row4col, jsel);
            if (r4c == -1) sink = jsel; else i = r4c;
            --num_rem;                                   // remaining[index] = remaining[--num_rem]
            if (lane == jsel) { SC = true; pos = -1; }
            else if (pos == num_rem) pos = key;
        }
        // dual update (the row / column roles of a lane are separate variables)
        const double spc_of_mine = __shfl(spc, col4row >= 0 ? col4row : 0);
        if (lane == cur) u += min_val;
        else if (lane < nr && SR) u += min_val - spc_of_mine;
        if (SC) v -= min_val - spc;
        // augment along the predecessor chain
        int j = sink;
        for (;;) {
            const int ii = lane_bcast_i(path, j);
            if (lane == j) row4col = ii;
            const int t = lane_bcast_i(col4row, ii);
            if (lane == ii) col4row = j;
            j = t;
            if (ii == cur) break;
        }
    }
    int n;
    if (tr) {                                            // solved rows are original columns; emit sorted by original row
        const bool has = lane < nc && row4col != -1;
        const unsigned long long b = __ballot(has);
        n = __popcll(b);
        const int at = __popcll(b & ((1ull << lane) - 1ull));
        if (has) { rows_out[at] = lane; cols_out[at] = row4col; }
        // lane L < n holds pair L: fetch it from the lane that owns it
        unsigned long long bb = b;
        int owner = -1;
        for (int q = 0; q <= lane && bb; ++q) { owner = __ffsll((long long)bb) - 1; bb &= bb - 1; if (q == lane) break; owner = -1; }
        const int src = owner >= 0 ? owner : 0;
        const int rc = __shfl(row4col, src);
        if (lane < n) { my_row = src; my_col = rc; }
    } else {
        n = nr;
        if (lane < nr) { rows_out[lane] = lane; cols_out[lane] = col4row; my_row = lane; my_col = col4row; }
    }
    return n;
}

// ---- a7: epipolar point-to-line distance, vectorised ("parallel") form, matching.py:115-151 ---------------------
// distance of point j to the line induced in its view by point i:  l = F[ci][cj]^T (xi, yi, 1)
__device__ __forceinline__ double epi_directed(const float* __restrict__ Fij, double xi, double yi, double xj, double yj) {
    double l0 = (double)Fij[0] * xi + (double)Fij[3] * yi + (double)Fij[6];
    double l1 = (double)Fij[1] * xi + (double)Fij[4] * yi + (double)Fij[7];
    double l2 = (double)Fij[2] * xi + (double)Fij[5] * yi + (double)Fij[8];
    // matching.py:141-146 normalises the line (l /= nu, nu = 0 -> 1) and then divides the dot product by the norm of the
    // normalised line again (= 1 to within an ulp).  One division by nu gives the same value to ~1e-16 relative at a third of
    // the fp64 sqrt/div cost; the parity tests bound the difference at 1e-9.
    const double s2 = l0 * l0 + l1 * l1;
    const double inv_nu = (s2 == 0.0) ? 1.0 : rsqrt(s2);
    return fabs(xj * l0 + yj * l1 + l2) * inv_nu;
}
// symmetric distance between view a (camera ca, point pa=(y,x,.)) and view b; 0 for the same camera (matching.py:133-134)
__device__ __forceinline__ double epi_sym(const CamSet& cs, int ca, const double* pa, int cb, const double* pb) {
    if (ca == cb) return 0.0;
    const float* Fab = cs.F + ((size_t)ca * cs.C + cb) * 9;
    const float* Fba = cs.F + ((size_t)cb * cs.C + ca) * 9;
    double dab = epi_directed(Fab, pa[1], pa[0], pb[1], pb[0]);
    double dba = epi_directed(Fba, pb[1], pb[0], pa[1], pa[0]);
    return (dab + dba) / 2.0;
}

// ---- a15: OpenCV computeCorrespondEpilines form, matching.py:50-91 ----------------------------------------------
// F = cams[c1].F[c2]; d1 = distance of p1 to the line F p2, d2 = distance of p2 to the line F^T p1
__device__ __forceinline__ void epi_pair_cv(const float* __restrict__ F, double x1, double y1, double x2, double y2,
                                            double& d1, double& d2) {
    // line in image 2 from p1 (whichImage = 2 -> F transposed)
    double a = (double)F[0] * x1 + (double)F[3] * y1 + (double)F[6];
    double b = (double)F[1] * x1 + (double)F[4] * y1 + (double)F[7];
    double c = (double)F[2] * x1 + (double)F[5] * y1 + (double)F[8];
    double nu = a * a + b * b;
    nu = (nu != 0.0) ? 1.0 / sqrt(nu) : 1.0;
    a *= nu; b *= nu; c *= nu;
    d2 = fabs(x2 * a + y2 * b + c) / sqrt(a * a + b * b);
    // line in image 1 from p2 (whichImage = 1)
    double e = (double)F[0] * x2 + (double)F[1] * y2 + (double)F[2];
    double f = (double)F[3] * x2 + (double)F[4] * y2 + (double)F[5];
    double g = (double)F[6] * x2 + (double)F[7] * y2 + (double)F[8];
    double mu = e * e + f * f;
    mu = (mu != 0.0) ? 1.0 / sqrt(mu) : 1.0;
    e *= mu; f *= mu; g *= mu;
    d1 = fabs(x1 * e + y1 * f + g) / sqrt(e * e + f * f);
}
// float32-stored symmetric init-path distance for views a<b in list order, matching.py:99-109
__device__ __forceinline__ float epi_sym_init(const CamSet& cs, int ca, const double* pa, int cb, const double* pb) {
    if (ca == cb) return 0.0f;
    double d1, d2;
    epi_pair_cv(cs.F + ((size_t)ca * cs.C + cb) * 9, pa[1], pa[0], pb[1], pb[0], d1, d2);
    return (float)((d1 + d2) / 2.0);
}

// ---- a8: back-projected ray to predicted joint, matching.py:10-17 + calculate.py:26-32 --------------------------
__device__ inline double ray_point_dist(const float* __restrict__ RK, const double* __restrict__ pos, double x, double y,
                                        const double* __restrict__ Xp) {
    double dx = (double)RK[0] * x + (double)RK[1] * y + (double)RK[2];
    double dy = (double)RK[3] * x + (double)RK[4] * y + (double)RK[5];
    double dz = (double)RK[6] * x + (double)RK[7] * y + (double)RK[8];
    const double in_ = rsqrt(dx * dx + dy * dy + dz * dz);
    dx *= in_; dy *= in_; dz *= in_;
    double ex = (pos[0] + dx) - pos[0], ey = (pos[1] + dy) - pos[1], ez = (pos[2] + dz) - pos[2];
    double wx = pos[0] - Xp[0], wy = pos[1] - Xp[1], wz = pos[2] - Xp[2];
    double cx = ey * wz - ez * wy, cy = ez * wx - ex * wz, cz = ex * wy - ey * wx;
    return sqrt(cx * cx + cy * cy + cz * cz) * rsqrt(ex * ex + ey * ey + ez * ez);
}

// ---- a8/a16: greedy per-joint view filter, matching.py:243-295 --------------------------------------------------
// conf[r] = bit c set iff c > r and affinity(r,c) < 0.  Pairs are visited row-major; a pair is skipped once either
// member is gone.  update: drop r iff ray[r] > ray[c] else c.  init: drop c iff sum[r] > sum[c] else r.
__device__ inline uint32_t greedy_keep_update(int V, const uint32_t* conf, const double* ray) {
    uint32_t alive = (V >= 32) ? 0xffffffffu : ((1u << V) - 1u);
    for (int r = 0; r < V; ++r) {
        uint32_t m = conf[r];
        while (m && ((alive >> r) & 1u)) {
            int c = __ffs((int)m) - 1;
            m &= m - 1;
            if (!((alive >> c) & 1u)) continue;
            if (ray[r] > ray[c]) alive &= ~(1u << r); else alive &= ~(1u << c);
        }
    }
    return alive;
}
__device__ inline uint32_t greedy_keep_init(int V, const uint32_t* conf, const float* rowsum) {
    uint32_t alive = (V >= 32) ? 0xffffffffu : ((1u << V) - 1u);
    for (int r = 0; r < V; ++r) {
        uint32_t m = conf[r];
        while (m && ((alive >> r) & 1u)) {
            int c = __ffs((int)m) - 1;
            m &= m - 1;
            if (!((alive >> c) & 1u)) continue;
            if (rowsum[r] > rowsum[c]) alive &= ~(1u << c); else alive &= ~(1u << r);
        }
    }
    return alive;
}
// NumPy's float32 add.reduce order over a strided 1-D row (pairwise_sum: <8 sequential, else 8 accumulators)
template <typename Fn>
__device__ inline float np_sum_f32(int n, Fn at) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) r += at(i);
        return r;
    }
    float r0 = at(0), r1 = at(1), r2 = at(2), r3 = at(3), r4 = at(4), r5 = at(5), r6 = at(6), r7 = at(7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += at(i); r1 += at(i + 1); r2 += at(i + 2); r3 += at(i + 3);
        r4 += at(i + 4); r5 += at(i + 5); r6 += at(i + 6); r7 += at(i + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += at(i);
    return res;
}

// ---- a9: weighted DLT, construction.py:89-114 -------------------------------------------------------------------
// Null vector of the stacked (2k x 4) system WITHOUT forming the normal matrix: rows are folded one at a time into a
// 4x4 upper-triangular R by Givens rotations (row-wise QR), then the right singular vector of R's smallest singular
// value comes from one-sided (Hestenes) Jacobi on R's columns.  Backward stable like the LAPACK SVD the reference calls
// (rows weighted by exp(-lambda_t*T) ~ 3e-7 would lose half the digits in A^T A).  Fully unrolled: registers only.
__device__ __forceinline__ void givens_fold_row(double R[4][4], double r[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double b = r[k];
        if (b != 0.0) {
            const double a = R[k][k];
            const double h2 = a * a + b * b;
            const double ih = rsqrt(h2);
            const double c = a * ih, s = b * ih;
            R[k][k] = h2 * ih;
#pragma unroll
            for (int m = k + 1; m < 4; ++m) {
                const double t = R[k][m];
                R[k][m] = c * t + s * r[m];
                r[m] = c * r[m] - s * t;
            }
        }
    }
}
__device__ inline void min_right_singular_vector(double W[4][4], double out[4]) {
    double E[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    // columns are orthogonal to working precision once |<wp,wq>| <= 1e-15 |wp||wq|; quadratic convergence gets there in 5-7
    // sweeps (a tighter bound sits below the rounding noise of the dot products and never triggers)
    for (int sweep = 0; sweep < 12; ++sweep) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { al += W[i][p] * W[i][p]; be += W[i][q] * W[i][q]; ga += W[i][p] * W[i][q]; }
                if (ga != 0.0 && ga * ga > 1e-30 * (al * be)) {
                    // tan of the rotation angle: t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (be - al) / (2 ga), written
                    // with one sqrt, one division and one rsqrt
                    const double df = be - al, g2 = 2.0 * ga;
                    const double t = copysign(1.0, df * g2) * fabs(g2) / (fabs(df) + sqrt(df * df + g2 * g2));
                    const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double wp = W[i][p], wq = W[i][q];
                        W[i][p] = c * wp - s * wq; W[i][q] = s * wp + c * wq;
                        const double ep = E[i][p], eq = E[i][q];
                        E[i][p] = c * ep - s * eq; E[i][q] = s * ep + c * eq;
                    }
                    rotated = 1;
                }
            }
        }
        if (!rotated) break;
    }
    double n[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) n[c] = W[0][c] * W[0][c] + W[1][c] * W[1][c] + W[2][c] * W[2][c] + W[3][c] * W[3][c];
    int k = 0;
    double best = n[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) if (n[i] < best) { best = n[i]; k = i; }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (k == 0) ? E[r][0] : (k == 1) ? E[r][1] : (k == 2) ? E[r][2] : E[r][3];
}

// The same vector for an UPPER-TRIANGULAR W (what the Givens folds leave) by inverse iteration on W^T W with two triangular solves per
// step -- never forming the product: x <- W^-1 (W^-T x).  A triangulation's system has ONE small singular value (the noise) below three
// large ones, so each step shrinks the error by (s4 / s3)^2: two or three steps against the 5-7 sweeps x 6 rotations (each with a sqrt,
// a division and an rsqrt in fp64) of the Jacobi above, which stays as the fall-back for a zero pivot or a slow case (nearly parallel
// rays: s3 ~ s4).  -> true when it converged.
__device__ inline bool min_right_singular_vector_tri(const double W[4][4], double out[4]) {
    const double d0 = W[0][0], d1 = W[1][1], d2 = W[2][2], d3 = W[3][3];
    if (d0 == 0.0 || d1 == 0.0 || d2 == 0.0 || d3 == 0.0) return false;
    const double i0 = 1.0 / d0, i1 = 1.0 / d1, i2 = 1.0 / d2, i3 = 1.0 / d3;
    double x0 = 0.5, x1 = 0.5, x2 = 0.5, x3 = 0.5;
    for (int it = 0; it < 8; ++it) {
        // W^T y = x (forward), then W z = y (backward)
        const double y0 = x0 * i0;
        const double y1 = (x1 - W[0][1] * y0) * i1;
        const double y2 = (x2 - W[0][2] * y0 - W[1][2] * y1) * i2;
        const double y3 = (x3 - W[0][3] * y0 - W[1][3] * y1 - W[2][3] * y2) * i3;
        const double z3 = y3 * i3;
        const double z2 = (y2 - W[2][3] * z3) * i2;
        const double z1 = (y1 - W[1][2] * z2 - W[1][3] * z3) * i1;
        const double z0 = (y0 - W[0][1] * z1 - W[0][2] * z2 - W[0][3] * z3) * i0;
        const double nn = z0 * z0 + z1 * z1 + z2 * z2 + z3 * z3;
        if (!(nn > 0.0) || nn == __builtin_huge_val()) return false;
        const double in_ = rsqrt(nn);
        double n0 = z0 * in_, n1 = z1 * in_, n2 = z2 * in_, n3 = z3 * in_;
        // (z = (W^T W)^-1 x keeps x's orientation: <z, x> > 0, so successive iterates are compared as they are)
        const double e = fmax(fmax(fabs(n0 - x0), fabs(n1 - x1)), fmax(fabs(n2 - x2), fabs(n3 - x3)));
        x0 = n0; x1 = n1; x2 = n2; x3 = n3;
        if (it > 0 && e <= 1e-15) { out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3; return true; }
    }
    return false;
}

// one joint: views listed in sel_cid[0..V) with ages T -> weight w_t[T]; keep = bitmask over the V list positions;
// pose(v) returns the (y,x,score) row of list position v for this joint.  dlt_fold folds the kept views v = v0, v0+vstep, ...
// into the triangular factor R (so several lanes can share one joint); dlt_merge folds another partial factor in.
template <typename PoseFn>
__device__ inline void dlt_fold(const CamSet& cs, int V, int v0, int vstep, const int* sel_cid, const int* sel_T, const double* w_t,
                                double lambda_t, uint32_t keep, PoseFn pose, double R[4][4]) {
    for (int v = v0; v < V; v += vstep) {
        if (!((keep >> v) & 1u)) continue;
        const float* P = cs.P + (size_t)sel_cid[v] * 12;
        const double* p = pose(v);
        const int T = sel_T[v];
        const double w = (T >= 0 && T < 4) ? w_t[T] : exp(-lambda_t * (double)T);
        const double xy[2] = {p[1], p[0]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double r0 = xy[h] * (double)P[8] - (double)P[4 * h + 0];
            double r1 = xy[h] * (double)P[9] - (double)P[4 * h + 1];
            double r2 = xy[h] * (double)P[10] - (double)P[4 * h + 2];
            double r3 = xy[h] * (double)P[11] - (double)P[4 * h + 3];
            const double wn = w * rsqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
            double rr[4] = {wn * r0, wn * r1, wn * r2, wn * r3};
            givens_fold_row(R, rr);
        }
    }
}
__device__ inline void dlt_zero(double R[4][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[r][c] = 0.0;
}
// packed upper triangle <-> R
__device__ inline void dlt_pack(const double R[4][4], double* o) {
    o[0] = R[0][0]; o[1] = R[0][1]; o[2] = R[0][2]; o[3] = R[0][3]; o[4] = R[1][1]; o[5] = R[1][2]; o[6] = R[1][3];
    o[7] = R[2][2]; o[8] = R[2][3]; o[9] = R[3][3];
}
__device__ inline void dlt_merge(double R[4][4], const double* o) {
    double r0[4] = {o[0], o[1], o[2], o[3]}; givens_fold_row(R, r0);
    double r1[4] = {0.0, o[4], o[5], o[6]}; givens_fold_row(R, r1);
    double r2[4] = {0.0, 0.0, o[7], o[8]}; givens_fold_row(R, r2);
    double r3[4] = {0.0, 0.0, 0.0, o[9]}; givens_fold_row(R, r3);
}
__device__ inline void dlt_solve(double R[4][4], double out[3]) {
    double X[4];
#ifdef PAM_DLT_JACOBI_ONLY                               // check build: the fall-back alone (tools/ab_build.sh PAM_DLT_JACOBI_ONLY '<parity tests>')
    min_right_singular_vector(R, X);
#else
    if (!min_right_singular_vector_tri(R, X)) min_right_singular_vector(R, X);
#endif
    out[0] = X[0] / X[3]; out[1] = X[1] / X[3]; out[2] = X[2] / X[3];
}
template <typename PoseFn>
__device__ inline void dlt_joint(const CamSet& cs, int V, const int* sel_cid, const int* sel_T, const double* w_t,
                                 double lambda_t, uint32_t keep, PoseFn pose, double out[3]) {
    double R[4][4];
    dlt_zero(R);
    dlt_fold(cs, V, 0, 1, sel_cid, sel_T, w_t, lambda_t, keep, pose, R);
    dlt_solve(R, out);
}

// ---- a12: last sample of scipy gaussian_filter1d(mode='reflect'), IterativeTracker.py:371-383 -------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p; if (i < 0) i += p;
    return (i < n) ? i : p - 1 - i;
}
// seq(i): i in [0,n) sample i of one coordinate (n-1 = the new raw value); w[0..ntaps) one-sided taps
template <typename SeqFn>
__device__ inline double smooth_last(int n, int ntaps, const double* w, SeqFn seq) {
    const int c = n - 1;
    double acc = seq(c) * w[0];
    for (int o = ntaps - 1; o >= 1; --o)
        acc += (seq(reflect_idx(c - o, n)) + seq(reflect_idx(c + o, n))) * w[o];
    return acc;
}

// ---- a13: float32 velocity, IterativeTracker.py:385-395 ----------------------------------------------------------
// hist(k): coordinate value of history entry k (0 oldest .. L-1 newest), L >= 2
template <typename HistFn>
__device__ inline float velocity_f32(int L, HistFn hist) {
    float acc = 0.0f;
    int cnt = 0;
    for (int k = L - 1; k >= 1 && cnt < 5; --k, ++cnt) {
        const float d = (float)hist(k) - (float)hist(k - 1);
        acc = (cnt == 0) ? d : acc + d;
    }
    return acc / (float)cnt;
}

// ---- a14: mean keypoint confidence, calculate.py:8-14 ------------------------------------------------------------
__device__ inline double believe(const double* det) {
    double s = 0.0; int n = 0;
    for (int j = 0; j < J; ++j) { const double w = det[j * 3 + 2]; if (w >= 0.0) { s += w; ++n; } }
    return s / (double)n;    // n == 0 -> NaN, which fails every '>' test like np.mean([]) does
}

// ---- a15: one member's contribution to Hypothesis.calculate_cost, hypothesis.py:61-66 ---------------------------
__device__ inline double hyp_member_cost(const CamSet& cs, int cm, const double* pm, int co, const double* po, double thr) {
    const float* F = cs.F + ((size_t)cm * cs.C + co) * 9;
    double s = 0.0;
    for (int j = 0; j < J; ++j) {
        double d1, d2;
        epi_pair_cv(F, pm[j * 3 + 1], pm[j * 3 + 0], po[j * 3 + 1], po[j * 3 + 0], d1, d2);
        s += (d1 * pm[j * 3 + 2] + d2 * po[j * 3 + 2]) / 2.0;
    }
    return (s / (double)J) / thr;
}

}  // namespace pam
