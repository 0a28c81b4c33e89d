"""CPU ORACLE -- test infrastructure, not product code.

NumPy restatement of the reference's per-frame matching / triangulation / tracking path
(B10532021/Part-Aware_Measurement_for_3D_Pose_Estimation_and_Tracking).  Only tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg may import this module, and only as the checker / the timed CPU baseline.
The product path (package ``pam``) never routes through it.

Pinning: every function below is checked in tests/test_oracle_golden.py against vectors captured by RUNNING the
reference in the build container (tools/make_goldens.py -> tests/golden/*.npz): per-function records and whole
tracker sequences.  Bar: assignment / view indices bit-equal, floats <= 1e-9.
Unpinned third-party boundaries (absent from /root/reference): OpenCV computeCorrespondEpilines
(opencv-python==4.2.0.32) -- restated from its published semantics; SciPy linear_sum_assignment -- restated below
and fuzzed against the installed SciPy; SciPy gaussian_filter1d -- restated and checked against the installed SciPy.

Layout conventions: keypoints inside the tracker are rows (y, x, score); P, K, RT, F,
RK_INV are float32, camera position float64; all association math is float64 except where the reference stores
float32 (init-path epipolar distances, velocity).
Citations are to /root/reference/src/.
"""
import math
import numpy as np

J = 17
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3


# ------------------------------------------------------------------------------------------------
# a18  cameras                                                              ivclabpose.py:35-46,162-181
# ------------------------------------------------------------------------------------------------
class Cam(object):
    __slots__ = ('cid', 'P', 'K', 'RT', 'F', 'RK_INV', 'position')


def fundamental_matrices(K, RT):
    """F[x, y] for every ordered camera pair in float32 torch CPU algebra, ivclabpose.py:166-177."""
    import torch
    C = len(K)
    Kt = [torch.tensor(K[i]) for i in range(C)]
    Rt = [torch.tensor(RT[i][:, :3]) for i in range(C)]
    Tt = [torch.tensor(RT[i][:, 3]) for i in range(C)]
    F = torch.zeros(C, C, 3, 3)
    for x in range(C):
        for y in range(C):
            v = Kt[y] @ Rt[y] @ Rt[x].t() @ (Tt[x] - Rt[x] @ Rt[y].t() @ Tt[y])
            skew = torch.tensor([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
            F[x, y] += torch.inverse(Kt[x]).t() @ (Rt[x] @ Rt[y].t()) @ Kt[y].t() @ skew
            if F[x, y].sum() == 0:
                F[x, y] += 1e-12
    return F.numpy()


def make_cameras(calib, F=None):
    P = np.asarray(calib['P']).astype(np.float32)
    K = np.asarray(calib['K']).astype(np.float32)
    RT = np.asarray(calib['RT']).astype(np.float32)
    if F is None:
        F = fundamental_matrices(K, RT)
    cams = []
    for j in range(len(P)):
        c = Cam()
        c.cid, c.P, c.K, c.RT, c.F = j, P[j], K[j], RT[j], F[j]
        c.RK_INV = np.linalg.inv(RT[j][:, :3]) @ np.linalg.inv(K[j])                   # :41 (float32)
        c.position = np.linalg.inv(np.vstack([RT[j], [0, 0, 0, 1]]))[:3, 3]           # :45-46 (float64)
        cams.append(c)
    return cams


def cameras_from_arrays(P32, K32, RT32, F, RK_INV, position):
    cams = []
    for j in range(len(P32)):
        c = Cam()
        c.cid, c.P, c.K, c.RT, c.F, c.RK_INV, c.position = j, P32[j], K32[j], RT32[j], F[j], RK_INV[j], position[j]
        cams.append(c)
    return cams


# ------------------------------------------------------------------------------------------------
# a3  re-projection                                                        ivclabpose.py:91-98
# ------------------------------------------------------------------------------------------------
def project_tracks(P, poses3d):
    """(n,17,3) world joints -> (n,17,2) image points in (y, x)."""
    n = poses3d.shape[0]
    hom = np.concatenate([poses3d, np.ones((n, poses3d.shape[1], 1))], axis=2).reshape(-1, 4)
    h = (P @ hom.T).T
    uv = h[:, :2] / h[:, 2].reshape(-1, 1)
    return uv[:, ::-1].reshape(n, -1, 2)


# ------------------------------------------------------------------------------------------------
# a4  track <-> detection affinity                                   IterativeTracker.py:137-149
# ------------------------------------------------------------------------------------------------
def association_affinity(reproj, dets, dt, alpha2d, lambda_a):
    """reproj (n,17,2) (y,x); dets (m,17,3) (y,x,s); dt (n,) frame gaps -> (n,m) affinity."""
    n, m = len(reproj), len(dets)
    dt = np.asarray(dt)
    diff = reproj[:, None, :, :] - dets[None, :, :, :2]
    dist = np.sqrt(diff[..., 0] ** 2 + diff[..., 1] ** 2)                              # (n,m,17)
    with np.errstate(divide='ignore', invalid='ignore'):
        c = 1 - dist / (alpha2d * dt)[:, None, None]
        pos = c > 0
        cnt = pos.sum(axis=2)
        aff = np.sum(c, where=pos, axis=2) / cnt
        aff[~(cnt > 10)] = 0
        aff = aff / np.exp(lambda_a * dt)[:, None]
    aff[np.isnan(aff)] = 0
    return aff


# ------------------------------------------------------------------------------------------------
# a5  rectangular LSAP (SciPy linear_sum_assignment; third-party, restated)   IterativeTracker.py:79,150
# ------------------------------------------------------------------------------------------------
def lsap(cost):
    """Shortest-augmenting-path assignment (Crouse 2016) with SciPy's conventions: tall matrices are transposed,
    the not-yet-scanned column list is filled in reverse so constant matrices give the identity, on equal
    path cost a still-unassigned column is preferred, results sorted by row.  Returns (rows, cols)."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    transposed = nc < nr
    if transposed:
        cost = cost.T.copy()
        nr, nc = nc, nr
    u = np.zeros(nr)
    v = np.zeros(nc)
    path = np.full(nc, -1, dtype=np.int64)
    col4row = np.full(nr, -1, dtype=np.int64)
    row4col = np.full(nc, -1, dtype=np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        SR = np.zeros(nr, dtype=bool)
        SC = np.zeros(nc, dtype=bool)
        remaining = [nc - 1 - k for k in range(nc)]
        num_rem = nc
        min_val = 0.0
        i = cur
        sink = -1
        while sink == -1:
            index = -1
            lowest = np.inf
            SR[i] = True
            for it in range(num_rem):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError('cost matrix is infeasible')
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_rem -= 1
            remaining[index] = remaining[num_rem]
        u[cur] += min_val
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for c_ in range(nc):
            if SC[c_]:
                v[c_] -= min_val - spc[c_]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transposed:
        order = np.argsort(col4row, kind='stable')
        return col4row[order].astype(np.int64), order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), col4row.astype(np.int64)


# ------------------------------------------------------------------------------------------------
# a7  epipolar distances, vectorised form (update path, float64)           matching.py:115-151
# ------------------------------------------------------------------------------------------------
def epi_dist_parallel(cams, pose_mat):
    """pose_mat (V,17,3) (y,x,s) of V views of one track -> symmetric distances (V,V,17)."""
    V = len(pose_mat)
    pose_mat = np.asarray(pose_mat, dtype=np.float64)
    nj = pose_mat.shape[1]
    x = np.concatenate([pose_mat[:, :, 1:2], pose_mat[:, :, 0:1], np.ones((V, nj, 1))], axis=2)   # (V,17,3) (x,y,1)
    Ft = np.zeros((V, V, 3, 3))
    for i in range(V):
        for j in range(V):
            if cams[i].cid != cams[j].cid:
                Ft[i, j] = cams[i].F[cams[j].cid].T
    # line in view j of point k of view i: l = F[ci][cj]^T (x_i)
    lines = np.transpose(Ft.reshape(V * V, 3, 3) @ np.transpose(np.repeat(x, V, 0), (0, 2, 1)), (0, 2, 1))
    nu = np.linalg.norm(lines[:, :, :2], axis=2).reshape(-1, nj, 1)
    nu[nu == 0] = 1
    lines = lines / nu
    nrm = np.sum(lines[:, :, :2] ** 2, axis=2)
    nrm[nrm == 0] = 1
    d = np.abs(np.sum(np.tile(x, (V, 1, 1)) * lines, axis=2)) / np.sqrt(nrm)
    d = d.reshape(V, V, -1)
    return (d + np.transpose(d, (1, 0, 2))) / 2


# ------------------------------------------------------------------------------------------------
# a15/a16  epipolar distances, loop form via OpenCV epilines (init path)    matching.py:50-113
# ------------------------------------------------------------------------------------------------
def _cv_epilines(xy, which, F):
    """cv::computeCorrespondEpilines for CV_64F points (OpenCV 4.2 fundam.cpp), restated."""
    f = np.asarray(F, dtype=np.float64)
    if which == 2:
        f = f.T
    f = f.reshape(-1)
    px, py = xy[:, 0], xy[:, 1]
    a = f[0] * px + f[1] * py + f[2]
    b = f[3] * px + f[4] * py + f[5]
    c = f[6] * px + f[7] * py + f[8]
    nu = a * a + b * b
    with np.errstate(divide='ignore'):
        nu = np.where(nu != 0, 1.0 / np.sqrt(nu), 1.0)
    return np.stack([a * nu, b * nu, c * nu], axis=1)


def epi_dist_pair(cam1, person1, cam2, person2):
    """(17,2): column 0 = distance of person1's points to the lines induced by person2, column 1 the converse."""
    F = cam1.F[cam2.cid]
    p1 = np.asarray(person1)[:, 1::-1]
    p2 = np.asarray(person2)[:, 1::-1]
    l_1to2 = _cv_epilines(p1, 2, F)
    l_2to1 = _cv_epilines(p2, 1, F)
    h1 = np.concatenate([p1, np.ones((len(p1), 1))], axis=1)
    h2 = np.concatenate([p2, np.ones((len(p2), 1))], axis=1)
    d1 = np.abs(np.sum(h1 * l_2to1, axis=1)) / np.sqrt(np.sum(l_2to1[:, :2] ** 2, axis=1))
    d2 = np.abs(np.sum(h2 * l_1to2, axis=1)) / np.sqrt(np.sum(l_1to2[:, :2] ** 2, axis=1))
    return np.stack([d1, d2], axis=1)


def epi_dist_loop(cams, poses):
    """Symmetric per-joint distances stored as float32 (matching.py:96,108-109)."""
    V = len(poses)
    out = np.zeros((V, V, J), dtype=np.float32)
    for i in range(V - 1):
        for j in range(i + 1, V):
            if cams[i].cid == cams[j].cid:
                continue
            d = epi_dist_pair(cams[i], poses[i], cams[j], poses[j])
            m = (d[:, 0] + d[:, 1]) / 2
            out[i, j] = m
            out[j, i] = m
    return out


# ------------------------------------------------------------------------------------------------
# a8  greedy per-joint view filter                          matching.py:10-17,243-295 ; calculate.py:26-32
# ------------------------------------------------------------------------------------------------
def ray_point_distance(cam, yx, X):
    """Distance from 3D point X to the back-projected ray of image point (y,x) of ``cam``."""
    h = np.array([yx[1], yx[0], 1.0])
    d = cam.RK_INV @ h
    d = d / np.linalg.norm(d)
    x1 = cam.position
    x2 = cam.position + d
    cr = np.cross(x2 - x1, x1 - np.asarray(X, dtype=np.float64))
    return np.linalg.norm(cr) / np.linalg.norm(x2 - x1)


def greedy_filter(cams, aff, mode, pose_j=None, next_pose_j=None):
    """aff (V,V): 1 - d/threshold for ONE joint.  Returns (kept view indices, 2V row mask).
    update mode: of each conflicting pair (aff<0, i<j, row-major) drop the view whose ray is farther from the
    predicted joint (ties drop j).  init mode: drop the view with the smaller affinity row-sum (ties drop i)."""
    V = aff.shape[0]
    alive = np.ones(V, dtype=bool)
    rows, cols = np.where(np.triu(aff) < 0)
    cache = np.zeros(V)
    for r, c in zip(rows, cols):
        if not (alive[r] and alive[c]):
            continue
        if mode == 'update':
            for k in (r, c):
                if cache[k] == 0:
                    cache[k] = ray_point_distance(cams[k], pose_j[k], next_pose_j)
            drop = r if cache[r] > cache[c] else c
        else:
            drop = c if np.sum(aff[r]) > np.sum(aff[c]) else r
        alive[drop] = False
    return np.nonzero(alive)[0], np.repeat(alive.astype(np.int64), 2)


# ------------------------------------------------------------------------------------------------
# a9  weighted DLT                                                         construction.py:89-114
# ------------------------------------------------------------------------------------------------
def dlt_rows(cams, pose_mat, Ts, lambda_t):
    """(17, 2V, 4): per view rows [x*P2-P0 ; y*P2-P1], each L2-normalised, times exp(-lambda_t*T)."""
    blocks = []
    for cam, poses, T in zip(cams, pose_mat, Ts):
        xy = np.asarray(poses)[:, 1::-1].reshape(-1, 1)                                # x0,y0,x1,y1,...
        R = xy * cam.P[2][None, :] - np.tile(cam.P[:2], (len(poses), 1))
        R = R / np.linalg.norm(R, axis=1).reshape(-1, 1)
        blocks.append((math.exp(-lambda_t * T) * R).reshape(-1, 2, 4))
    return np.concatenate(blocks, axis=1)


def dlt_solve(A, mask, nviews, next_pose=None):
    """A (17,2V,4), mask (17,2V) 0/1, nviews (17,) kept views per joint -> (17,3)."""
    out = np.zeros((J, 3))
    mask = np.asarray(mask) == 1
    for k in np.unique(nviews):
        js = np.nonzero(nviews == k)[0]
        if k <= 1:
            out[js] = next_pose[js]
            continue
        sub = A[js][mask[js]].reshape(len(js), -1, 4)
        _, _, VT = np.linalg.svd(sub)
        X = VT[:, -1, :]
        out[js] = X[:, :3] / X[:, 3].reshape(-1, 1)
    return out


# ------------------------------------------------------------------------------------------------
# a12  temporal Gaussian smoothing (SciPy gaussian_filter1d, mode reflect, truncate 4; restated)
#                                                                          IterativeTracker.py:371-383
# ------------------------------------------------------------------------------------------------
def gaussian_taps(sigma, truncate=4.0):
    """One-sided taps w[0..r] (w[0] centre), as scipy.ndimage._gaussian_kernel1d builds them."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return phi[r:].copy()


def _reflect(i, n):
    """scipy 'reflect' (half-sample symmetric) index extension."""
    if n == 1:
        return 0
    p = 2 * n
    i = i % p
    return i if i < n else p - 1 - i


def smooth_last(hist, raw, sigma, arm_sigma):
    """hist (L,17,3) stored (already smoothed) poses, raw (17,3) new pose -> smoothed new pose: the LAST sample of
    gaussian_filter1d over [hist..., raw] along time; wrists (9,10) use arm_sigma."""
    seq = np.concatenate([np.asarray(hist, dtype=np.float64), np.asarray(raw, dtype=np.float64)[None]], axis=0)
    n = len(seq)
    out = np.empty((J, 3))
    for joints, s in ((np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14, 15, 16]), sigma), (np.array([9, 10]), arm_sigma)):
        w = gaussian_taps(s)
        r = len(w) - 1
        c = n - 1
        acc = seq[c][joints] * w[0]
        for o in range(r, 0, -1):                                                      # farthest tap first
            acc = acc + (seq[_reflect(c - o, n)][joints] + seq[_reflect(c + o, n)][joints]) * w[o]
        out[joints] = acc
    return out


# ------------------------------------------------------------------------------------------------
# a13  velocity                                                            IterativeTracker.py:385-395
# ------------------------------------------------------------------------------------------------
def velocity_from_history(hist):
    """Mean of the <=5 newest consecutive differences, in float32.  hist (L,17,3), L >= 2."""
    diffs = []
    for k in range(len(hist) - 1, 0, -1):
        diffs.append(hist[k].astype(np.float32) - hist[k - 1].astype(np.float32))
        if len(diffs) > 4:
            break
    return np.mean(diffs, axis=0)


# ------------------------------------------------------------------------------------------------
# a14/a15  hypothesis cost                                   hypothesis.py:53-68 ; calculate.py:8-14
# ------------------------------------------------------------------------------------------------
def believe(pose):
    s = np.asarray(pose)[:, 2]
    return np.mean(s[s >= 0])


def hyp_cost(member_cams, member_poses, o_cam, o_pose, threshold):
    veto = False
    total = 0
    for cam, person in zip(member_cams, member_poses):
        d = epi_dist_pair(cam, person, o_cam, o_pose)
        p = np.mean([(d[k, 0] * person[k][2] + d[k, 1] * o_pose[k][2]) / 2 for k in range(J)]) / threshold
        total += p
        if p > 1 and believe(o_pose) > 0.5:
            veto = True
    return total / len(member_poses), veto


# ------------------------------------------------------------------------------------------------
# tracker state machine            IterativeTracker.py:52-180,182-395 ; hypothesis.py:9-44,70-77
# ------------------------------------------------------------------------------------------------
class Track(object):
    def __init__(self, track_id, time, cams, poses2d, pose3d, nviews, V):
        self.track_id = track_id
        self.hits = 1
        self.age = 1
        self.tsu = 0
        self.already = False
        self.state = TENTATIVE
        # ordered 2D store: view order = insertion order (Appendix A-5)
        self.order = []
        self.p2d = {}
        for cam, pose in zip(cams, poses2d):
            if cam.cid not in self.p2d:
                self.order.append(cam.cid)
            self.p2d[cam.cid] = (time, cam, pose)
        self.hist_t = [time]
        self.hist = [np.array(pose3d)]
        self.nviews = np.array(nviews)       # kept views per joint of the newest pose
        self.V = V                           # number of views offered for the newest pose
        self.velocity = np.zeros((J, 3))


class Params(object):
    def __init__(self, matcher, conf_threshold):
        g = (lambda k: matcher[k]) if isinstance(matcher, dict) else (lambda k: getattr(matcher, k))
        self.conf_threshold = conf_threshold
        self.epi_threshold = g('EPI_THRESHOLD')
        self.init_threshold = g('INIT_THRESHOLD')
        self.joint_threshold = g('JOINT_THRESHOLD')
        self.n_init = g('N_INIT')
        self.max_age = g('MAX_AGE')
        self.alpha2d = g('ALPHA2D')
        self.lambda_a = g('LAMBDA_A')
        self.lambda_t = g('LAMBDA_T')
        self.sigma = g('SIGMA')
        self.arm_sigma = g('ARM_SIGMA')


def joints_views_list(nviews, V):
    """The reference's public ``joints_views``: list of length V, entry k = joints triangulated from k+1 views."""
    out = [[] for _ in range(V)]
    for j in range(J):
        out[int(nviews[j]) - 1].append(j)
    return out


class Tracker(object):
    def __init__(self, params, cams):
        self.p = params
        self.cams = cams
        self.tracks = []
        self.next_id = 0
        self.unmatched = {}

    # -- per-track update ---------------------------------------------------  IterativeTracker.py:253-395
    def _triangulate_update(self, tr, time, cams, Ts, pose_mat):
        p = self.p
        last_t, last = tr.hist_t[-1], tr.hist[-1]
        pred = last + tr.velocity * (time - last_t)
        V = len(cams)
        aff = 1 - epi_dist_parallel(cams, pose_mat) / p.joint_threshold
        mask = np.ones((J, 2 * V), dtype=np.int64)
        nviews = np.zeros(J, dtype=np.int64)
        fails = 0
        for j in range(J):
            keep, mask[j] = greedy_filter(cams, aff[:, :, j], 'update', pose_mat[:, j, :], pred[j])
            nviews[j] = len(keep)
            if len(keep) < 2:
                fails += 1
        pose3d = dlt_solve(dlt_rows(cams, pose_mat, Ts, p.lambda_t), mask, nviews, pred)
        return pose3d, nviews, not (fails > J / 3)

    def _update_track(self, tr, time):
        p = self.p
        ok = False
        if tr.already:
            sel = [cid for cid in tr.order if time - tr.p2d[cid][0] <= 3]
            if len(sel) >= 2:
                cams = [tr.p2d[c][1] for c in sel]
                Ts = [time - tr.p2d[c][0] for c in sel]
                pose_mat = np.array([tr.p2d[c][2] for c in sel])
                pose3d, nviews, ok = self._triangulate_update(tr, time, cams, Ts, pose_mat)
                if ok:
                    sm = smooth_last(np.array(tr.hist), pose3d, p.sigma, p.arm_sigma)
                    tr.hist.append(sm)
                    tr.hist_t.append(time)
                    tr.nviews, tr.V = nviews, len(sel)
                    if time - tr.hist_t[0] > p.max_age:
                        del tr.hist[0]
                        del tr.hist_t[0]
        if ok:
            if len(tr.hist) >= 2:
                tr.velocity = velocity_from_history(tr.hist)
            tr.hits += 1
            tr.tsu = 0
            if tr.state == TENTATIVE and tr.hits >= p.n_init:
                tr.state = CONFIRMED
        else:
            if tr.state == TENTATIVE and not tr.already:
                tr.state = DELETED
            elif tr.tsu >= p.max_age:
                tr.state = DELETED

    # -- new-track initialisation -------------------------------------------  IterativeTracker.py:52-113
    def _init_tracks(self, time):
        p = self.p
        if len(self.unmatched) < 2:
            return
        hyps = []                                  # each: ([cams], [poses])
        for idx, cid in enumerate(self.unmatched):
            cam, dets = self.unmatched[cid]
            dets = [d for d in dets if believe(d) > p.conf_threshold]
            self.unmatched[cid] = (cam, dets)
            if idx == 0:
                hyps = [([cam], [d]) for d in dets]
                continue
            nh, nd = len(hyps), len(dets)
            Cm = np.zeros((nh, nd))
            veto = np.zeros((nh, nd), dtype=bool)
            for h, (hc, hp) in enumerate(hyps):
                for d, det in enumerate(dets):
                    Cm[h, d], veto[h, d] = hyp_cost(hc, hp, cam, det, p.epi_threshold)
            rows, cols = lsap(Cm)
            handled = set()
            for h, d in zip(rows, cols):
                handled.add(int(d))
                if veto[h, d]:
                    hyps.append(([cam], [dets[d]]))
                else:
                    hyps[h][0].append(cam)
                    hyps[h][1].append(dets[d])
            for d in range(nd):
                if d not in handled:
                    hyps.append(([cam], [dets[d]]))
        for hc, hp in hyps:
            if len(hp) < 2:
                continue
            V = len(hc)
            aff = 1 - epi_dist_loop(hc, hp) / p.init_threshold                        # float32
            mask = np.ones((J, 2 * V), dtype=np.int64)
            nviews = np.zeros(J, dtype=np.int64)
            good = True
            for j in range(J):
                keep, mask[j] = greedy_filter(hc, aff[:, :, j], 'init')
                nviews[j] = len(keep)
                if len(keep) < 2:
                    good = False
                    break
            if not good:
                continue
            pose3d = dlt_solve(dlt_rows(hc, hp, [0] * V, p.lambda_t), mask, nviews)
            self.tracks.append(Track(self.next_id, time, hc, hp, pose3d, nviews, V))
            self.next_id += 1

    # -- one frame ----------------------------------------------------------  IterativeTracker.py:115-180
    def step(self, frame_id, detections):
        """detections: list over views of (m,17,3) float64 arrays in (y,x,score) (m may be 0)."""
        p = self.p
        poses, dts = [], []
        for tr in self.tracks:
            tr.already = False
            tr.age += 1
            tr.tsu += 1
            poses.append(tr.hist[-1])
            dts.append(frame_id - tr.hist_t[-1])
        self.assign = []                             # per view: det index -> matched track position or -1
        for cam, dets in zip(self.cams, detections):
            n, m = len(self.tracks), len(dets)
            asg = np.full(m, -1, dtype=np.int64)
            if n > 0 and m > 0:
                dets = np.asarray(dets)
                aff = association_affinity(project_tracks(cam.P, np.array(poses)), dets, dts, p.alpha2d, p.lambda_a)
                rows, cols = lsap(-aff)
                for t, d in zip(rows, cols):
                    if aff[t, d] > 0:
                        tr = self.tracks[t]
                        tr.already = True
                        if cam.cid not in tr.p2d:
                            tr.order.append(cam.cid)
                        tr.p2d[cam.cid] = (frame_id, cam, dets[d])
                        asg[d] = t
                rest = [dets[d] for d in range(m) if asg[d] < 0]
            else:
                rest = [d for d in dets]
            self.unmatched[cam.cid] = (cam, rest)
            self.assign.append(asg)
        for tr in self.tracks:
            self._update_track(tr, frame_id)
        self._init_tracks(frame_id)
        self.tracks = [t for t in self.tracks if t.state != DELETED]

    # -- output collection --------------------------------------------------  ivclabpose.py:259-287
    def collect(self, frame_id):
        camera_ids, pts, person_ids, pts3d, jviews, ids = [], [], [], [], [], []
        for tr in self.tracks:
            if tr.tsu > 0 or tr.state != CONFIRMED:
                continue
            pts3d.append(tr.hist[-1].T)
            jviews.append(joints_views_list(tr.nviews, tr.V))
            ids.append(tr.track_id)
            person_ids.append([tr.track_id] * len(tr.order))
            cs = [c for c in tr.order if tr.p2d[c][0] == frame_id]
            camera_ids.append(cs)
            pts.append([tr.p2d[c][2] for c in cs])
        return camera_ids, pts, person_ids, np.array(pts3d), jviews, np.array(ids)


def unpack_dump(dump_results):
    """a2: dump dicts -> per-view (n,17,3) float64 (y, x, score).                 ivclabpose.py:221-254"""
    out = []
    for items in dump_results:
        rows = []
        for it in items:
            k = np.array(it['keypoints'], dtype=np.float64).reshape(J, 3)
            rows.append(np.stack([k[:, 1], k[:, 0], np.array(it['keypoints_score'], dtype=np.float64)], axis=1))
        out.append(np.array(rows).reshape(-1, J, 3))
    return out


class OracleIvclabpose(object):
    """The reference façade's matching surface on top of the oracle tracker (CPU)."""
    def __init__(self, matcher, conf_threshold):
        self.params = Params(matcher, conf_threshold)
        self.tracker = None

    def GetCameraParameters(self, calib, im_width=0, im_height=0, F=None):
        self.cameras = make_cameras(calib, F)
        self.tracker = Tracker(self.params, self.cameras)
        return self.cameras

    def PersonTrack_Project3DPose(self, frame_id, person_bbox_list=None, dump_results=None, build3D='SVD'):
        assert build3D == 'SVD'
        dets = unpack_dump(dump_results)
        self.tracker.step(frame_id, dets)
        c, p, pid, p3, jv, ids = self.tracker.collect(frame_id)
        return (np.array(c, dtype='object'), np.array(p, dtype='object'), pid, p3, jv, ids, 0.0, 0.0, 0.0)
