"""Seeded synthetic multi-view scenes.

Produces camera rigs in the reference's calibration layout ({'P','K','RT'}, the dict
``testmodel.py`` unpickles, /root/reference/src/testmodel.py:29-30) and per-frame 2D
detections in the ``dump_results`` layout that ``HRNetPose.predict`` hands to
``ivclabpose.PersonTrack_Project3DPose`` (/root/reference/src/ivclabpose.py:233-246):
per view a list of dicts with ``bbox`` (x, y, w, h), ``keypoints`` (flat 51 = 17 x [x, y, .])
and ``keypoints_score`` (17).

Used by bench.py, the tests and tools/make_goldens.py.  NumPy only.
"""
import numpy as np

NUM_JOINTS = 17

# name: (C views, P persons, image w, image h, focal px)
SIZES = {
    'S1': dict(C=3, P=3, w=360, h=288, f=400.0),     # Campus-like
    'S2': dict(C=5, P=4, w=1032, h=776, f=1000.0),   # Shelf-like
    'S3': dict(C=5, P=7, w=1920, h=1080, f=1000.0),  # Panoptic 5 HD
    'S4': dict(C=31, P=7, w=1920, h=1080, f=1000.0),  # Panoptic 31 HD
}

# Matcher thresholds of the reference YAMLs (src/configs/*/model_configs.yaml).
MATCHER_CFG = {
    'CampusSeq1': dict(CONF_THRESHOLD=0.4, EPI_THRESHOLD=25, INIT_THRESHOLD=15, JOINT_THRESHOLD=15,
                       NUM_JOINTS=17, INIT_METHOD='GD', N_INIT=3, MAX_AGE=10, W2D=0.4, ALPHA2D=30,
                       W3D=0.6, ALPHA3D=0.1, LAMBDA_A=3, LAMBDA_T=5, SIGMA=0.6, ARM_SIGMA=0.8),
    'Shelf': dict(CONF_THRESHOLD=0.5, EPI_THRESHOLD=60, INIT_THRESHOLD=30, JOINT_THRESHOLD=60,
                  NUM_JOINTS=17, INIT_METHOD='GD', N_INIT=3, MAX_AGE=10, W2D=0.4, ALPHA2D=70,
                  W3D=0.6, ALPHA3D=0.15, LAMBDA_A=3, LAMBDA_T=5, SIGMA=0.3, ARM_SIGMA=0.8),
    'Panoptic': dict(CONF_THRESHOLD=0.4, EPI_THRESHOLD=60, INIT_THRESHOLD=50, JOINT_THRESHOLD=30,
                     NUM_JOINTS=17, INIT_METHOD='GD', N_INIT=3, MAX_AGE=10, W2D=0.4, ALPHA2D=60,
                     W3D=0.6, ALPHA3D=0.15, LAMBDA_A=3, LAMBDA_T=5, SIGMA=0.3, ARM_SIGMA=0.8),
}
SIZE_TO_DATASET = {'S1': 'CampusSeq1', 'S2': 'Shelf', 'S3': 'Panoptic', 'S4': 'Panoptic'}

# COCO-17 template, metres, z up, facing +x, ~1.7 m tall.
TEMPLATE = np.array([
    [0.10, 0.00, 1.58], [0.08, 0.03, 1.62], [0.08, -0.03, 1.62], [0.00, 0.08, 1.60], [0.00, -0.08, 1.60],
    [0.00, 0.20, 1.42], [0.00, -0.20, 1.42], [0.02, 0.26, 1.12], [0.02, -0.26, 1.12],
    [0.10, 0.28, 0.86], [0.10, -0.28, 0.86], [0.00, 0.11, 0.92], [0.00, -0.11, 0.92],
    [0.03, 0.12, 0.50], [0.03, -0.12, 0.50], [0.00, 0.12, 0.08], [0.00, -0.12, 0.08]], dtype=np.float64)

OUTLIER_JOINTS = (9, 10, 15, 16)  # wrists / ankles


def make_rig(C, w, h, f, radius=5.0, height=2.5, target=(0.0, 0.0, 1.0)):
    """Ring of C pin-hole cameras looking at ``target``.  Returns {'P','K','RT'} float64."""
    P = np.zeros((C, 3, 4)); K = np.zeros((C, 3, 3)); RT = np.zeros((C, 3, 4))
    tgt = np.asarray(target, dtype=np.float64)
    for c in range(C):
        th = 2.0 * np.pi * c / C + 0.1
        pos = np.array([radius * np.cos(th), radius * np.sin(th), height])
        fwd = tgt - pos; fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        K[c] = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
        RT[c, :, :3] = R
        RT[c, :, 3] = -R @ pos
        P[c] = K[c] @ RT[c]
    return {'P': P, 'K': K, 'RT': RT}


def _pose_world(center, heading, jitter):
    c, s = np.cos(heading), np.sin(heading)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    return TEMPLATE @ Rz.T + np.array([center[0], center[1], 0.0]) + jitter


def make_sequence(size='S2', n_frames=303, seed=0, noise_px=1.5, outlier_p=0.05,
                  occlusion_every=50, empty_view_every=37, birth_death_frame=100, shuffle=True,
                  blank_frames=(), churn_every=0, churn_len=0):
    """Synthetic sequence.  Returns dict with
       'calib'   : {'P','K','RT'}
       'frames'  : list over frames of list over views of (n,17,3) float64 arrays in the HRNet dump
                   layout (x, y, score) (n may be 0)
       'gt3d'    : list over frames of dict person_id -> (17,3) world joints
       'meta'    : size parameters
    churn_every / churn_len (round 5, the longer S3 / S4 golden traces): person (t // churn_every) % P is invisible in EVERY view during the
    first churn_len frames of each period of churn_every frames -- its track dies (max_age) and a new one is born when it comes back, so
    a trace sees a death and a birth every period instead of once per sequence.  0 = off (the random stream is then unchanged).
    """
    cfg = SIZES[size]
    C, Pn, w, h, f = cfg['C'], cfg['P'], cfg['w'], cfg['h'], cfg['f']
    rng = np.random.default_rng(seed)
    calib = make_rig(C, w, h, f)
    Pm = calib['P']
    # persons: Pn regular ones + one that appears at birth_death_frame; one regular leaves there
    n_slots = Pn + 1
    ang = 2 * np.pi * np.arange(n_slots) / n_slots + 0.3
    centers = np.stack([1.2 * np.cos(ang), 1.2 * np.sin(ang)], axis=1)
    headings = rng.uniform(0, 2 * np.pi, n_slots)
    frames, gt3d = [], []
    for t in range(n_frames):
        centers = centers + rng.normal(0, 0.01, centers.shape)
        headings = headings + rng.normal(0, 0.01, n_slots)
        if t < birth_death_frame:
            alive = list(range(Pn))
        else:
            alive = [p for p in range(n_slots) if p != 0] if Pn > 1 else list(range(Pn))
        world = {}
        for p in alive:
            world[p] = _pose_world(centers[p], headings[p], rng.normal(0, 0.01, (NUM_JOINTS, 3)))
        gt3d.append(world)
        empty_view = int(rng.integers(C)) if (empty_view_every and t > 0 and t % empty_view_every == 0) else -1
        occl = (int(rng.integers(C)), int(rng.integers(len(alive)))) if (occlusion_every and t > 0 and t % occlusion_every == 0) else (-1, -1)
        views = []
        for c in range(C):
            dets = []
            for k, p in enumerate(alive):
                X = np.concatenate([world[p], np.ones((NUM_JOINTS, 1))], axis=1)
                hm = X @ Pm[c].T
                xy = hm[:, :2] / hm[:, 2:3] + rng.normal(0, noise_px, (NUM_JOINTS, 2))
                sc = rng.uniform(0.7, 0.95, NUM_JOINTS)
                if rng.uniform() < outlier_p:
                    j = OUTLIER_JOINTS[int(rng.integers(len(OUTLIER_JOINTS)))]
                    a = rng.uniform(0, 2 * np.pi); r = rng.uniform(80, 200)
                    xy[j] += r * np.array([np.cos(a), np.sin(a)])
                if c == empty_view or (c == occl[0] and k == occl[1]) or t in blank_frames:
                    continue
                if churn_every and t >= churn_every and (t % churn_every) < churn_len and p == (t // churn_every) % Pn:
                    continue
                dets.append(np.concatenate([xy, sc[:, None]], axis=1))
            if shuffle and len(dets) > 1:
                order = rng.permutation(len(dets))
                dets = [dets[i] for i in order]
            views.append(np.array(dets, dtype=np.float64).reshape(-1, NUM_JOINTS, 3))
        frames.append(views)
    return {'calib': calib, 'frames': frames, 'gt3d': gt3d,
            'meta': dict(size=size, C=C, P=Pn, w=w, h=h, f=f, seed=seed, n_frames=n_frames)}


def to_dump_results(view_dets):
    """(n,17,3) (x,y,score) arrays per view -> (person_bbox_list, dump_results) as the reference's
    PersonPoseDetect returns them (ivclabpose.py:195-203, 233-246)."""
    person_bbox_list, dump_results = [], []
    for dets in view_dets:
        pb, dr = [], []
        for kp in dets:
            x0, y0 = kp[:, 0].min(), kp[:, 1].min()
            x1, y1 = kp[:, 0].max(), kp[:, 1].max()
            bw, bh = (x1 - x0) * 1.25, (y1 - y0) * 1.25
            bbox = [float(x0 - 0.125 * (x1 - x0)), float(y0 - 0.125 * (y1 - y0)), float(bw), float(bh)]
            pb.append(dict(image_id=0, category_id=1, score=1.0, bbox=bbox, data=None, feature=[]))
            kk = kp.copy()
            dr.append(dict(bbox=bbox, keypoints=kk.reshape(-1).tolist(),
                           keypoints_score=kp[:, 2].tolist(), feature=[]))
        person_bbox_list.append(pb)
        dump_results.append(dr)
    return person_bbox_list, dump_results


def pack_frames(frames, Pmax):
    """list[frame][view] (n,17,3) -> n_det (F,C) int32 and det (F,C,Pmax,17,3) float64 in the tracker's
    internal (y, x, score) layout (the swap of ivclabpose.py:236-244)."""
    F, C = len(frames), len(frames[0])
    n_det = np.zeros((F, C), dtype=np.int32)
    det = np.zeros((F, C, Pmax, NUM_JOINTS, 3), dtype=np.float64)
    for t, views in enumerate(frames):
        for c, d in enumerate(views):
            n = len(d)
            if n > Pmax:
                raise ValueError('Pmax too small')
            n_det[t, c] = n
            if n:
                det[t, c, :n, :, 0] = d[:, :, 1]
                det[t, c, :n, :, 1] = d[:, :, 0]
                det[t, c, :n, :, 2] = d[:, :, 2]
    return n_det, det
