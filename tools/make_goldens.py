#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference through
tools/ref_shims.py) on seeded synthetic scenes.  Development-time only: the reference is read-only and does
not exist on the GPU box; the fixtures this script writes are committed.

Usage (from the repo root, in its own process):  python tools/make_goldens.py [--out tests/golden]

Fixtures:
  cameras_<S>.npz   calibration dict in, Camera attributes (F, RK_INV, position) out        (a18)
  ops_<S>.npz       per-function input/output records captured INSIDE a real tracker run      (a3-a16)
  trace_<S>.npz     whole sequences: per-frame detections in, the 9-tuple + tracker state out (a2-a17)
  pcp_S2.npz        Evaluate3DPose_PCP on a synthetic actorsGT.mat                              (gate)
"""
import argparse
import os
import sys
import tempfile
import pickle
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shims  # noqa: E402
ref = ref_shims.load_reference()
ref_tracker = sys.modules['tracking.IterativeTracker']
import pam  # noqa: E402  (repo package alias)
from pam import synth  # noqa: E402

REC = defaultdict(list)
CTX = dict(phase=None, view=None, tracks_pose=None, dt=None, frame_id=None, dets=None)
KEEP_FIRST = 10
KEEP_INTERESTING = 14
COUNT = defaultdict(int)


def _keep(op, interesting=False):
    key = op + ('*' if interesting else '')
    COUNT[key] += 1
    return COUNT[key] <= (KEEP_INTERESTING if interesting else KEEP_FIRST)


def cam_ids(cams):
    return np.array([c.cid for c in cams], dtype=np.int32)


def install_recorders():
    tm = ref_tracker
    hm = ref.hypothesis
    mm = ref.matching

    # ---- a3 projection -------------------------------------------------------------------------
    Camera = ref.ivclabpose.Camera
    o_proj = Camera.projectPoints_parallel

    def proj(self, points3d):
        out = o_proj(self, points3d)
        CTX['view'] = self.cid
        if _keep('project'):
            REC['project'].append(dict(cid=np.int32(self.cid), pts=np.array(points3d), out=out.copy()))
        return out
    Camera.projectPoints_parallel = proj

    # ---- a5 LSAP (association + init) ----------------------------------------------------------
    o_lsap = tm.linear_sum_assignment

    def lsap(cost):
        rows, cols = o_lsap(cost)
        if CTX['phase'] == 'assoc':
            v = CTX['view']
            n, m = cost.shape
            inter = (n != m) or bool(np.any(cost[rows, cols] >= 0))
            if _keep('assoc', inter):
                REC['assoc'].append(dict(cid=np.int32(v), tracks_pose=np.array(CTX['tracks_pose']),
                                         dt=np.array(CTX['dt'], dtype=np.int64),
                                         dets=np.array(CTX['dets'][v]),
                                         affinity=-np.array(cost), rows=rows.astype(np.int32),
                                         cols=cols.astype(np.int32)))
        else:
            if _keep('lsap_init', cost.shape[0] != cost.shape[1]):
                REC['lsap_init'].append(dict(cost=np.array(cost), rows=rows.astype(np.int32),
                                             cols=cols.astype(np.int32)))
        return rows, cols
    tm.linear_sum_assignment = lsap

    o_tracking = tm.IterativeTracker.tracking

    def tracking(self, frame_id, camera_list, frame_list, boxes_list, detections_list, build3D='TopDown'):
        CTX['phase'] = 'assoc'
        CTX['frame_id'] = frame_id
        CTX['tracks_pose'] = [t.poses3d[-1]['pose3d'].copy() for t in self.tracks]
        CTX['dt'] = [frame_id - t.poses3d[-1]['time'] for t in self.tracks]
        CTX['dets'] = detections_list
        return o_tracking(self, frame_id, camera_list, frame_list, boxes_list, detections_list, build3D)
    tm.IterativeTracker.tracking = tracking

    o_init = tm.IterativeTracker.init_target_GD

    def init_gd(self, time):
        CTX['phase'] = 'init'
        return o_init(self, time)
    tm.IterativeTracker.init_target_GD = init_gd

    # ---- a7 epipolar (parallel form) -----------------------------------------------------------
    o_epar = tm.epipolar_affinity_parallel

    def epar(cameras, sub, pose_mat, num_joints):
        aff, dist = o_epar(cameras, sub, pose_mat, num_joints)
        if _keep('epi_par', len(cameras) != 5):
            REC['epi_par'].append(dict(cids=cam_ids(cameras), pose_mat=np.array(pose_mat), dist=dist.copy(),
                                       aff=aff.copy()))
        return aff, dist
    tm.epipolar_affinity_parallel = epar

    # ---- a8 greedy filter (both modes) ---------------------------------------------------------
    def make_greedy(orig, tag):
        def greedy(cameras, pose_mat=None, affinity_mat=None, costs=None, next_pose=None, mode='update'):
            ml, bl, am = orig(cameras, pose_mat=pose_mat, affinity_mat=affinity_mat, costs=costs,
                              next_pose=next_pose, mode=mode)
            inter = len(ml) < affinity_mat.shape[0]
            op = 'greedy_' + mode
            if _keep(op, inter):
                r = dict(cids=cam_ids(cameras), aff=np.array(affinity_mat), matched=np.array(ml, dtype=np.int32),
                         binary=np.array(bl, dtype=np.int32))
                if mode == 'update':
                    r['pose'] = np.array(pose_mat)
                    r['next_pose'] = np.array(next_pose)
                REC[op].append(r)
            return ml, bl, am
        return greedy
    tm.Greedy_matching = make_greedy(tm.Greedy_matching, 't')
    hm.Greedy_matching = make_greedy(hm.Greedy_matching, 'h')

    # ---- a9 DLT --------------------------------------------------------------------------------
    def make_dlt(orig, tag):
        def dlt(cameras, Ts, pose_mat, lambda_t, remains, joints_views, next_pose=None):
            rem_in = np.array(remains).copy()
            out = orig(cameras, Ts, pose_mat, lambda_t, remains, joints_views, next_pose)
            inter = bool(np.any(rem_in == 0))
            if _keep('dlt_' + tag, inter):
                nv = np.zeros(17, dtype=np.int32)
                for k, js in enumerate(joints_views):
                    for j in js:
                        nv[j] = k + 1
                REC['dlt_' + tag].append(dict(cids=cam_ids(cameras), Ts=np.array(Ts, dtype=np.int64),
                                              pose_mat=np.array(pose_mat), lambda_t=np.float64(lambda_t),
                                              remains=rem_in.astype(np.int32), nviews=nv,
                                              next_pose=(np.zeros((17, 3)) if next_pose is None else np.array(next_pose)),
                                              out=np.array(out)))
            return out
        return dlt
    tm.SVD_pose_kernel_jf = make_dlt(tm.SVD_pose_kernel_jf, 'update')
    hm.SVD_pose_kernel_jf = make_dlt(hm.SVD_pose_kernel_jf, 'init')

    # ---- a15/a16 loop-form epipolar + hypothesis cost ------------------------------------------
    o_edist = mm.epipolar_distance

    def edist(cam1, person1, cam2, person2):
        out = o_edist(cam1, person1, cam2, person2)
        if _keep('epi_distance'):
            REC['epi_distance'].append(dict(c1=np.int32(cam1.cid), p1=np.array(person1), c2=np.int32(cam2.cid),
                                            p2=np.array(person2), out=np.array(out)))
        return out
    mm.epipolar_distance = edist
    hm.epipolar_distance = edist

    o_eloop = hm.epipolar_affinity

    def eloop(cameras, sub, pose_mat, num_joints):
        aff, dist = o_eloop(cameras, sub, pose_mat, num_joints)
        if _keep('epi_loop', len(cameras) not in (3, 5)):
            REC['epi_loop'].append(dict(cids=cam_ids(cameras), pose_mat=np.array(pose_mat), dist=dist.copy()))
        return aff, dist
    hm.epipolar_affinity = eloop

    o_cost = hm.Hypothesis.calculate_cost

    def hcost(self, o_cam, o_pose):
        c, veto = o_cost(self, o_cam, o_pose)
        if _keep('hyp_cost', bool(veto)):
            REC['hyp_cost'].append(dict(cids=cam_ids(self.cams), poses=np.array(self.poses), o_cid=np.int32(o_cam.cid),
                                        o_pose=np.array(o_pose), thr=np.float64(self.threshold),
                                        cost=np.float64(c), veto=np.int32(bool(veto))))
        return c, veto
    hm.Hypothesis.calculate_cost = hcost

    # ---- a12 smoothing, a13 motion -------------------------------------------------------------
    IT = tm.IterTrack
    o_smooth = IT.smooth_3dpose

    def smooth(self, time, pose3d=None, sigma=0.3, arm_sigma=0.8):
        hist = np.array([p['pose3d'] for p in self.poses3d])
        raw = np.array(pose3d).copy()
        out = o_smooth(self, time, pose3d, sigma, arm_sigma)
        if _keep('smooth', len(hist) in (1, 2, 3)):
            REC['smooth'].append(dict(hist=hist, raw=raw, sigma=np.float64(sigma), arm_sigma=np.float64(arm_sigma),
                                      out=np.array(out)))
        return out
    IT.smooth_3dpose = smooth

    o_motion = IT.update_motion

    def motion(self, time):
        r = o_motion(self, time)
        if _keep('motion', len(self.poses3d) in (2, 3, 4, 5)):
            REC['motion'].append(dict(hist=np.array([p['pose3d'] for p in self.poses3d]),
                                      vel=np.array(self.velocity_3d), vel_dtype=str(np.asarray(self.velocity_3d).dtype)))
        return r
    IT.update_motion = motion


def matcher_cfg(dataset):
    d = dict(synth.MATCHER_CFG[dataset])
    conf = d.pop('CONF_THRESHOLD')
    m = ref_shims.AttrDict(NAME='Iterative', **d)
    return m, conf


def dump_state(model, C):
    tr = model.tracker.tracks
    n = len(tr)
    st = dict(
        ids=np.array([t.track_id for t in tr], dtype=np.int32),
        state=np.array([t.state for t in tr], dtype=np.int32),
        hits=np.array([t.hits for t in tr], dtype=np.int32),
        age=np.array([t.age for t in tr], dtype=np.int32),
        tsu=np.array([t.time_since_update for t in tr], dtype=np.int32),
        nhist=np.array([len(t.poses3d) for t in tr], dtype=np.int32),
        last_time=np.array([t.poses3d[-1]['time'] for t in tr], dtype=np.int32),
        last_pose=np.array([t.poses3d[-1]['pose3d'] for t in tr], dtype=np.float64).reshape(n, 17, 3),
        velocity=np.array([np.asarray(t.velocity_3d, dtype=np.float64) for t in tr]).reshape(n, 17, 3),
        p2d_order=np.full((n, C), -1, dtype=np.int32),
        p2d_time=np.full((n, C), -1, dtype=np.int32),
    )
    for i, t in enumerate(tr):
        for k, (cid, v) in enumerate(t.poses2d.items()):
            st['p2d_order'][i, k] = cid
            st['p2d_time'][i, cid] = v['time']
    return st


def run_trace(size, n_frames, seed, **seq_kw):
    dataset = synth.SIZE_TO_DATASET[size]
    seq = synth.make_sequence(size, n_frames=n_frames, seed=seed, **seq_kw)
    C = seq['meta']['C']
    matcher, conf = matcher_cfg(dataset)
    model = ref.ivclabpose.ivclabpose(person_detector=ref_shims.AttrDict(NAME=''), pose_detector=None,
                                      person_matcher=matcher, conf_threshold=conf)
    cams = model.GetCameraParameters(seq['calib'], seq['meta']['h'], seq['meta']['w'])
    cam_rec = dict(P=seq['calib']['P'], K=seq['calib']['K'], RT=seq['calib']['RT'],
                   P32=np.array([c.P for c in cams]), K32=np.array([c.K for c in cams]),
                   RT32=np.array([c.RT for c in cams]), F=np.array([c.F for c in cams]),
                   RK_INV=np.array([c.RK_INV for c in cams]), position=np.array([c.position for c in cams]))
    out = dict()
    # inputs, ragged: all detections concatenated + counts
    n_det = np.array([[len(v) for v in views] for views in seq['frames']], dtype=np.int32)
    flat = [v for views in seq['frames'] for v in views if len(v)]
    out['in.n_det'] = n_det
    out['in.dets'] = np.concatenate(flat, axis=0) if flat else np.zeros((0, 17, 3))
    out['meta.size'] = np.array(size)
    out['meta.dataset'] = np.array(dataset)
    out['meta.seed'] = np.int64(seed)
    skipped = []
    for t, views in enumerate(seq['frames']):
        pbl, dr = synth.to_dump_results(views)
        if np.array(dr, dtype='object').size > 0:
            (camera_ids, pts, person_ids, pts3d, jviews, p3d_ids, _, _, _) = \
                model.PersonTrack_Project3DPose(frame_id=t, person_bbox_list=pbl, dump_results=dr, build3D='SVD')
            n = len(p3d_ids)
            out['f%d.ids' % t] = np.array(p3d_ids, dtype=np.int32)
            out['f%d.pts3d' % t] = np.array(pts3d, dtype=np.float64).reshape(n, 3, 17)
            nv = np.zeros((n, 17), dtype=np.int32)
            nV = np.zeros(n, dtype=np.int32)
            for i, jv in enumerate(jviews):
                nV[i] = len(jv)
                for k, js in enumerate(jv):
                    for j in js:
                        nv[i, j] = k + 1
            out['f%d.nviews' % t] = nv
            out['f%d.V' % t] = nV
            cam_pad = np.full((n, C), -1, dtype=np.int32)
            for i, cids in enumerate(camera_ids):
                cam_pad[i, :len(cids)] = cids
            out['f%d.camera_ids' % t] = cam_pad
            out['f%d.n_person_ids' % t] = np.array([len(p) for p in person_ids], dtype=np.int32)
            # matched 2D poses (tracker (y,x,score) layout), concatenated in camera_ids order
            det_pad = np.full((n, C), -1, dtype=np.int32)
            for i, (cids, ps) in enumerate(zip(camera_ids, pts)):
                for k, (cid, p) in enumerate(zip(cids, ps)):
                    cand = views[cid]  # (x, y, score) dump layout; tracker layout is (y, x, score)
                    hit = [q for q in range(len(cand)) if np.array_equal(cand[q][:, [1, 0, 2]], np.asarray(p))]
                    assert len(hit) == 1
                    det_pad[i, k] = hit[0]
            out['f%d.pts_det' % t] = det_pad
        else:
            skipped.append(t)
        st = dump_state(model, C)
        for k, v in st.items():
            out['f%d.st.%s' % (t, k)] = v
    out['meta.skipped'] = np.array(skipped, dtype=np.int32)
    return seq, cam_rec, out, model


def save_ops(path):
    flat = {}
    for op, recs in REC.items():
        flat['%s.count' % op] = np.int32(len(recs))
        for i, r in enumerate(recs):
            for k, v in r.items():
                flat['%s.%d.%s' % (op, i, k)] = v
    np.savez_compressed(path, **flat)


def make_pcp(seq, trace, out_path):
    """Synthetic actorsGT.mat in the layout Evaluate3DPose_PCP indexes (evalmodel.py:136-157) + the reference's
    numbers on the reference's own tracker output."""
    import scipy.io as scio
    sys.path.insert(0, ref_shims.REF)
    import evalmodel as ref_eval
    from transformation import coco2shelf3D
    F = len(seq['gt3d'])
    pids = sorted({p for w in seq['gt3d'] for p in w})
    actor = np.empty((1, len(pids)), dtype=object)
    gt_arr = np.full((len(pids), F, 14, 3), np.nan)
    for a, p in enumerate(pids):
        fr = np.empty((F, 1), dtype=object)
        for t in range(F):
            if p in seq['gt3d'][t]:
                g = coco2shelf3D(seq['gt3d'][t][p].T.copy())
                fr[t, 0] = g
                gt_arr[a, t] = g
            else:
                fr[t, 0] = np.zeros((0, 0))
        actor[0, a] = fr
    tmp = tempfile.mkdtemp()
    scio.savemat(os.path.join(tmp, 'actorsGT.mat'), {'actor3D': actor})
    preds = {}
    skipped = set(trace['meta.skipped'].tolist())
    rng = np.random.default_rng(7)
    for t in range(F):
        preds[t] = [] if t in skipped else trace['f%d.pts3d' % t] + rng.normal(0, 0.05, trace['f%d.pts3d' % t].shape)
    pred_path = os.path.join(tmp, 'pred.pkl')
    with open(pred_path, 'wb') as f:
        pickle.dump(preds, f)
    eval_ranges = [[3, F]]
    check, table = ref_eval.Evaluate3DPose_PCP(eval_ranges, pred_path, gt_path=tmp, dataset_name='Shelf')
    rows = [[str(c) for c in r] for r in table]
    np.savez_compressed(out_path, gt=gt_arr, eval_ranges=np.array(eval_ranges, dtype=np.int32),
                        check_result=check, table=np.array(rows),
                        n_frames=np.int32(F), skipped=np.array(sorted(skipped), dtype=np.int32),
                        **{'pred.%d' % t: np.asarray(preds[t], dtype=np.float64).reshape(-1, 3, 17) for t in range(F)})


def make_panoptic(out_path):
    """Synthetic hdPose3d_stage1_coco19/*.json + predictions -> the numbers the reference's EvaluatePanoptic prints
    (evalmodel.py:208-350; it returns nothing, so stdout is captured)."""
    import io, json, contextlib
    sys.path.insert(0, ref_shims.REF)
    import evalmodel as ref_eval
    rng = np.random.default_rng(11)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'hdPose3d_stage1_coco19'))
    M = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
    Minv = np.linalg.inv(M)
    n_files, interval = 60, 12
    preds, gt_raw = {}, {}
    for i in range(n_files):
        ts = 1000 + i
        nb = int(rng.integers(0, 4)) if i % interval == 0 else 2
        bodies, pred = [], []
        for b in range(nb):
            coco = rng.normal(0, 0.5, (3, 17)) + np.array([[b * 1.5], [0.3], [1.0]])              # metres, (3,17)
            p = coco.T * 1000.0
            pelvis = (p[11] + p[12]) / 2
            p14 = np.insert(p[[0, 5, 7, 9, 11, 13, 15, 6, 8, 10, 12, 14, 16]], 3, pelvis).reshape(-1, 3)
            noise = rng.normal(0, [5.0, 20.0, 60.0, 120.0][int(rng.integers(4))], p14.shape)       # mm
            gt14 = p14 + noise
            raw = np.zeros((19, 4))
            raw[1:15, :3] = (gt14 / 10.0) @ Minv
            raw[1:15, 3] = rng.uniform(0.0, 1.0, 14)
            raw[3, 3] = 0.9 if rng.uniform() > 0.15 else 0.05                                       # joints_vis[2] gate
            bodies.append({'id': b, 'joints19': raw.reshape(-1).tolist()})
            if rng.uniform() > 0.1:
                pred.append(coco)
        if i % interval == 0 and nb > 0 and rng.uniform() > 0.5:
            pred.append(rng.normal(0, 0.5, (3, 17)) + np.array([[9.0], [0.3], [1.0]]))             # a false positive
        with open(os.path.join(tmp, 'hdPose3d_stage1_coco19', 'body3DScene_%08d.json' % ts), 'w') as f:
            json.dump({'bodies': bodies}, f)
        preds[ts] = np.array(pred).reshape(-1, 3, 17)
        gt_raw[ts] = np.array([b['joints19'] for b in bodies]).reshape(-1, 19, 4)
    pred_path = os.path.join(tmp, 'pred.pkl')
    with open(pred_path, 'wb') as f:
        pickle.dump(preds, f)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_eval.EvaluatePanoptic([[0, n_files]], pred_path, dataset='Panoptic', seqs=[], data_root=tmp)
    text = buf.getvalue()
    lines = [l for l in text.splitlines() if l.strip()]
    ap_row = [l for l in lines if l.startswith("['AP'")][0]
    rec_row = [l for l in lines if l.startswith("['Recall'")][0]
    mp = [l for l in lines if l.startswith('MPJPE')][0]
    aps = [float(x.strip(" '")) for x in ap_row.strip('[]').split(',')[1:]]
    recs = [float(x.strip(" '")) for x in rec_row.strip('[]').split(',')[1:]]
    mpjpe = float(mp.split(':')[1].replace('mm', ''))
    flat = {'n_files': np.int32(n_files), 'aps': np.array(aps), 'recs': np.array(recs), 'mpjpe': np.float64(mpjpe)}
    for ts in preds:
        flat['pred.%d' % ts] = preds[ts]
        flat['gt.%d' % ts] = gt_raw[ts]
    np.savez_compressed(out_path, **flat)
    print('panoptic golden: AP', aps, 'recall', recs, 'MPJPE', mpjpe)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--only-panoptic', action='store_true')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if args.only_panoptic:
        make_panoptic(os.path.join(args.out, 'panoptic_eval.npz'))
        return
    install_recorders()
    plans = [
        ('S1', 160, 0, dict(blank_frames=(60,))),
        ('S2', 160, 0, {}),
        # round 5: the two hardest rigs at >= 80 frames with a track death + a birth every 22 / 20 frames (churn: one person invisible for 12 frames > MAX_AGE), denser occlusions, empty views and blank frames
        ('S3', 132, 1, dict(birth_death_frame=30, occlusion_every=9, empty_view_every=7, blank_frames=(40, 41, 77), churn_every=22, churn_len=12)),
        ('S4', 100, 2, dict(birth_death_frame=12, occlusion_every=5, empty_view_every=4, blank_frames=(33,), churn_every=20, churn_len=12)),
    ]
    for size, nf, seed, kw in plans:
        REC.clear()
        COUNT.clear()
        seq, cam_rec, trace, model = run_trace(size, nf, seed, **kw)
        np.savez_compressed(os.path.join(args.out, 'cameras_%s.npz' % size), **cam_rec)
        np.savez_compressed(os.path.join(args.out, 'trace_%s.npz' % size), **trace)
        save_ops(os.path.join(args.out, 'ops_%s.npz' % size))
        n_out = sum(len(trace['f%d.ids' % t]) for t in range(nf) if ('f%d.ids' % t) in trace)
        print(size, 'frames', nf, 'emitted poses', n_out, 'tracks now', len(model.tracker.tracks),
              {k: len(v) for k, v in REC.items()})
        if size == 'S2':
            make_pcp(seq, trace, os.path.join(args.out, 'pcp_S2.npz'))
    make_panoptic(os.path.join(args.out, 'panoptic_eval.npz'))
    print('done ->', args.out)


if __name__ == '__main__':
    main()
