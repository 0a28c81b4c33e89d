"""Evaluation gate of the drivers: PCP (Campus / Shelf) -- the reference's ``Evaluate3DPose_PCP``
(/root/reference/src/evalmodel.py:120-206), ``coco2shelf3D`` (src/eval/transformation.py:5-39) and
``vectorize_distance`` (src/eval/numeric.py:5-25) restated on the host in NumPy (evaluation is not per-frame work; SURVEY 2
row 10).  Pinned by tests/golden/pcp_S2.npz, produced by running the reference's function on a synthetic actorsGT.mat."""
import os
import pickle
from collections import OrderedDict

import numpy as np

# Shelf-14 order: r_ankle r_knee r_hip l_hip l_knee l_ankle r_wrist r_elbow r_shoulder l_shoulder l_elbow l_wrist (bottom/top head)
_COCO_TO_SHELF = np.array([16, 14, 12, 11, 13, 15, 10, 8, 6, 5, 7, 9])
_LIMBS = [(0, 1), (1, 2), (3, 4), (4, 5), (6, 7), (7, 8), (9, 10), (10, 11), (12, 13)]
BONE_GROUPS = OrderedDict([('Head', [8]), ('Torso', [9]), ('Upper arms', [5, 6]), ('Lower arms', [4, 7]),
                           ('Upper legs', [1, 2]), ('Lower legs', [0, 3])])


def coco2shelf3D(coco_pose):
    """(3,17) COCO joints -> (14,3) Shelf joints; head bottom / top extrapolated from the shoulders' midpoint and the nose."""
    c = np.asarray(coco_pose, dtype=np.float64).T
    out = np.zeros((14, 3))
    out[:12] = c[_COCO_TO_SHELF]
    mid = (out[8] + out[9]) / 2
    out[13] = mid + (c[0] - mid) * np.array([0.78, 0.5, 1.5])
    out[12] = mid + (c[0] - mid) * np.array([0.3, 0.4, 0.6])
    return out


def closest_prediction(gt_pose, model_poses):
    """Index of the predicted pose with the smallest squared distance to ``gt_pose`` (NaN joints of a prediction ignored)."""
    g = gt_pose.reshape(1, -1)
    d = []
    for p in model_poses:
        p = p.reshape(1, -1)
        ok = ~np.isnan(p)
        gg, pp = g[ok].reshape(1, -1), p[ok].reshape(1, -1)
        d.append(float(np.sum(gg ** 2) + np.sum(pp ** 2) - 2 * (gg @ pp.T)[0, 0]) / ok.size)
    return int(np.argmin(d))


def _limb_ok(ms, me, gs, ge, alpha=0.5):
    return (np.linalg.norm(gs - ms) + np.linalg.norm(ge - me)) / 2 <= alpha * np.linalg.norm(ge - gs)


def evaluate_pcp(eval_ranges, multi_poses3d, actors_gt, verbose=True):
    """actors_gt[pid][frame] -> (14,3) array or an empty array; multi_poses3d[frame] -> (n,3,17) or [].
    Returns (check_result (frames, actors, 10) in {1,-1,0}, table rows like the reference's ``list_tb``)."""
    n_act = len(actors_gt)
    n_frames = len(actors_gt[0])
    check = np.zeros((n_frames, n_act, 10), dtype=np.int32)
    for start, end in eval_ranges:
        for f in range(start, end):
            poses = np.asarray(multi_poses3d[f], dtype=np.float64)
            for a in range(n_act):
                gt = actors_gt[a][f]
                if gt is None or np.size(gt) == 0:
                    continue
                if len(poses) == 0:
                    check[f, a, :] = -1
                    if verbose:
                        print('Cannot get any pose in frame:{}'.format(f))
                    continue
                cand = np.stack([coco2shelf3D(p) for p in poses])
                m = cand[closest_prediction(np.asarray(gt), cand)]
                for k, (s, e) in enumerate(_LIMBS):
                    check[f, a, k] = 1 if _limb_ok(m[s], m[e], gt[s], gt[e]) else -1
                check[f, a, 9] = 1 if _limb_ok((m[2] + m[3]) / 2, m[12], (gt[2] + gt[3]) / 2, gt[12]) else -1
    rows = [['Bone Group'] + ['Actor {}'.format(i) for i in range(3)] + ['Average']]
    with np.errstate(invalid='ignore', divide='ignore'):
        for name, idx in BONE_GROUPS.items():
            v = np.sum(check[:, :, idx] > 0, axis=(0, 2)) / np.sum(np.abs(check[:, :, idx]), axis=(0, 2))
            rows.append([name] + [float('%.2f' % (x * 100)) for x in v[:3]] + [float('%.2f' % (np.sum(v[:3]) * 100 / len(v[:3])))])
        v = np.sum(check > 0, axis=(0, 2)) / np.sum(np.abs(check), axis=(0, 2))
        rows.append(['Total'] + [float('%.2f' % (x * 100)) for x in v[:3]] + [float('%.2f' % (np.sum(v[:3]) * 100 / len(v[:3])))])
    if verbose:
        print(format_table(rows))
    return check, rows


def format_table(rows):
    w = [max(len(str(r[c])) for r in rows) for c in range(len(rows[0]))]
    line = '+' + '+'.join('-' * (x + 2) for x in w) + '+'
    out = [line]
    for i, r in enumerate(rows):
        out.append('| ' + ' | '.join(str(v).ljust(w[c]) for c, v in enumerate(r)) + ' |')
        if i == 0:
            out.append(line)
    out.append(line)
    return '\n'.join(out)


def load_actors_gt(gt_path):
    """actorsGT.mat in the layout the reference indexes: actor3D[0][pid][frame][0] -> (14,3) or empty."""
    import scipy.io as scio
    raw = scio.loadmat(os.path.join(gt_path, 'actorsGT.mat'))['actor3D'][0]
    return [[raw[p][f][0] for f in range(len(raw[p]))] for p in range(len(raw))]


def Evaluate3DPose_PCP(eval_ranges, pred_path, gt_path='CatchImage/CampusSeq1', dataset_name='CampusSeq1'):
    with open(pred_path, 'rb') as f:
        multi_poses3d = pickle.load(f)
    return evaluate_pcp(eval_ranges, multi_poses3d, load_actors_gt(gt_path))


def Write3DResult(multi_poses3d, filepath):
    os.makedirs(os.path.dirname(filepath) or '.', exist_ok=True)
    with open(filepath, 'wb') as f:
        pickle.dump(multi_poses3d, f)


# ---- Panoptic AP / recall / MPJPE (evalmodel.py:208-350), restated -------------------------------------------------------
_PANOPTIC_M = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
_COCO_TO_PANOPTIC13 = [0, 5, 7, 9, 11, 13, 15, 6, 8, 10, 12, 14, 16]


def panoptic_gt_from_bodies(bodies_joints19):
    """(n,19,4) raw joints19 of one annotation file -> (poses (k,14,3) in mm, vis (k,14) bool), bodies whose mid-hip is not
    visible dropped (evalmodel.py:228-248)."""
    poses, vis = [], []
    for j19 in bodies_joints19:
        p = np.asarray(j19, dtype=np.float64).reshape(-1, 4)[1:15].copy()
        v = p[:, -1] > 0.1
        if not v[2]:
            continue
        poses.append(p[:, :3].dot(_PANOPTIC_M) * 10.0)
        vis.append(v)
    return poses, vis


def load_panoptic_gt(data_root, interval=12):
    """{timestamp: (poses, vis)} from hdPose3d_stage1_coco19/body3DScene_*.json, every `interval`-th file."""
    import glob
    import json
    out = {}
    files = sorted(glob.glob(os.path.join(data_root, 'hdPose3d_stage1_coco19', '*.json')))
    for i, fn in enumerate(files):
        if i % interval:
            continue
        with open(fn) as f:
            bodies = json.load(f)['bodies']
        if len(bodies) == 0:
            continue
        ts = int(os.path.basename(fn)[:-5].replace('body3DScene_', ''))
        out[ts] = panoptic_gt_from_bodies([b['joints19'] for b in bodies])
    return out


def evaluate_panoptic(gts, preds, thresholds=(25, 50, 75, 100, 125, 150)):
    """gts {ts: (poses, vis)}, preds {ts: (n,3,17) metres} -> (APs, recalls, MPJPE@500, recall@500)."""
    ev, total_gt = [], 0
    for ts, (poses, vis) in gts.items():
        if len(poses) == 0:
            continue
        for pose in preds[ts]:
            p = np.asarray(pose, dtype=np.float64).T * 1000.0
            pelvis = (p[11] + p[12]) / 2
            p14 = np.insert(p[_COCO_TO_PANOPTIC13], 3, pelvis).reshape(-1, 3)
            errs = [np.mean(np.sqrt(np.sum((p14[v] - g[v]) ** 2, axis=-1))) for g, v in zip(poses, vis)]
            k = int(np.argmin(errs))
            ev.append((float(errs[k]), total_gt + k))
        total_gt += len(poses)

    def ap_at(th):
        tp, fp, seen = np.zeros(len(ev)), np.zeros(len(ev)), set()
        for i, (e, gid) in enumerate(ev):
            if e < th and gid not in seen:
                tp[i] = 1; seen.add(gid)
            else:
                fp[i] = 1
        tp, fp = np.cumsum(tp), np.cumsum(fp)
        recall = tp / (total_gt + 1e-5)
        prec = tp / (tp + fp + 1e-5)
        for n in range(len(ev) - 2, -1, -1):
            prec[n] = max(prec[n], prec[n + 1])
        prec = np.concatenate(([0], prec, [0]))
        recall = np.concatenate(([0], recall, [1]))
        idx = np.where(recall[1:] != recall[:-1])[0]
        return float(np.sum((recall[idx + 1] - recall[idx]) * prec[idx + 1])), float(recall[-2])
    aps, recs = zip(*[ap_at(t) for t in thresholds])
    seen, errs = set(), []
    for e, gid in ev:
        if e < 500 and gid not in seen:
            errs.append(e); seen.add(gid)
    mpjpe = float(np.mean(errs)) if errs else float('inf')
    rec500 = len({gid for e, gid in ev if e < 500}) / total_gt if total_gt else 0.0
    return list(aps), list(recs), mpjpe, rec500


def EvaluatePanoptic(eval_ranges, pred_path, dataset='Panoptic', seqs=(), data_root='CatchImage/Panoptic/160906_pizza1'):
    with open(pred_path, 'rb') as f:
        preds = pickle.load(f)
    aps, recs, mpjpe, _ = evaluate_panoptic(load_panoptic_gt(data_root), preds)
    rows = [['Threshold/mm'] + [str(t) for t in (25, 50, 75, 100, 125, 150)], ['AP'] + ['%.2f' % (a * 100) for a in aps],
            ['Recall'] + ['%.2f' % (r * 100) for r in recs]]
    print(format_table(rows))
    print('MPJPE: %.2fmm' % mpjpe)
    return aps, recs, mpjpe
