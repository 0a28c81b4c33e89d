"""Host-side driver surface: PCP evaluator vs the reference's numbers (golden), config / dataset plumbing, and (GPU) the
evalmodel.py loop on a synthetic on-disk dataset."""
import os
import pickle

import numpy as np
import pytest

import pam
from pam import synth
from pam.dataset import GetConfig, LoadFilenames, LoadImages, natural_key
from pam import evaluation as E
import golden_io as G


def _pcp_case():
    z = G.load('pcp_S2.npz')
    gt = z['gt']
    actors = [[(None if np.isnan(gt[a, f]).all() else gt[a, f]) for f in range(gt.shape[1])] for a in range(gt.shape[0])]
    preds = {t: z['pred.%d' % t] for t in range(int(z['n_frames']))}
    for t in z['skipped']:
        preds[int(t)] = []
    return z, actors, preds


def test_pcp_matches_reference():
    z, actors, preds = _pcp_case()
    check, rows = E.evaluate_pcp(z['eval_ranges'].tolist(), preds, actors, verbose=False)
    assert np.array_equal(check, z['check_result'])
    ref_rows = z['table']
    assert 0 < (check < 0).sum() < (check > 0).sum()             # a non-trivial case: some limbs fail
    for r, rr in zip(rows[1:], ref_rows[1:]):
        assert r[0] == str(rr[0])
        for a, b in zip(r[1:], rr[1:]):
            assert abs(float(a) - float(b)) < 1e-9, (r, rr)


def test_panoptic_eval_matches_reference():
    z = G.load('panoptic_eval.npz')
    n = int(z['n_files'])
    gts, preds = {}, {}
    for i in range(n):
        ts = 1000 + i
        preds[ts] = z['pred.%d' % ts]
        if i % 12 == 0 and len(z['gt.%d' % ts]):
            gts[ts] = E.panoptic_gt_from_bodies(z['gt.%d' % ts])
    aps, recs, mpjpe, _ = E.evaluate_panoptic(gts, preds)
    assert [round(a * 100, 2) for a in aps] == z['aps'].tolist()
    assert [round(r * 100, 2) for r in recs] == z['recs'].tolist()
    assert round(mpjpe, 2) == float(z['mpjpe'])
    assert 0 < aps[0] < aps[-1] < 1                   # a non-trivial case


def test_coco2shelf_layout():
    p = np.arange(51, dtype=float).reshape(3, 17)
    s = E.coco2shelf3D(p)
    assert s.shape == (14, 3)
    assert np.array_equal(s[0], p[:, 16]) and np.array_equal(s[11], p[:, 9])
    mid = (p[:, 6] + p[:, 5]) / 2
    np.testing.assert_allclose(s[12], mid + (p[:, 0] - mid) * np.array([0.3, 0.4, 0.6]))


def test_configs_parse_and_keys():
    root = os.path.join(pam.PKG_DIR, 'configs')
    ref = {'CampusSeq1': (25, 15, 15, 30, 0.6, 0.4), 'Shelf': (60, 30, 60, 70, 0.3, 0.5), 'Panoptic': (60, 50, 30, 60, 0.3, 0.4)}
    for name, (epi, ini, jt, a2d, sig, conf) in ref.items():
        c = GetConfig(os.path.join(root, name, 'model_configs.yaml'))
        m = c.PERSON_MATCHERS.ITERATIVE
        assert (m.EPI_THRESHOLD, m.INIT_THRESHOLD, m.JOINT_THRESHOLD, m.ALPHA2D, m.SIGMA) == (epi, ini, jt, a2d, sig)
        assert c.PIPELINE_COMBINATION.CONF_THRESHOLD == conf and m.N_INIT == 3 and m.MAX_AGE == 10
        assert c.POSE_MODELS.HRPOSE.RESOLUTION == [384, 288] and c.TEST_FUNCTION == 'PersonTrack_Project3DPose'
        assert dict(synth.MATCHER_CFG[name], NAME='Iterative', CONF_THRESHOLD=conf).items() >= {k: v for k, v in m.items() if k in synth.MATCHER_CFG[name]}.items()


def test_natural_sort_and_bgr(tmp_path):
    from PIL import Image
    names = ['img10.png', 'img2.png', 'img1.png']
    assert sorted(names, key=natural_key) == ['img1.png', 'img2.png', 'img10.png']
    for cam in ('Camera0', 'Camera1'):
        os.makedirs(tmp_path / cam)
        for i, n in enumerate(names):
            a = np.zeros((4, 6, 3), dtype=np.uint8); a[..., 0] = 200; a[..., 2] = 10 + i     # RGB: R=200, B=10+i
            Image.fromarray(a).save(tmp_path / cam / n)
    ds = pam.dataset.AttrDict(dict(ROOT=str(tmp_path), FOLDERS_ORDER=['Camera0', 'Camera1'], DATA_FORMAT='*.png'))
    files = LoadFilenames(ds)
    assert [os.path.basename(f[0]) for f in files] == ['img1.png', 'img2.png', 'img10.png']
    imgs, ts = LoadImages('Shelf', files[0])
    assert ts == 'img1' and imgs[0].shape == (4, 6, 3) and imgs[0][0, 0, 0] == 12 and imgs[0][0, 0, 2] == 200   # BGR


def test_frame_loader_matches_loadimages(tmp_path):
    from PIL import Image
    from pam.ingest import FrameLoader
    rng = np.random.default_rng(0)
    for cam in ('Camera0', 'Camera1', 'Camera2'):
        os.makedirs(tmp_path / cam)
        for i in range(12):
            Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).save(tmp_path / cam / ('%03d.jpg' % i), quality=90)
    ds = pam.dataset.AttrDict(dict(ROOT=str(tmp_path), FOLDERS_ORDER=['Camera0', 'Camera1', 'Camera2'], DATA_FORMAT='*.jpg'))
    files = LoadFilenames(ds)
    loader = FrameLoader('Shelf', files, indices=range(2, 11), workers=4, depth=3)
    seen = []
    for idx, imgs, ts in loader:
        ref, ts_ref = LoadImages('Shelf', files[idx])
        assert ts == ts_ref and len(imgs) == 3
        for a, b in zip(imgs, ref):
            assert a.dtype == np.uint8 and np.array_equal(a, b)
        seen.append(idx)
    loader.close()
    assert seen == list(range(2, 11))                 # order preserved


@pytest.mark.gpu
def test_frame_loader_device_path_matches_the_host_path(tmp_path):
    """The workers decode straight into pinned staging buffers that are reused every depth + 1 frames; the uploaded tensors equal the host
    decode for every frame of a run longer than the ring, incl. cameras of different sizes."""
    import torch
    from PIL import Image
    from pam.ingest import FrameLoader
    rng = np.random.default_rng(1)
    sizes = {'Camera0': (24, 32), 'Camera1': (24, 32), 'Camera2': (30, 20)}
    for cam, (h, w) in sizes.items():
        os.makedirs(tmp_path / cam)
        for i in range(23):
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / cam / ('%03d.jpg' % i), quality=90)
    ds = pam.dataset.AttrDict(dict(ROOT=str(tmp_path), FOLDERS_ORDER=list(sizes), DATA_FORMAT='*.jpg'))
    files = LoadFilenames(ds)
    dev = torch.device('cuda:0')
    loader = FrameLoader('Shelf', files, workers=4, depth=3, device=dev)
    kept, seen = [], []
    for idx, imgs, ts in loader:
        kept.append(imgs)                             # hold on to every frame: later uploads must not overwrite earlier tensors
        seen.append(idx)
    loader.close()
    torch.cuda.synchronize()
    assert seen == list(range(23))
    for idx, imgs in zip(seen, kept):
        ref, _ = LoadImages('Shelf', files[idx])
        for a, b in zip(imgs, ref):
            assert a.is_cuda and a.dtype == torch.uint8 and np.array_equal(a.cpu().numpy(), b)


@pytest.mark.gpu
def test_evalmodel_loop_on_synthetic_dataset(tmp_path, capsys):
    """evalmodel.py's loop end to end: images on disk, calibration pickle, precomputed 2D poses -> result pickle identical to
    the oracle's output, PCP table printed."""
    from PIL import Image
    import scipy.io as scio
    from oracle import cpu_ref as O
    from pam import evalmodel
    seq = synth.make_sequence('S1', n_frames=40, seed=11, occlusion_every=9, empty_view_every=13, birth_death_frame=20)
    root = tmp_path / 'CampusSeq1'
    for c in range(3):
        os.makedirs(root / ('Camera%d' % c))
        for t in range(40):
            Image.fromarray(np.zeros((8, 10, 3), dtype=np.uint8)).save(root / ('Camera%d' % c) / ('%04d.png' % t))
    with open(root / 'camera_parameter.pickle', 'wb') as f:
        pickle.dump(seq['calib'], f)
    pre = {}
    for t, views in enumerate(seq['frames']):
        _, dr = synth.to_dump_results(views)
        pre[t] = [[dict(bbox=d['bbox'], keypoints=d['keypoints'], keypoints_score=d['keypoints_score']) for d in v] for v in dr]
    with open(root / 'detections.pickle', 'wb') as f:
        pickle.dump(pre, f)
    pids = sorted({p for w in seq['gt3d'] for p in w})
    actor = np.empty((1, len(pids)), dtype=object)
    for a, p in enumerate(pids):
        fr = np.empty((40, 1), dtype=object)
        for t in range(40):
            fr[t, 0] = E.coco2shelf3D(seq['gt3d'][t][p].T) if p in seq['gt3d'][t] else np.zeros((0, 0))
        actor[0, a] = fr
    scio.savemat(str(root / 'actorsGT.mat'), {'actor3D': actor})
    cfg = GetConfig(os.path.join(pam.PKG_DIR, 'configs', 'CampusSeq1', 'model_configs.yaml'))
    cfg.DATASET.ROOT = str(root); cfg.DATASET.DATA_FORMAT = '*.png'; cfg.DATASET.TEST_RANGE = [0, 40]
    cfg.DATASET.EVAL_RANGE = [[3, 40]]; cfg.OUTPUT = str(tmp_path / 'out')
    cfg.PIPELINE_COMBINATION.POSE_MODEL = 'Precomputed'
    evalmodel.eval_ivclabpose_PersonTrack_Project3DPose(cfg, LoadFilenames(cfg.DATASET))
    out = capsys.readouterr().out
    assert 'Total' in out and 'tracking fps' in out
    with open(tmp_path / 'out' / 'CampusSeq1' / 'logs' / 'None_Precomputed_Iterative_CampusSeq1.pkl', 'rb') as f:
        res = pickle.load(f)
    mc = dict(synth.MATCHER_CFG['CampusSeq1']); conf = mc.pop('CONF_THRESHOLD')
    ref = O.OracleIvclabpose(mc, conf)
    ref.GetCameraParameters(seq['calib'])
    n = 0
    for t, views in enumerate(seq['frames']):
        pbl, dr = synth.to_dump_results(views)
        if not any(len(v) for v in dr):
            assert len(res[t]) == 0
            continue
        e3 = ref.PersonTrack_Project3DPose(t, pbl, dr, 'SVD')[3]
        assert np.asarray(res[t]).shape == np.asarray(e3).shape, t
        if len(e3):
            np.testing.assert_allclose(res[t], e3, rtol=0, atol=1e-5)
            n += len(e3)
    assert n > 30


def test_overlay_draws_confident_joints_only():
    from pam.visualization import joints_dict, draw_points_and_skeleton
    jd = joints_dict()['coco']
    assert len(jd['keypoints']) == 17 and len(jd['skeleton']) == 19 and max(max(p) for p in jd['skeleton']) == 16
    img = np.zeros((120, 160, 3), dtype=np.uint8)
    pts = np.zeros((17, 3)); pts[:, 0] = np.linspace(10, 110, 17); pts[:, 1] = np.linspace(20, 140, 17); pts[:, 2] = 0.9
    pts[16, 2] = 0.1                                                       # below threshold: not drawn
    out = draw_points_and_skeleton(img, pts, jd['skeleton'], person_index=3, points_color_palette='gist_rainbow',
                                   skeleton_color_palette='tab20', points_palette_samples=17, confidence_threshold=0.5)
    assert out.shape == img.shape and out.dtype == np.uint8 and not img.any()        # input untouched
    assert out[int(pts[0, 0]), int(pts[0, 1])].any() and out[int(pts[8, 0]), int(pts[8, 1])].any()
    assert not out[int(pts[16, 0]) - 2:int(pts[16, 0]) + 3, int(pts[16, 1]) - 2:int(pts[16, 1]) + 3].any()
    assert tuple(out[int(pts[0, 0]), int(pts[0, 1])]) == (0, 0, 255)               # joint 0 = first rainbow hue (red), BGR


@pytest.mark.gpu
def test_testmodel_full_pipeline_with_detector(tmp_path, capsys):
    """DETECT_MODEL: YOLOv3 + POSE_MODEL: HRPose + Iterative matcher on images from disk (random networks: this checks the
    reference's loop structure, the wiring of every stage and the overlay / SAVE_IMAGE output, not accuracy)."""
    import warnings
    from PIL import Image
    from pam import testmodel
    seq = synth.make_sequence('S1', n_frames=16, seed=2)
    rng = np.random.default_rng(0)
    root = tmp_path / 'CampusSeq1'
    for c in range(3):
        os.makedirs(root / ('Camera%d' % c))
        for t in range(16):
            Image.fromarray(rng.integers(0, 256, (288, 360, 3), dtype=np.uint8)).save(root / ('Camera%d' % c) / ('%04d.png' % t))
    with open(root / 'camera_parameter.pickle', 'wb') as f:
        pickle.dump(seq['calib'], f)
    cfg = GetConfig(os.path.join(pam.PKG_DIR, 'configs', 'CampusSeq1', 'model_configs.yaml'))
    cfg.DATASET.ROOT = str(root); cfg.DATASET.DATA_FORMAT = '*.png'; cfg.DATASET.TEST_RANGE = [0, 16]
    cfg.OUTPUT = str(tmp_path / 'out'); cfg.SAVE_IMAGE = True
    cfg.PIPELINE_COMBINATION.DETECT_MODEL = 'YOLOv3'
    cfg.DETECT_MODELS.YOLOV3.SCORE_THRESH = 0.6
    seen = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        testmodel.test_ivclabpose_PersonTrack_Project3DPose(cfg, LoadFilenames(cfg.DATASET), on_frame=lambda fid, ts, res: seen.append((fid, res)))
    out = capsys.readouterr().out
    assert 'Person Detector :  YOLOv3' in out and 'Person Detect Processing time' in out and 'tracking fps' in out
    assert [f for f, _ in seen] == list(range(16))
    got = [r for _, r in seen if r is not None]
    assert got, 'the random detector found no box in any frame'
    for r in got:
        assert len(r) == 9                                             # the reference's 9-tuple
        camera_ids, pts, person_ids = r[0], r[1], r[2]
        for cids, poses, pids in zip(camera_ids, pts, person_ids):
            assert len(cids) == len(poses) == len(pids)
            for p in poses:
                assert np.asarray(p).shape == (17, 3)
    saved = os.listdir(tmp_path / 'out' / 'CampusSeq1' / 'Images')
    assert any(name.endswith('_0.jpg') for name in saved)


def test_synthetic_sequences_with_track_churn():
    """synth.make_sequence(churn_every, churn_len): one person is invisible in every view for churn_len frames of each period (the longer
    S3 / S4 golden traces use it to see a track death and a birth per period); without churn the sequence is what it always was."""
    import numpy as np
    from pam import synth
    a = synth.make_sequence('S1', n_frames=60, seed=4)
    b = synth.make_sequence('S1', n_frames=60, seed=4, churn_every=20, churn_len=12)
    for t in range(60):                                  # (after the first hidden frame the two random streams differ: only the counts compare)
        na = [len(v) for v in a['frames'][t]]
        nb = [len(v) for v in b['frames'][t]]
        hidden = t >= 20 and (t % 20) < 12
        assert all(0 <= x - y <= (1 if hidden else 0) for x, y in zip(na, nb)), (t, na, nb)
        if t < 20:
            assert all(np.array_equal(va, vb) for va, vb in zip(a['frames'][t], b['frames'][t]))
    assert sum(len(v) for f in b['frames'] for v in f) < sum(len(v) for f in a['frames'] for v in f)
    c = synth.make_sequence('S1', n_frames=60, seed=4)
    assert all(np.array_equal(x, y) for fa, fc in zip(a['frames'], c['frames']) for x, y in zip(fa, fc))
