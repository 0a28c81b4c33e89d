"""GPU: the image-side HIP kernels (crop/resize/normalise, heat-map decode) against plain torch float32 restatements,
the bf16 conv stack against its own fp32 form, and the on-device pipeline decode -> tracker against the oracle."""
import os

import numpy as np
import pytest
import torch

import pam
from pam import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def net():
    from pam import hrnet
    return hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False)


def test_preprocess_vs_torch(net):
    from pam import hrnet
    dev = net.device
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (3, 288, 360, 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frames[i].data_ptr() for i in range(3)], dtype=torch.int64, device=dev)
    view_of = torch.tensor([0, 2, 1, 2], dtype=torch.int32, device=dev)
    boxes = torch.tensor([[10, 20, 100, 200], [-15.5, -8.25, 120, 260], [300, 200, 90, 120], [0, 0, 360, 288]],
                         dtype=torch.float32, device=dev)          # includes boxes leaving the frame (border replicate)
    x = net.input_buffer(4)
    assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.bfloat16
    net.preprocess(ptrs, 288, 360, view_of, boxes, x)
    ref = hrnet.reference_preprocess(frames, view_of, boxes, (384, 288))
    torch.cuda.synchronize()
    assert x.shape[1] == 8 and float(x[:, 3:].float().abs().max()) == 0.0      # RGB + 5 zero channels for the MFMA conv
    assert (x[:, :3].float() - ref).abs().max().item() <= 2.0 ** -7 * 2.7 + 1e-3       # one bf16 ulp at |v| <= 2.7
    x3 = torch.empty((4, 3, 384, 288), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    net.preprocess(ptrs, 288, 360, view_of, boxes, x3)                          # 3-channel form (MIOpen backend)
    assert torch.equal(x3, x[:, :3])


@pytest.mark.parametrize('layout', ['nhwc', 'nchw'])
def test_decode_vs_torch(net, layout):
    from pam import hrnet
    dev = net.device
    g = torch.Generator().manual_seed(1)
    n = 5
    hm = torch.randn((n, 17, 96, 72), generator=g).to(dev)
    hm[0, 3] = 0.25                                   # all-equal map: first index must win
    hm[1, 5, 10, 11] = 9.0; hm[1, 5, 40, 2] = 9.0     # exact tie: smaller flat index wins
    hm[2, 0, 95, 71] = 50.0                           # last cell
    if layout == 'nhwc':
        hm = hm.contiguous(memory_format=torch.channels_last)
    view_of = torch.tensor([0, 0, 1, 2, 2], dtype=torch.int32, device=dev)
    slot_of = torch.tensor([0, 1, 0, 0, 1], dtype=torch.int32, device=dev)
    boxes = (torch.rand((n, 4), generator=g) * 300 + 20).to(dev)
    det = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
    kp = torch.zeros((n, 17, 3), dtype=torch.float32, device=dev)
    net.decode(hm, view_of, slot_of, boxes, det, kp)
    exp = hrnet.reference_decode(hm, boxes)
    torch.cuda.synchronize()
    for i in range(n):
        got = det[int(view_of[i]), int(slot_of[i])]
        assert torch.equal(got, exp[i]), (i, (got - exp[i]).abs().max())
        assert torch.equal(kp[i][:, 0].double(), exp[i][:, 1]) and torch.equal(kp[i][:, 1].double(), exp[i][:, 0])
    assert float(det[0, 0, 3, 0]) == float(boxes[0, 1]) and float(det[0, 0, 3, 2]) == 0.25
    assert float(det[0, 1, 5, 0]) == float(np.float32(10 / 96 * float(boxes[1, 3]) + float(boxes[1, 1])))


def test_bf16_stack_vs_fp32(net):
    """Self-consistency of the conv stack (parity with the authors' backend is unpinned): bf16 channels-last folded-BN
    heat-maps vs the same weights in fp32 eager; reports arg-max drift."""
    from pam import hrnet
    dev = net.device
    ref = hrnet.fold_batchnorm(hrnet.init_random(hrnet.PoseHighResolutionNet(), seed=0)).to(dev).eval()
    g = torch.Generator().manual_seed(2)
    x32 = torch.randn((2, 3, 384, 288), generator=g).to(dev)
    x8 = torch.cat([x32, torch.zeros((2, 5, 384, 288), device=dev)], dim=1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        h32 = ref(x32.to(torch.bfloat16).float())
        hb = net.heatmaps(x8)
        from torch_ref import bf16_torch_heatmaps                    # PyTorch-ROCm's own bf16 convolutions on the same folded weights
        hm = bf16_torch_heatmaps(x32.to(torch.bfloat16))
    rel = ((hb.float() - h32).norm() / h32.norm()).item()
    rel_mi = ((hm.float() - h32).norm() / h32.norm()).item()
    drift = (hb.float().flatten(2).argmax(2) != h32.flatten(2).argmax(2)).float().mean().item()
    print('rel err vs fp32: hip %.4f miopen %.4f; arg-max drift %.3f' % (rel, rel_mi, drift))
    assert rel < 0.03, rel
    assert rel <= 1.5 * rel_mi + 5e-3
    # the recorded drift figure (bench.py's `hrnet_drift`): bounded, and no worse than PyTorch-ROCm's own bf16 convs on the same input
    d = hrnet.measure_bf16_drift(net, n_crops=2, seed=2)
    drift_mi = (hm.float().flatten(2).argmax(2) != h32.flatten(2).argmax(2)).float().mean().item()
    print('hrnet_drift', d, 'miopen bf16 drift %.3f' % drift_mi)
    assert abs(d['rel_l2_err'] - rel) < 1e-6
    assert d['rel_l2_err'] < 0.03 and d['score_max_abs_err'] < 0.05 * float(h32.flatten(2).max(2)[0].abs().max())
    assert d['argmax_moved_frac'] <= max(2.0 * drift_mi, 0.15), (d, drift_mi)     # random weights: near-flat maps move easily
    # the figures that bound something: a joint whose fp32 peak stands > 8 error sigmas above the rest of its map stays within one cell,
    # and a planted peak of 16 error sigmas (the same cell raised in both heat-maps) decodes to the same cell
    assert d['decided']['8']['max_cells'] <= 1, d
    assert d['planted_peak_max_cells']['16'] == 0, d
    dm = hrnet.drift_statistics(h32, hm.float())                                  # PyTorch-ROCm's own bf16 convolutions: the same order of error
    assert dm['decided']['8']['max_cells'] <= 1 and d['rms_err'] <= 2.0 * dm['rms_err'] + 1e-4, (dm, d)


def test_hipgraph_replay_matches_eager():
    from pam import hrnet
    a = hrnet.HRNetPose(48, 17, None, use_graph=False)
    b = hrnet.HRNetPose(48, 17, None, use_graph=True)
    x = a.input_buffer(3)
    x.copy_(torch.randn(x.shape, device=x.device).to(x.dtype))
    y0 = a.heatmaps(x).clone()
    y1 = b.heatmaps(x).clone()
    y2 = b.heatmaps(x).clone()                      # replay
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(y0, y1)          # the MFMA conv kernels are deterministic


def test_device_pipeline_decode_to_tracker_vs_oracle():
    """Heat-maps with planted peaks -> k_decode -> det buffer (device) -> k_frame, no host round trip; the oracle is fed the
    torch-decoded keypoints.  ids / view sets bit-exact, 3D <= 1e-6 m."""
    from pam import hrnet
    from pam.ivclabpose import Camera, fundamental_matrices
    from pam.pipeline import FramePipeline
    from oracle import cpu_ref as O
    size = 'S2'
    meta = synth.SIZES[size]
    seq = synth.make_sequence(size, n_frames=30, seed=4, occlusion_every=7, empty_view_every=11, birth_death_frame=15)
    cfg = dict(synth.MATCHER_CFG['Shelf']); conf = cfg.pop('CONF_THRESHOLD')
    P32 = seq['calib']['P'].astype(np.float32); K32 = seq['calib']['K'].astype(np.float32); RT32 = seq['calib']['RT'].astype(np.float32)
    Fm = fundamental_matrices(K32, RT32)
    cams = [Camera(j, P32[j], K32[j], RT32[j], Fm[j]) for j in range(meta['C'])]
    pipe = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=8, max_tracks=16, hrnet=False)
    pipe.net = hrnet.HRNetPose(48, 17, None, use_graph=False, max_dets=8)
    dev = pipe.device
    ref = O.OracleIvclabpose(cfg, conf)
    ref.GetCameraParameters(seq['calib'], F=Fm)
    emitted = 0
    for t, views in enumerate(seq['frames']):
        vl, sl, bx, cells = [], [], [], []
        for v, dets in enumerate(views):
            for s, kp in enumerate(dets):
                x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
                b = [x0 - 0.125 * (x1 - x0), y0 - 0.125 * (y1 - y0), 1.25 * (x1 - x0), 1.25 * (y1 - y0)]
                vl.append(v); sl.append(s); bx.append(b)
                px = np.clip(np.round((kp[:, 0] - b[0]) / b[2] * 72), 0, 71).astype(int)
                py = np.clip(np.round((kp[:, 1] - b[1]) / b[3] * 96), 0, 95).astype(int)
                cells.append((py, px, kp[:, 2]))
        n = len(vl)
        hm = torch.zeros((n, 17, 96, 72), dtype=torch.float32)
        for i, (py, px, sc) in enumerate(cells):
            hm[i, torch.arange(17), torch.tensor(py), torch.tensor(px)] = torch.tensor(sc, dtype=torch.float32)
        hm = hm.to(dev).contiguous(memory_format=torch.channels_last)
        view_of = torch.tensor(vl, dtype=torch.int32, device=dev); slot_of = torch.tensor(sl, dtype=torch.int32, device=dev)
        boxes = torch.tensor(bx, dtype=torch.float32, device=dev).reshape(n, 4)
        pipe.det_local.zero_()
        if n:
            pipe.net.decode(hm, view_of, slot_of, boxes, pipe.det_local)
        n_det = torch.tensor([len(d) for d in views], dtype=torch.int32, device=dev)
        pipe.track_step(t, n_det, pipe.det_local)
        got = pipe.results()
        # oracle on the torch-decoded keypoints
        dec = hrnet.reference_decode(hm, boxes).cpu().numpy() if n else np.zeros((0, 17, 3))
        per_view = [[] for _ in views]
        for i in range(n):
            per_view[vl[i]].append(dec[i])
        if n == 0:
            continue
        ref.tracker.step(t, [np.array(p).reshape(-1, 17, 3) for p in per_view])
        _, _, _, p3, jv, ids = ref.tracker.collect(t)
        em = [tr for tr in got['tracks'] if tr['emitted']]
        assert [tr['track_id'] for tr in em] == list(ids), t
        for tr, e3, ejv in zip(em, p3, jv):
            np.testing.assert_allclose(tr['pose3d'], e3.T, rtol=0, atol=1e-6)
            out = [[] for _ in range(tr['V'])]
            for j in range(17):
                out[int(tr['nviews'][j]) - 1].append(j)
            assert out == ejv
        emitted += len(em)
    assert emitted > 20


def test_predict_dump_format_and_facade_end_to_end():
    """HRNetPose.predict through the reference-shaped facade: person_bbox_list (BGR frames + xywh boxes) -> dump_results with
    the keys ivclabpose.PersonTrack_Project3DPose reads (ivclabpose.py:233-246), keypoints inside their boxes, and the
    result accepted by the tracker step."""
    from pam.ivclabpose import ivclabpose
    pose_cfg = dict(NAME='HRPose', C=48, NUM_JOINTS=17, CHECKPOINT_FILE='', MODEL_NAME='HRNet', RESOLUTION=[384, 288])
    mcfg = dict(synth.MATCHER_CFG['CampusSeq1']); conf = mcfg.pop('CONF_THRESHOLD')
    model = ivclabpose({'NAME': ''}, pose_cfg, dict(mcfg, NAME='Iterative'), conf)
    seq = synth.make_sequence('S1', n_frames=2, seed=0)
    model.GetCameraParameters(seq['calib'], 288, 360)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (288, 360, 3), dtype=np.uint8) for _ in range(3)]
    boxes = [[[20.0, 30.0, 100.0, 200.0], [150.5, 40.25, 90.0, 180.0]], [], [[200.0, 60.0, 120.0, 210.0]]]
    pbl = [[dict(image_id=0, category_id=1, score=0.9, bbox=b, data=frames[v], feature=[]) for b in bs] for v, bs in enumerate(boxes)]
    dump = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=20)
    assert [len(v) for v in dump] == [2, 0, 1]
    for v, items in enumerate(dump):
        for it, b in zip(items, boxes[v]):
            assert set(it) >= {'bbox', 'keypoints', 'keypoints_score', 'feature'}
            k = np.array(it['keypoints']).reshape(17, 3)
            assert len(it['keypoints_score']) == 17 and np.allclose(k[:, 2], it['keypoints_score'])
            assert (k[:, 0] >= b[0] - 1e-3).all() and (k[:, 0] <= b[0] + b[2]).all()       # x inside the box
            assert (k[:, 1] >= b[1] - 1e-3).all() and (k[:, 1] <= b[1] + b[3]).all()       # y inside the box
    dump2 = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=2)       # chunked batches: same result
    for a, b in zip(dump, dump2):
        for ia, ib in zip(a, b):
            assert ia['keypoints'] == ib['keypoints']
    G = model.pose_model._graphs
    assert model.pose_model.graph_bucket == 4 and (4, 'features', 0) in G and (3, 'features', 0) not in G   # 3 crops ran padded to 4
    model.pose_model.graph_bucket = 1                                                         # exact batch sizes: same result
    dump3 = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=20)
    assert (3, 'features', 0) in model.pose_model._graphs
    for a, b in zip(dump, dump3):
        for ia, ib in zip(a, b):
            assert ia['keypoints'] == ib['keypoints']
    out = model.PersonTrack_Project3DPose(0, pbl, dump, 'SVD')
    assert len(out) == 9 and len(out[5]) == 0          # nothing confirmed on the first frame


def test_head_kernel_vs_torch_fp32():
    """pam_head_heatmaps (final 1x1 conv, float32 FMA over bf16 features) against F.conv2d in float32, incl. a ragged last workgroup."""
    import ctypes as C
    import torch.nn.functional as F
    from pam import _lib
    g = torch.Generator().manual_seed(4)
    for (n, h, w) in [(3, 96, 72), (1, 7, 5)]:
        f = torch.randn((n, 48, h, w), generator=g).to(torch.bfloat16).to('cuda:0').contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((17, 48), generator=g) * 0.2).to('cuda:0'); b = torch.randn(17, generator=g).to('cuda:0')
        out = torch.full((n, 17, h, w), 9.0, dtype=torch.float32, device='cuda:0').contiguous(memory_format=torch.channels_last)
        rc = _lib.load().pam_head_heatmaps(None, n * h * w, C.c_void_p(f.data_ptr()), 48, C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()),
                                           17, C.c_void_p(out.data_ptr()))
        assert rc == 0
        ref = F.conv2d(f.float(), wt.reshape(17, 48, 1, 1), b)
        torch.cuda.synchronize()
        assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-5


def test_head_decode_fused_vs_two_kernels(net):
    """pam_head_decode (1x1 head + arg-max in one pass, heat-maps optional) against pam_head_heatmaps + the torch arg-max reference:
    identical heat-map values, identical keypoint rows, np.argmax's first-index rule on exact ties (engineered through the weights:
    an all-zero head makes every map constant = its bias), ragged last tile, crops that are not the whole feature batch."""
    import ctypes as C
    from pam import hrnet, _lib
    dev = net.device
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    for (n, h, w) in [(5, 96, 72), (3, 7, 5), (2, 33, 17)]:
        f = torch.randn((n, 48, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
        # features with planted exact ties: two pixels of crop 1 get identical feature vectors that dominate joint 5
        f[1 % n, :, 1, 2] = f[1 % n, :, 0, 1]
        wt = (torch.randn((17, 48), generator=g) * 0.2).to(dev); b = torch.randn(17, generator=g).to(dev)
        wt[3] = 0.0                                                    # joint 3: constant map -> index 0 must win
        view_of = torch.arange(n, dtype=torch.int32, device=dev) % 3
        slot_of = torch.arange(n, dtype=torch.int32, device=dev) // 3
        boxes = (torch.rand((n, 4), generator=g) * 300 + 20).to(dev)
        ref_hm = torch.empty((n, 17, h, w), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
        assert lib.pam_head_heatmaps(None, n * h * w, C.c_void_p(f.data_ptr()), 48, C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), 17,
                                     C.c_void_p(ref_hm.data_ptr())) == 0
        exp = hrnet.reference_decode(ref_hm, boxes)
        old_w, old_b = net.head_w, net.head_b
        net.head_w, net.head_b = wt.contiguous(), b.contiguous()
        try:
            for with_heat in (True, False):
                det = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
                kp = torch.zeros((n, 17, 3), dtype=torch.float32, device=dev)
                heat = torch.full_like(ref_hm, 7.0) if with_heat else None
                net.head_decode(f, view_of, slot_of, boxes, det, kp, heat=heat)
                torch.cuda.synchronize()
                if with_heat:
                    assert torch.equal(heat, ref_hm)
                for i in range(n):
                    got = det[int(view_of[i]), int(slot_of[i])]
                    assert torch.equal(got, exp[i]), (n, h, w, i, (got - exp[i]).abs().max())
                    assert torch.equal(kp[i][:, 0].double(), exp[i][:, 1]) and torch.equal(kp[i][:, 1].double(), exp[i][:, 0])
                assert float(det[0, 0, 3, 0]) == float(boxes[0, 1]) and float(det[0, 0, 3, 2]) == float(b[3])
            # only the first m crops
            det2 = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
            net.head_decode(f, view_of[:1].contiguous(), slot_of[:1].contiguous(), boxes[:1].contiguous(), det2, n=1)
            torch.cuda.synchronize()
            assert torch.equal(det2[0, 0], exp[0]) and float(det2[1:].abs().max()) == 0.0
        finally:
            net.head_w, net.head_b = old_w, old_b


@pytest.mark.gpu
def test_soft_argmax_decode_vs_torch(net):
    """pam_head_decode_soft: per joint the softmax(beta * heat-map)-weighted mean (column, row), mapped through the box like the hard
    decode; confidence = the maximum.  Against float64 torch on the heat-maps the same pass writes (sub-pixel positions to 1e-3 px of
    the heat-map grid), over ragged last tiles and several betas; a large beta reproduces the hard arg-max keypoints; a constant map
    gives the centre of the heat-map."""
    dev = net.device
    g = torch.Generator().manual_seed(23)
    old = (net.head_w, net.head_b, net.soft_beta)
    try:
        for (n, h, w) in [(4, 96, 72), (3, 7, 5), (2, 33, 17)]:
            f = torch.randn((n, 48, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
            wt = (torch.randn((17, 48), generator=g) * 0.2).to(dev); b = torch.randn(17, generator=g).to(dev)
            wt[3] = 0.0                                                # joint 3: constant map
            net.head_w, net.head_b = wt.contiguous(), b.contiguous()
            view_of = torch.arange(n, dtype=torch.int32, device=dev) % 3
            slot_of = torch.arange(n, dtype=torch.int32, device=dev) // 3
            boxes = (torch.rand((n, 4), generator=g) * 300 + 20).to(dev)
            ys = torch.arange(h, dtype=torch.float64, device=dev)[:, None].expand(h, w).reshape(-1)
            xs = torch.arange(w, dtype=torch.float64, device=dev)[None, :].expand(h, w).reshape(-1)
            for beta in (0.7, 4.0, 25.0):
                net.soft_beta = beta
                det = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
                kp = torch.zeros((n, 17, 3), dtype=torch.float32, device=dev)
                heat = torch.zeros((n, 17, h, w), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
                net.head_decode(f, view_of, slot_of, boxes, det, kp, heat=heat)
                torch.cuda.synchronize()
                hm = heat.double().reshape(n, 17, h * w)
                p = torch.softmax(beta * hm, dim=2)
                ey, ex = (p * ys).sum(2), (p * xs).sum(2)                  # (n, 17) expected row / column
                for i in range(n):
                    bx = boxes[i].double()
                    want_y = ey[i] / h * bx[3] + bx[1]; want_x = ex[i] / w * bx[2] + bx[0]
                    got = det[int(view_of[i]), int(slot_of[i])]
                    tol_y, tol_x = 1e-3 * float(bx[3]) / h + 1e-4, 1e-3 * float(bx[2]) / w + 1e-4
                    assert float((got[:, 0] - want_y).abs().max()) <= tol_y and float((got[:, 1] - want_x).abs().max()) <= tol_x, (n, h, w, beta, i)
                    assert torch.equal(got[:, 2].float(), hm[i].max(1).values.float())
                    assert torch.allclose(kp[i][:, 0].double(), got[:, 1], atol=1e-4) and torch.allclose(kp[i][:, 1].double(), got[:, 0], atol=1e-4)
                # constant map: centre of the grid
                got3 = det[0, 0, 3]
                assert abs(float(got3[0]) - ((h - 1) / 2.0 / h * float(boxes[0, 3]) + float(boxes[0, 1]))) < 1e-2
            # large beta -> hard arg-max, wherever the maximum stands clear of the runner-up (weight of everything else < e^-40)
            net.soft_beta = 4.0e4
            det_s = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
            net.head_decode(f, view_of, slot_of, boxes, det_s)
            net.soft_beta = None
            det_h = torch.zeros((3, 4, 17, 3), dtype=torch.float64, device=dev)
            net.head_decode(f, view_of, slot_of, boxes, det_h)
            torch.cuda.synchronize()
            top2 = hm.topk(2, dim=2).values
            clear = (top2[:, :, 0] - top2[:, :, 1]) > 1e-3                 # (n, 17)
            assert int(clear.sum()) > n * 8
            for i in range(n):
                a_, b_ = det_s[int(view_of[i]), int(slot_of[i])], det_h[int(view_of[i]), int(slot_of[i])]
                assert float((a_ - b_)[clear[i]].abs().max()) < 1e-3, (n, h, w, i)
    finally:
        net.head_w, net.head_b, net.soft_beta = old


def test_preprocess_hd_frames_with_boxes_leaving_the_frame(net):
    """k_preprocess_crops at the Panoptic frame size (1920 x 1080, S3 / S4 workloads), boxes clipped by every border and one covering the
    whole frame, against the torch float32 restatement."""
    from pam import hrnet
    dev = net.device
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (3, 1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frames[i].data_ptr() for i in range(3)], dtype=torch.int64, device=dev)
    boxes = torch.tensor([[-40.5, -25.25, 300, 420],            # leaves top-left
                          [1700, 800, 400, 500],                # leaves bottom-right
                          [900.3, -60, 210.7, 380],             # leaves the top only
                          [-10, 500, 180, 700],                 # left and bottom
                          [0, 0, 1920, 1080],                   # the whole frame (down-scaling 5x / 2.8x)
                          [1000.5, 400.25, 96.5, 130.75],       # small box (up-scaling)
                          [1919, 1079, 30, 30]],                # starts on the last pixel
                         dtype=torch.float32, device=dev)
    view_of = torch.tensor([0, 1, 2, 0, 1, 2, 0], dtype=torch.int32, device=dev)
    x = net.input_buffer(7)
    net.preprocess(ptrs, 1080, 1920, view_of, boxes, x)
    ref = hrnet.reference_preprocess(frames, view_of, boxes, (384, 288))
    torch.cuda.synchronize()
    assert float(x[:, 3:].float().abs().max()) == 0.0
    assert (x[:, :3].float() - ref).abs().max().item() <= 2.0 ** -7 * 2.7 + 1e-3       # one bf16 ulp at |v| <= 2.7


def test_predict_s3_sized_call_through_the_graph_buckets():
    """One Panoptic-5-like call of the drop-in predict(): 5 HD views x 7 persons = 35 crops -> batches of 20 + 15 (padded to 16 by the
    graph bucket), hipGraph replays.  A 2-crop subset (one from each batch) is checked against (a) the same kernels driven by hand --
    k_preprocess_crops -> eager conv stack -> torch decode: identical keypoints -- and (b) reference_preprocess + the fp32 PyTorch module
    with the same weights: confidences close, positions equal except for the recorded bf16 drift."""
    from pam import hrnet
    net = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=True, max_dets=8, graph_bucket=4)
    dev = net.device
    g = torch.Generator().manual_seed(11)
    frames = [torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(5)]
    rng = np.random.default_rng(3)
    pbl = []
    for v in range(5):
        persons = []
        for p in range(7):
            w, h = rng.uniform(120, 320), rng.uniform(300, 620)
            x0, y0 = rng.uniform(-30, 1920 - w + 30), rng.uniform(-30, 1080 - h + 30)
            persons.append(dict(image_id=0, category_id=1, score=0.9, bbox=[float(x0), float(y0), float(w), float(h)], data=frames[v], feature=[]))
        pbl.append(persons)
    dump = net.predict(pbl, batch_size=20)
    assert [len(d) for d in dump] == [7] * 5 and sorted(k[0] for k in net._graphs) == [16, 20]
    assert dump.device_valid() and tuple(dump.device_det.shape) == (5, 8, 17, 3)
    pick = [(0, 3), (4, 5)]                                       # crop 3 (first batch) and crop 33 (second, padded batch)
    view_of = torch.tensor([v for v, _ in pick], dtype=torch.int32, device=dev)
    boxes = torch.tensor([pbl[v][p]['bbox'] for v, p in pick], dtype=torch.float32, device=dev)
    ptrs = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64, device=dev)
    eager = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False, max_dets=8)
    x = eager.input_buffer(2)
    eager.preprocess(ptrs, 1080, 1920, view_of, boxes, x)
    with torch.no_grad():
        exp = hrnet.reference_decode(eager.heatmaps(x), boxes)    # rows (y, x, score)
        ref = hrnet.fold_batchnorm(hrnet.init_random(hrnet.PoseHighResolutionNet(), seed=0)).to(dev).eval()
        x32 = hrnet.reference_preprocess(torch.stack(frames), view_of, boxes, (384, 288))
        exp32 = hrnet.reference_decode(ref(x32), boxes)
    torch.cuda.synchronize()
    moved = 0
    for i, (v, p) in enumerate(pick):
        kp = np.asarray(dump[v][p]['keypoints']).reshape(17, 3)   # (x, y, score)
        got = torch.tensor(np.stack([kp[:, 1], kp[:, 0], kp[:, 2]], axis=1))
        assert torch.equal(got, exp[i].cpu()), (i, (got - exp[i].cpu()).abs().max())
        assert torch.equal(dump.device_det[v, p].cpu(), exp[i].cpu())
        assert np.array_equal(np.asarray(dump[v][p]['keypoints_score']), kp[:, 2])
        e32 = exp32[i].cpu()
        assert (got[:, 2] - e32[:, 2]).abs().max() <= 0.05 * float(e32[:, 2].abs().max()) + 1e-3
        moved += int(((got[:, :2] - e32[:, :2]).abs().sum(1) > 0).sum())
    assert moved <= 0.35 * 34, moved                              # random weights: near-flat heat-maps (bench.py `hrnet_drift` records the rate)


@pytest.fixture(scope='module')
def eager_and_replay():
    """ONE eager and ONE replaying network for all configurations and crop counts of the test below (a construction packs 63 M weights:
    24 of them were 280 s of the round-5 suite)."""
    from pam import hrnet
    return hrnet.HRNetPose(48, 17, None, use_graph=False), hrnet.HRNetPose(48, 17, None, use_graph=True)


@pytest.mark.parametrize('n', [1, 5, 20])
def test_replay_equals_eager_forward_in_every_configuration(n, eager_and_replay):
    """The multi-stream forward as a hipGraph replay computes exactly what the same launches compute issued eagerly (three replays: a
    missing cross-stream dependency shows up as run-to-run differences), in both executor configurations; the two configurations are
    the same network up to bf16 summation order (the fused 96-channel block adds the residual before the products)."""
    from pam import hrnet_hip
    a, b = eager_and_replay
    outs = {}
    saved = b.__dict__.get('config_for')
    try:
        for name in hrnet_hip.HipHRNet.CONFIGS:
            a.hip.apply_config(name)
            b.config_for = lambda n, name=name: name
            # the replay cache is keyed by crop count: forget the previous configuration's captures (kept alive, never destroyed)
            b._dead_graphs.extend(v for v in b._graphs.values()); b._dead_graphs.extend(v for v in b._alt.values() if v is not None)
            b._graphs.clear(); b._alt.clear(); b.flag_synced.clear(); b.flag_timing.clear()
            x = a.input_buffer(n)
            x.copy_(torch.randn(x.shape, generator=torch.Generator().manual_seed(n)).to(x.device).to(x.dtype)); x[:, 3:] = 0
            ref = a.features(x).clone()
            for _ in range(3):
                y = b.features(x).clone()
                torch.cuda.synchronize()
                assert torch.equal(ref, y), name
            assert b.hip.config_name == name
            outs[name] = ref.float()
    finally:
        if saved is None:
            b.__dict__.pop('config_for', None)
        else:
            b.config_for = saved
    v = list(outs.values())
    assert float((v[0] - v[1]).norm() / v[0].norm()) < 1e-2
    # the 32-channel-slab form of the deep branches' layers is the same arithmetic in the same order: bit-identical
    assert torch.equal(outs['fused48_fused96_fsum'], outs['fused48_fused96_fsum_s32'])


def test_bf16_stack_vs_fp32_at_the_256x192_resolution():
    """The reference's other pose input size (256 x 192 crops -> 64 x 48 maps, every branch a different tiling of the fused kernels):
    heat-maps against the same folded weights in fp32, replay against eager, as at 384 x 288."""
    from pam import hrnet
    net = hrnet.HRNetPose(48, 17, None, resolution=(256, 192), use_graph=True)
    dev = net.device
    ref = hrnet.fold_batchnorm(hrnet.init_random(hrnet.PoseHighResolutionNet(), seed=0)).to(dev).eval()
    g = torch.Generator().manual_seed(4)
    x32 = torch.randn((3, 3, 256, 192), generator=g).to(dev)
    x8 = torch.cat([x32, torch.zeros((3, 5, 256, 192), device=dev)], dim=1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        h32 = ref(x32.to(torch.bfloat16).float())
        hb = net.heatmaps(x8).clone()
        eager = hrnet.HRNetPose(48, 17, None, resolution=(256, 192), use_graph=False)
        he = eager.heatmaps(x8)
    torch.cuda.synchronize()
    assert tuple(hb.shape) == (3, 17, 64, 48)
    assert torch.equal(hb, he)
    rel = ((hb.float() - h32).norm() / h32.norm()).item()
    assert rel < 0.03, rel


@pytest.mark.parametrize('n', [3, 20])
def test_fused_head_is_bit_identical_to_the_launches_it_replaces(n):
    """k_stem_fused (stem + first conv1) and k_bneck (3x3 + pointwise tail of every layer1 Bottleneck) keep the operand order of the
    kernels they replace: the whole network's features are bit-identical with each of them switched off (eager forwards of one network
    object, flags toggled)."""
    from pam import hrnet
    a = hrnet.HRNetPose(48, 17, None, use_graph=False)
    x = a.input_buffer(n)
    x.copy_(torch.randn(x.shape, generator=torch.Generator().manual_seed(40 + n)).to(x.device).to(x.dtype)); x[:, 3:] = 0
    assert a.hip.fuse_stem and a.hip.fuse_bneck and a.hip.fuse_bneck0
    ref = a.features(x).clone()
    for flags in (dict(fuse_stem=False), dict(fuse_bneck=False), dict(fuse_bneck0=False), dict(fuse_stem=False, fuse_bneck=False)):
        for k, v in flags.items():
            setattr(a.hip, k, v)
        y = a.features(x).clone()
        torch.cuda.synchronize()
        for k in flags:
            setattr(a.hip, k, True)
        assert torch.equal(ref, y), flags


def test_full_pipeline_panoptic31_sized_frame():
    """Config #5's frames on ONE GPU through the drop-in surface: 31 HD views x 7 persons = 217 crops -> PersonPoseDetect (eleven batches
    of 20 through the replay, the last padded) -> PersonTrack_Project3DPose on the device-resident keypoints (k_frame<1024>: more than 8
    views), four frames.  The network has random weights, so -- exactly as bench.py does -- the decoded keypoints are checked
    for plumbing (every crop in its (view, slot), inside its box, stable across calls) and then REPLACED on the device by the seeded
    synthetic ones; the tracker's output on them must equal the oracle's (ids, 3D <= 1e-6 m)."""
    from pam import hrnet, synth
    from pam.ivclabpose import ivclabpose
    from oracle import cpu_ref as O
    nf = 4
    seq = synth.make_sequence('S4', n_frames=nf, seed=0)
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET['S4']]); conf = cfg.pop('CONF_THRESHOLD')
    model = ivclabpose({'NAME': ''}, None, dict(cfg, NAME='Iterative'), conf, max_dets=8, max_tracks=16)
    cams = model.GetCameraParameters(seq['calib'], 1920, 1080)
    ref = O.OracleIvclabpose(cfg, conf)
    ref.GetCameraParameters(seq['calib'], F=np.stack([c.F for c in cams]))
    model.pose_model = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=True, max_dets=8)
    dev = model.pose_model.device
    g = torch.Generator().manual_seed(3)
    base = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev)
    frames = [torch.roll(base, shifts=17 * v, dims=1).contiguous() for v in range(31)]
    n_det_all, det_all = synth.pack_frames(seq['frames'], 8)
    emitted = 0
    for t in range(nf):
        pbl = []
        for v in range(31):
            persons = []
            for kp in seq['frames'][t][v]:
                x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
                persons.append(dict(image_id=t, category_id=1, score=0.9, data=frames[v], feature=[],
                                    bbox=[float(x0 - 0.125 * (x1 - x0)), float(y0 - 0.125 * (y1 - y0)), float(1.25 * (x1 - x0)), float(1.25 * (y1 - y0))]))
            pbl.append(persons)
        dump = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=20)
        assert [len(d) for d in dump] == [int(n) for n in n_det_all[t]] and sum(len(d) for d in dump) == 217
        assert tuple(dump.device_det.shape) == (31, 8, 17, 3) and dump.device_valid()
        assert sorted(k[0] for k in model.pose_model._graphs) == [20]          # 10 x 20 crops + 17 padded to 20
        if t == 0:
            for v in range(31):
                for it in dump[v]:
                    k = np.array(it['keypoints']).reshape(17, 3); b = it['bbox']
                    assert (k[:, 0] >= b[0] - 1e-2).all() and (k[:, 0] <= b[0] + b[2] + 1e-2).all()
                    assert (k[:, 1] >= b[1] - 1e-2).all() and (k[:, 1] <= b[1] + b[3] + 1e-2).all()
            dump2 = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=20)
            for a_, b_ in zip(dump, dump2):
                for ia, ib in zip(a_, b_):
                    assert ia['keypoints'] == ib['keypoints']
        dump.device_det.copy_(torch.tensor(det_all[t], dtype=torch.float64, device=dev))          # synthetic keypoints instead of the random net's
        dump.poses_host = [det_all[t][v][:n_det_all[t][v]] for v in range(31)]
        out = model.PersonTrack_Project3DPose(t, pbl, dump, 'SVD')
        assert model.tracker.last['status'] == 0 and model.tracker.last['status_sticky'] == 0
        pbl_o, dr_o = synth.to_dump_results(seq['frames'][t])
        exp = ref.PersonTrack_Project3DPose(t, pbl_o, dr_o, 'SVD')
        assert list(out[5]) == list(exp[5])
        if len(out[5]):
            assert np.abs(np.asarray(out[3]) - np.asarray(exp[3])).max() < 1e-6
            emitted += len(out[5])
    assert emitted >= 7


def test_configuration_follows_the_crop_count_and_stays_consistent():
    """HRNetPose(autotune=True): the executor configuration of a replay follows its crop count by a fixed rule (config_for); the result
    is the same network (equal to the default configuration up to bf16 summation-order noise), replays are deterministic, and a second
    replay slot of the same crop count is the same configuration."""
    from pam import hrnet, hrnet_hip
    a = hrnet.HRNetPose(48, 17, None, use_graph=True)
    b = hrnet.HRNetPose(48, 17, None, use_graph=True, autotune=True)
    x = a.input_buffer(9)
    x.copy_(torch.randn(x.shape, device=x.device).to(x.dtype)); x[:, 3:] = 0
    ref = a.features(x).float().clone()
    y1 = b.features(x).clone()
    y2 = b.features(x).clone()
    y3 = b.features(x, slot=1).clone()
    torch.cuda.synchronize()
    assert b.tuned[9]['choice'] == 'fused48_fused96_fsum_s32' and a.tuned[9]['choice'] == hrnet_hip.HipHRNet.config_name == 'fused48_fused96'
    # up to 12 crops: fused sums + 32-channel slabs in the deep branches; up to 20: fused sums (round 5)
    assert b.config_for(4) == b.config_for(12) == 'fused48_fused96_fsum_s32' and b.config_for(13) == b.config_for(20) == 'fused48_fused96_fsum'
    assert b.config_for(21) == 'fused48_fused96'
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    assert float((y1.float() - ref).norm() / ref.norm()) < 1e-2
    x24 = b.input_buffer(24)
    x24.copy_(torch.randn(x24.shape, device=x24.device).to(x24.dtype)); x24[:, 3:] = 0
    b.features(x24); torch.cuda.synchronize()
    assert b.tuned[24]['choice'] == 'fused48_fused96' and set(c['choice'] for c in b.tuned.values()) <= set(hrnet_hip.HipHRNet.CONFIGS)


def test_preprocess_bucket_padding_repeats_the_last_crop(net):
    """A replay bucket larger than the call: the crop kernel fills the spare rows with the last crop (no padded box table on the host)."""
    dev = net.device
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (2, 288, 360, 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frames[i].data_ptr() for i in range(2)], dtype=torch.int64, device=dev)
    view_of = torch.tensor([0, 1, 1], dtype=torch.int32, device=dev)
    boxes = torch.tensor([[10, 20, 100, 200], [30, 40, 120, 160], [200, 100, 90, 120]], dtype=torch.float32, device=dev)
    x3, x8 = net.input_buffer(3).clone(), net.input_buffer(8).clone()
    net.preprocess(ptrs, 288, 360, view_of, boxes, x3)
    net.preprocess(ptrs, 288, 360, view_of, boxes, x8)
    torch.cuda.synchronize()
    assert torch.equal(x8[:3], x3) and all(torch.equal(x8[k], x3[2]) for k in range(3, 8))


def test_preprocess_antialias_matches_the_pil_formula(net):
    """The optional resize of upstream simple-HRNet (PIL image through torchvision Resize): support-scaled triangle filter.  Checked
    against torch's antialiased bilinear interpolation of the integer-aligned crop (the same ImagingResample formula in float32), for
    boxes that are down-scaled (HD frames), up-scaled, and mixed; and equal to the plain bilinear form where nothing is down-scaled."""
    import torch.nn.functional as F
    from pam import hrnet
    dev = net.device
    g = torch.Generator().manual_seed(6)
    frames = torch.randint(0, 256, (2, 1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frames[i].data_ptr() for i in range(2)], dtype=torch.int64, device=dev)
    view_of = torch.tensor([0, 1, 0, 1], dtype=torch.int32, device=dev)
    boxes = torch.tensor([[100, 50, 600, 900], [700, 0, 432, 1080], [400, 300, 150, 200], [1500, 600, 400, 300]], dtype=torch.float32, device=dev)
    aa = hrnet.HRNetPose(48, 17, None, use_graph=False, antialias=True)
    x, xp = net.input_buffer(4).clone(), net.input_buffer(4).clone()
    aa.preprocess(ptrs, 1080, 1920, view_of, boxes, x)
    net.preprocess(ptrs, 1080, 1920, view_of, boxes, xp)
    torch.cuda.synchronize()
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    for i in range(4):
        bx, by, bw, bh = [int(v) for v in boxes[i].tolist()]
        crop = frames[int(view_of[i]), by:by + bh, bx:bx + bw].permute(2, 0, 1)[[2, 1, 0]].float().unsqueeze(0)      # BGR -> RGB
        ref = (F.interpolate(crop, size=(384, 288), mode='bilinear', antialias=True, align_corners=False) / 255.0 - mean) / std
        err = (x[i:i + 1, :3].float() - ref).abs().max().item()
        assert err <= 2.0 ** -7 * 2.7 + 2e-2, (i, err)
        down = bw > 288 or bh > 384
        d = (x[i, :, 4:-4, 4:-4].float() - xp[i, :, 4:-4, 4:-4].float()).abs().max().item()
        # where nothing is down-scaled the two modes are the same interpolation in the crop's interior (to a bf16 rounding: another
        # evaluation order; at the crop's border the plain form reads the frame pixels just outside the box, the PIL form -- which resizes
        # the cut-out crop -- does not); a down-scaled box of random pixels is visibly smoother
        assert (d > 0.2) if down else (d <= 2.0 ** -7 * 2.7 + 1e-3), (i, down, d)
    assert float(x[:, 3:].float().abs().max()) == 0.0


