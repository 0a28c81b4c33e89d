"""GPU: the person-detector side (csrc/pam_detect.hip + the Darknet activation codes of csrc/pam_conv.hip) through the C ABI,
against oracle/yolo_ref.py (NumPy) for the streaming kernels and a plain PyTorch fp32 Darknet for the conv stack."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam
from oracle import yolo_ref as Y

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _engine():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.ConvEngine()
    e.lib = _lib.load(); e.device = torch.device(DEV)
    return e


@pytest.mark.parametrize('case', [
    # n, h, w, cin, cout, k, stride, act, residual ('none' | 'before' | 'after')
    (2, 26, 26, 64, 128, 3, 1, 'leaky', 'after'),       # Darknet residual block tail (k_conv3x3 path)
    (2, 26, 26, 128, 64, 1, 1, 'leaky', 'none'),
    (1, 52, 52, 32, 64, 3, 1, 'leaky', 'after'),        # Cin = 32: generic kernel
    (2, 13, 13, 512, 1024, 3, 1, 'leaky', 'after'),
    (2, 26, 26, 256, 512, 3, 1, 'leaky', 'after'),      # 32-channel slabs
    (2, 52, 52, 128, 256, 3, 1, 'leaky', 'after'),
    (1, 104, 104, 64, 128, 3, 1, 'leaky', 'after'),     # wide rows: one-row tiles on a 3-wave block
    (1, 208, 208, 64, 64, 3, 1, 'leaky', 'after'),      # rows too wide for the patch kernel -> generic kernel
    (2, 13, 13, 1024, 256, 1, 1, 'linear', 'none'),     # head conv (255 -> 256 padded below)
    (2, 30, 22, 8, 64, 3, 1, 'leaky', 'none'),          # first conv on the stem kernel, stride 1
    (2, 31, 23, 8, 32, 3, 1, 'leaky', 'none'),          # Darknet's own first layer: 32 real channels, ragged width
    (2, 31, 23, 8, 32, 3, 2, 'relu', 'none'),
    (2, 30, 22, 64, 64, 3, 2, 'leaky', 'none'),
    (2, 24, 18, 48, 48, 3, 1, 'leaky', 'before'),
    (2, 24, 18, 48, 48, 3, 1, 'relu', 'after'),
])
def test_conv_activation_codes(case):
    from pam import hrnet_hip
    n, h, w, cin, cout, k, stride, act, resmode = case
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(11)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((n, cin, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    res = None
    if resmode != 'none':
        res = torch.randn((n, cout, ho, wo), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    y = _engine().conv(op, x, res=res, relu=act, res_after_act=(resmode == 'after'))
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev), stride, k // 2)
    f = {'leaky': lambda t: F.leaky_relu(t, 0.1), 'relu': torch.relu, 'linear': lambda t: t}[act]
    ref = f(ref) + res.float() if resmode == 'after' else f(ref + res.float() if res is not None else ref)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all()), (case, err.max().item())


def test_padded_output_channels_are_zero_and_real_ones_unchanged():
    from pam import hrnet_hip
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(2)
    conv = nn.Conv2d(64, 255, 1, 1, 0, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1); conv.bias.copy_(torch.randn(255, generator=g))
    x = torch.randn((2, 64, 13, 13), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    y = _engine().conv(hrnet_hip.PackedConv(conv, dev, pad_cout_to=256), x, relu='linear')
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev))
    assert y.shape[1] == 256 and float(y[:, 255:].float().abs().max()) == 0.0
    assert bool(((y[:, :255].float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2e-2).all())


@pytest.mark.parametrize('shape', [(2, 97, 131, 64, 48), (3, 480, 640, 416, 416), (1, 776, 1032, 416, 416), (1, 20, 20, 40, 60)])
def test_resize_frames_vs_oracle(shape):
    from pam import _lib
    n, fh, fw, oh, ow = shape
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (n, fh, fw, 3), dtype=np.uint8)
    d = torch.from_numpy(frames).to(DEV)
    ptrs = torch.tensor([d[i].data_ptr() for i in range(n)], dtype=torch.int64, device=DEV)
    out = torch.full((n, oh, ow, 8), 7.0, dtype=torch.bfloat16, device=DEV)
    rc = _lib.load().pam_resize_frames(None, n, C.c_void_p(ptrs.data_ptr()), fh, fw, oh, ow, C.c_void_p(out.data_ptr()))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.float().cpu().numpy(), Y.resize_frames(frames, oh, ow))       # bit-exact (same fp32 op order)


def test_upsample_concat_vs_oracle():
    e = _engine()
    g = torch.Generator().manual_seed(4)
    for (n, h, w, ca, cb) in [(2, 13, 13, 256, 512), (1, 7, 5, 8, 16), (3, 26, 26, 128, 256)]:
        a = torch.randn((n, ca, h, w), generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
        b = torch.randn((n, cb, 2 * h, 2 * w), generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
        y = e.upsample_concat(a, b)
        exp = Y.upsample_concat(a.permute(0, 2, 3, 1).float().cpu().numpy(), b.permute(0, 2, 3, 1).float().cpu().numpy())
        assert np.array_equal(y.permute(0, 2, 3, 1).float().cpu().numpy(), exp)
    rc = e.lib.pam_upsample_concat_nhwc_bf16(None, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), 1, 7, 6, 8, 8)
    assert rc == -1                                              # odd output height: PAM_E_ARG


def _run_detect(heads_np, anchors, net, nc, cls, st, nt, fw, fh, max_det, split=False):
    from pam import _lib
    n = heads_np[0].shape[0]
    hd = [torch.from_numpy(h).to(torch.bfloat16).to(DEV).contiguous() for h in heads_np]
    hp = (C.c_void_p * 3)(*[C.c_void_p(h.data_ptr()) for h in hd])
    gh = (C.c_int32 * 3)(*[h.shape[1] for h in hd]); gw = (C.c_int32 * 3)(*[h.shape[2] for h in hd]); cs = (C.c_int32 * 3)(*[h.shape[3] for h in hd])
    an = np.ascontiguousarray(anchors.reshape(-1), dtype=np.float32)
    boxes = torch.full((n, max_det, 5), -1.0, dtype=torch.float32, device=DEV)
    count = torch.full((2 * n,), -1, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    if split:
        # the scoring pass over several workgroups per image; the SAME workspace twice: the kernel must leave its tickets zero
        need = lib.pam_yolo_detect_workspace_bytes(n, gh, gw)
        assert need > 0
        ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
        args = (None, n, hp, gh, gw, cs, an.ctypes.data_as(C.c_void_p), net[0], net[1], nc, cls, st, nt, fw, fh, max_det)
        scratch_boxes, scratch_count = torch.empty_like(boxes), torch.empty_like(count)
        assert lib.pam_yolo_detect_ws(*args, C.c_void_p(scratch_boxes.data_ptr()), C.c_void_p(scratch_count.data_ptr()), C.c_void_p(ws.data_ptr()), need) == 0
        rc = lib.pam_yolo_detect_ws(*args, C.c_void_p(boxes.data_ptr()), C.c_void_p(count.data_ptr()), C.c_void_p(ws.data_ptr()), need)
        assert lib.pam_yolo_detect_ws(*args, C.c_void_p(boxes.data_ptr()), C.c_void_p(count.data_ptr()), C.c_void_p(ws.data_ptr()), need - 1) == -1
        assert lib.pam_yolo_detect_ws(*args, C.c_void_p(boxes.data_ptr()), C.c_void_p(count.data_ptr()), None, need) == -1
        torch.cuda.synchronize()
        assert int(ws[:4 * n].view(torch.int32).abs().sum()) == 0
    else:
        rc = lib.pam_yolo_detect(None, n, hp, gh, gw, cs, an.ctypes.data_as(C.c_void_p), net[0], net[1], nc, cls, st, nt, fw, fh,
                                 max_det, C.c_void_p(boxes.data_ptr()), C.c_void_p(count.data_ptr()))
    assert rc == 0
    torch.cuda.synchronize()
    return boxes.cpu().numpy(), count.cpu().numpy(), [h.float().cpu().numpy() for h in hd]


@pytest.mark.parametrize('cfg', [
    # n, grids, nc, chan stride, cls, score_thresh, nms_thresh, max_det, logit sigma
    (3, (13, 26, 52), 80, 256, 0, 0.6, 0.45, 64, 1.5),
    (2, (13, 26, 52), 80, 255, 17, 0.5, 0.3, 8, 1.5),       # max_det cap reached
    (1, (4, 8, 16), 1, 24, 0, 0.2, 0.5, 100, 1.0),           # tiny grids, low threshold: > 1024 candidates are counted, 1024 enter
    (2, (5, 9, 19), 3, 32, 2, 0.999999, 0.45, 16, 1.0),      # nothing passes
])
@pytest.mark.parametrize('split', [False, True])
def test_yolo_detect_vs_oracle(cfg, split):
    from pam import yolov3
    n, grids, nc, cs, cls, st, nt, max_det, sigma = cfg
    rng = np.random.default_rng(3)
    anchors = np.array(yolov3.ANCHORS, dtype=np.float32).reshape(3, 3, 2)[::-1].copy()
    heads = [(rng.standard_normal((n, g, g + 1, cs)) * sigma).astype(np.float32) for g in grids]      # non-square grids
    if cfg[1] == (4, 8, 16):
        for h in heads:
            h[..., 4::(5 + nc)] += 4.0; h[..., 5::(5 + nc)] += 4.0
    boxes, count, hq = _run_detect(heads, anchors, (416, 448), nc, cls, st, nt, 1032, 776, max_det, split)
    for i in range(n):
        exp, nfound = Y.detect([h[i] for h in hq], anchors, 416, 448, nc, cls, st, nt, 1032, 776, max_det)
        assert count[n + i] == nfound, (i, count[n + i], nfound)
        assert count[i] == len(exp), (i, count[i], len(exp))
        got = boxes[i, :count[i]]
        assert np.allclose(got, exp, rtol=2e-5, atol=1e-3), np.abs(got - exp).max()
        assert (boxes[i, count[i]:] == -1.0).all()                       # rows past the count are untouched
    if st > 0.99:
        assert (count[:n] == 0).all()
    if cfg[1] == (4, 8, 16):
        assert count[n] > 1024


def test_darknet_hip_vs_torch_fp32_and_detector_end_to_end():
    """Whole Darknet-53 + heads on the MFMA kernels vs the float32 PyTorch form of the same random network, then the full
    detector call (resize -> net -> decode + NMS, hipGraph replay) against the oracle applied to the kernel's own heads."""
    from pam import yolov3
    det = yolov3.YOLOv3(None, None, None, score_thresh=0.5, nms_thresh=0.45, use_cuda=True, max_det=32, seed=1)
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(2)]
    frames = torch.from_numpy(np.stack(imgs)).to(DEV)
    x8 = torch.from_numpy(Y.resize_frames(np.stack(imgs), 416, 416)).to(DEV).permute(0, 3, 1, 2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    heads = det.net.forward(x8)
    model = yolov3.Darknet().init_random(1).eval().to(DEV)
    with torch.no_grad():
        ref = model(x8[:, :3].float())
    torch.cuda.synchronize()
    for h, r in zip(heads, ref):
        assert h.shape[1] == 256 and float(h[:, 255:].float().abs().max()) == 0.0
        d = (h[:, :255].float() - r)
        rel = float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
        assert rel < 0.03, rel                                   # 75 bf16 layers vs fp32
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                          # a random net may exceed the pre-NMS capacity; the oracle caps alike
        res1 = det(imgs)
        res2 = det(imgs)                                         # graph replay
        single = det(imgs[0])
    assert len(res1) == 2 and all(r.shape[1] == 5 and r.dtype == np.float32 for r in res1)
    for a, b in zip(res1, res2):
        assert np.array_equal(a, b)
    hq = [h.permute(0, 2, 3, 1).float().cpu().numpy() for h in det.net.forward(x8)]
    for i in range(2):
        exp, _ = Y.detect([h[i] for h in hq], det.anchors, 416, 416, 80, 0, 0.5, 0.45, 320, 240, 32)
        assert len(exp) == len(res1[i]) and np.allclose(res1[i], exp, rtol=2e-5, atol=1e-3)
        assert (np.diff(res1[i][:, 4]) <= 0).all()
    assert np.array_equal(single, res1[0])


def test_persondetect_facade_format():
    """ivclabpose.PersonDetect (ivclabpose.py:183-204): per image a list of dicts with clamped xywh boxes and rounded scores."""
    from pam import ivclabpose as IV, dataset
    cfg = dataset.AttrDict(dict(NAME='YOLOv3', CFG=None, WEIGHT=None, CLASS_NAMES=None, SCORE_THRESH=0.5, NMS_THRESH=0.45))
    api = IV.ivclabpose(person_detector=cfg, pose_detector=None, person_matcher=None)
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(3)]
    out = api.PersonDetect(imgs, image_id=7)
    assert len(out) == 3
    for persons, im in zip(out, imgs):
        for p in persons:
            assert p['image_id'] == 7 and p['category_id'] == 1 and p['data'] is im and p['feature'] == []
            x, y, w, h = p['bbox']
            assert x >= 0 and y >= 0 and x + w <= 320 + 1e-3 and y + h <= 240 + 1e-3
            assert isinstance(p['score'], float) and round(p['score'], 4) == p['score']


def test_detection_a_frame_ahead_gives_persondetects_boxes():
    """PersonDetectAhead / PersonDetectResult (not in the reference: PersonDetect split in two so that a driver can issue the next frame's
    detection early): the same dicts as PersonDetect for the same images, also with a second ticket in flight before the first is
    collected (two pinned landing buffers), for NumPy images and for CUDA tensors."""
    import warnings
    from pam import ivclabpose as IV, dataset
    cfg = dataset.AttrDict(dict(NAME='YOLOv3', CFG=None, WEIGHT=None, CLASS_NAMES=None, SCORE_THRESH=0.5, NMS_THRESH=0.45))
    api = IV.ivclabpose(person_detector=cfg, pose_detector=None, person_matcher=None)
    rng = np.random.default_rng(11)
    sets = [[rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(3)] for _ in range(3)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        want = [api.PersonDetect(imgs, image_id=k) for k, imgs in enumerate(sets)]
        t0 = api.PersonDetectAhead(sets[0], 0)
        got = []
        for k in range(3):
            nxt = api.PersonDetectAhead(sets[k + 1], k + 1) if k + 1 < 3 else None      # issued BEFORE frame k is collected
            got.append(api.PersonDetectResult(t0))
            t0 = nxt
        dev = [torch.from_numpy(im).to('cuda:0') for im in sets[1]]
        got_dev = api.PersonDetectResult(api.PersonDetectAhead(dev, 1))
    strip = lambda frames: [[(p['image_id'], p['bbox'], p['score']) for p in v] for v in frames]
    assert [strip(f) for f in got] == [strip(f) for f in want]
    assert strip(got_dev) == strip(want[1]) and all(p['data'] is dev[v] for v, ps in enumerate(got_dev) for p in ps)
    assert sum(len(v) for f in want for v in f) > 0


def test_detect_then_pose_pipeline_runs():
    """DETECT_MODEL: YOLOv3 path of the driver: PersonDetect boxes feed PersonPoseDetect (random weights: shapes / plumbing only)."""
    import warnings
    from pam import ivclabpose as IV, dataset
    dcfg = dataset.AttrDict(dict(NAME='YOLOv3', CFG=None, WEIGHT=None, CLASS_NAMES=None, SCORE_THRESH=0.5, NMS_THRESH=0.4))
    pcfg = dataset.AttrDict(dict(NAME='HRPose', C=48, NUM_JOINTS=17, CHECKPOINT_FILE=None, MODEL_NAME='HRNet', RESOLUTION=[384, 288]))
    api = IV.ivclabpose(person_detector=dcfg, pose_detector=pcfg, person_matcher=None)
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(2)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pbl = api.PersonDetect(imgs, image_id=0)
    pbl = [[p for p in v if p['bbox'][2] > 4 and p['bbox'][3] > 4][:3] for v in pbl]
    dump = api.PersonPoseDetect(person_bbox_list=pbl, batch_size=20)
    assert len(dump) == 2
    for v, persons in zip(dump, pbl):
        assert len(v) == len(persons)
        for d in v:
            assert len(d['keypoints']) == 51 and len(d['keypoints_score']) == 17      # flat (x, y, score) x 17
