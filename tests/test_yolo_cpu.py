"""CPU: the detector's host logic (cfg generator / parser, Darknet .weights I/O, launch plan) and known answers for the
NumPy checker of its post-processing (oracle/yolo_ref.py; parity unpinned -- the reference's backend.YOLOv3 is absent)."""
import numpy as np
import pytest
import torch

import pam
from pam import yolov3
from oracle import yolo_ref as Y


def test_default_cfg_is_standard_yolov3():
    net, layers = yolov3.parse_cfg(yolov3.default_cfg())
    kinds = [l['type'] for l in layers]
    assert len(layers) == 107 and kinds.count('convolutional') == 75 and kinds.count('shortcut') == 23
    assert [i for i, k in enumerate(kinds) if k == 'yolo'] == [82, 94, 106]
    assert layers[86]['layers'] == [-1, 61] and layers[98]['layers'] == [-1, 36]
    assert [l['mask'] for l in layers if l['type'] == 'yolo'] == [[6, 7, 8], [3, 4, 5], [0, 1, 2]]
    assert layers[82]['anchors'][8] == (373.0, 326.0) and int(net['width']) == 416
    m = yolov3.Darknet()
    assert sum(p.numel() for p in m.parameters()) == 61949149           # the published YOLOv3-416 parameter count (incl. BN)


def test_parse_cfg_comments_and_unsupported():
    txt = "[net]\nwidth=64 # px\nheight=32\n\n# stem\n[convolutional]\nfilters=16\nsize=3\nstride=2\npad=1\nactivation=leaky\n"
    net, layers = yolov3.parse_cfg(txt)
    assert net['width'] == '64' and layers[0]['filters'] == 16 and layers[0]['batch_normalize'] == 0 and layers[0]['stride'] == 2
    with pytest.raises(NotImplementedError):
        yolov3.parse_cfg("[net]\n[maxpool]\nsize=2\n")
    with pytest.raises(ValueError):
        yolov3.parse_cfg("[convolutional]\nfilters=1\n")


def test_darknet_weights_roundtrip(tmp_path):
    a = yolov3.Darknet().init_random(3)
    p = str(tmp_path / 'w.weights')
    a.save_darknet_weights(p)
    b = yolov3.Darknet()
    b.load_darknet_weights(p)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        if 'num_batches_tracked' in ka:
            continue
        assert ka == kb and torch.equal(va, vb), ka
    with open(p, 'ab') as f:
        f.write(b'\0\0\0\0')
    with pytest.raises(ValueError):
        b.load_darknet_weights(p)


def test_launch_plan_on_meta():
    class _NoLib(object):
        def __getattr__(self, name):
            return lambda *a, **k: 0
    real_load = pam._lib.load
    pam._lib.load = lambda: _NoLib()
    try:
        hd = yolov3.HipDarknet(yolov3.Darknet().init_random(0), torch.device('meta'))
    finally:
        pam._lib.load = real_load
    kinds = [s[0] for s in hd.plan]
    assert kinds.count('conv') == 75 and kinds.count('upcat') == 2 and kinds.count('add') == 0 and kinds.count('head') == 3
    assert sum(1 for s in hd.plan if s[0] == 'conv' and s[5] is not None) == 23          # every shortcut fused into its conv
    assert hd.padded[0] == 32 and hd.real[0] == 32 and hd.padded[2] == 64 and hd.real[2] == 32          # stem kernel writes 32 real channels
    assert hd.padded[81] == 256 and hd.real[81] == 255
    hd.count = dict(bytes=0, flops=0, launches=0)
    x8 = torch.empty((2, 8, 416, 416), dtype=torch.bfloat16, device='meta').contiguous(memory_format=torch.channels_last)
    heads = hd.forward(x8)
    assert [tuple(h.shape) for h in heads] == [(2, 256, 13, 13), (2, 256, 26, 26), (2, 256, 52, 52)]
    assert hd.count['launches'] == 78          # 75 convolutions + 2 upsample-concats + the 208-wide block's shortcut as its own k_upsample_add (round 5)


def test_oracle_resize_known_answers():
    img = np.full((1, 6, 8, 3), 0, dtype=np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 20, 30                    # B, G, R
    out = Y.resize_frames(img, 4, 4)
    assert np.allclose(out[..., 0], Y.bf16_round(np.float32(30 / 255))) and np.allclose(out[..., 2], Y.bf16_round(np.float32(10 / 255)))
    assert not out[..., 3:].any()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 5, 7, 3), dtype=np.uint8)
    same = Y.resize_frames(img, 5, 7)                                      # identity size: exact pixels
    assert np.array_equal(same[..., :3], Y.bf16_round(img[..., ::-1].astype(np.float32) * np.float32(1 / 255)))
    up = Y.resize_frames(img, 10, 14)                                      # x2: output (0,1) = 0.75*p0 + 0.25*p1 along x
    exp = (0.75 * img[0, 0, 0, ::-1].astype(np.float32) + 0.25 * img[0, 0, 1, ::-1]) / 255
    assert np.allclose(up[0, 0, 1, :3], Y.bf16_round(exp.astype(np.float32)), atol=4e-3)


def test_oracle_decode_and_nms_known_answers():
    nc = 2
    heads = [np.full((g, g, 3 * (5 + nc)), -20.0, dtype=np.float32) for g in (2, 4, 8)]
    anchors = np.array(yolov3.ANCHORS, dtype=np.float32).reshape(3, 3, 2)[::-1].copy()
    # one box in head 1, cell (y=1, x=2), anchor 1: tx=ty=0 -> centre (2.5/4, 1.5/4); tw=th=0 -> anchor size / net size
    heads[1][1, 2, 7:14] = [0, 0, 0, 0, 20, -20, 20]                    # class 1 present, class 0 absent
    boxes, n = Y.detect(heads, anchors, 416, 416, nc, 1, 0.5, 0.45, 832, 416, 10)
    assert n == 1 and boxes.shape == (1, 5)
    aw, ah = anchors[1, 1]
    exp = [(0.625 - 0.5 * aw / 416) * 832, (0.375 - 0.5 * ah / 416) * 416, (0.625 + 0.5 * aw / 416) * 832, (0.375 + 0.5 * ah / 416) * 416]
    assert np.allclose(boxes[0, :4], exp, rtol=1e-5) and boxes[0, 4] > 0.999
    assert Y.detect(heads, anchors, 416, 416, nc, 0, 0.5, 0.45, 832, 416, 10)[0].shape == (0, 5)
    # NMS: b overlaps a with IoU 0.6 -> dropped at 0.45, kept at 0.7; c is disjoint; order is by score
    bx = np.array([[0, 0, 10, 10], [0, 0, 10, 6], [20, 20, 30, 30]], dtype=np.float32)
    sc = np.array([0.8, 0.9, 0.7], dtype=np.float32)
    assert Y.greedy_nms(bx, sc, 0.45, 10) == [1, 2]
    assert Y.greedy_nms(bx, sc, 0.7, 10) == [1, 0, 2]
    assert Y.greedy_nms(bx, sc, 0.7, 2) == [1, 0]
    assert Y.greedy_nms(bx, np.array([0.5, 0.5, 0.5], dtype=np.float32), 0.45, 10) == [0, 2]      # tie -> lower index first
