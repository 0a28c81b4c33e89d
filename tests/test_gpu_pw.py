"""GPU: the fused pointwise tail of a layer1 Bottleneck (csrc/pam_pw.hip, pam_bottleneck_tail_nhwc_bf16) against a plain PyTorch fp32
reference of the same ops on the same bf16-rounded operands: X = ReLU(conv3(y2) [+ downsample(x0)] [+ residual]), then
y1 = ReLU(conv1_next(bf16(X))); every (downsample / residual / second product / wave-tile) combination, ragged pixel counts."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.ConvEngine()
    e.lib = _lib.load()
    return e, hrnet_hip


def _bf(t):
    return t.to(torch.bfloat16).float()


CASES = [
    # n, h, w, first block (downsample source), residual, second product, tile_cfg
    (2, 16, 12, False, True, True, 0),
    (2, 16, 12, True, False, True, 0),       # block 0: conv3 + downsample as two K chunks, no residual tensor
    (2, 16, 12, False, True, False, 0),      # last block: no next conv1
    (1, 5, 3, False, True, True, 0),         # 15 pixels: less than one 16-pixel tile
    (3, 7, 11, False, True, True, 1),        # 231 pixels, ragged against every wave-tile size
    (3, 7, 11, False, True, True, 2),
    (3, 7, 11, False, True, True, 3),
    (3, 7, 11, True, False, True, 3),
    (1, 9, 7, False, False, True, 0),        # neither residual nor downsample
    (1, 9, 7, False, False, False, 2),
    (20, 96, 72, False, True, True, 0),      # the real layer at 20 crops: every persistent workgroup walks several tiles
    (20, 96, 72, True, False, True, 0),
    (7, 96, 72, False, True, False, 3),
]


@pytest.mark.parametrize('case', CASES)
def test_bottleneck_tail_vs_torch(eng, case):
    e, hh = eng
    n, h, w, first, use_res, second, cfg = case
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w + cfg)
    rnd = lambda *s: torch.randn(*s, generator=g)
    conv3 = nn.Conv2d(64, 256, 1, bias=True); down = nn.Conv2d(64, 256, 1, bias=True) if first else None
    conv1 = nn.Conv2d(256, 64, 1, bias=True) if second else None
    with torch.no_grad():
        for cv, sc in ((conv3, 0.15), (down, 0.15), (conv1, 0.08)):
            if cv is not None:
                cv.weight.copy_(rnd(*cv.weight.shape) * sc); cv.bias.copy_(rnd(*cv.bias.shape) * 0.5)
    op = hh.PackedTail(conv3, down, conv1, dev)
    cl = lambda t: t.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    y2, x0, res = rnd(n, 64, h, w), (rnd(n, 64, h, w) if first else None), (rnd(n, 256, h, w) * 2 if use_res else None)
    X, Y = e.bottleneck_tail(op, cl(y2), cl(x0) if first else None, cl(res) if use_res else None, cfg)
    torch.cuda.synchronize()
    with torch.no_grad():
        acc = F.conv2d(_bf(y2), _bf(conv3.weight), conv3.bias)
        if first:
            acc = acc + F.conv2d(_bf(x0), _bf(down.weight), down.bias)
        if use_res:
            acc = acc + _bf(res)
        xr = torch.relu(acc)
        err = (X.float().cpu() - xr).abs()
        assert float(err.max()) <= 2.0 ** -7 * float(xr.abs().max()) + 1e-2, float(err.max())     # one bf16 rounding of an fp32 sum
        assert float(err.mean()) < 4e-3 * max(1.0, float(xr.abs().mean()))
        if second:
            # the second product reads the ROUNDED X the kernel itself produced (bit-exact input), fp32 accumulate
            yr = torch.relu(F.conv2d(X.float().cpu(), _bf(conv1.weight), conv1.bias))
            e2 = (Y.float().cpu() - yr).abs()
            assert float(e2.max()) <= 2.0 ** -7 * float(yr.abs().max()) + 1e-2, float(e2.max())
        else:
            assert Y is None


def test_layer1_fused_tail_matches_unfused_path():
    """The whole stem + layer1 through the executor, fused tails vs one launch per convolution (both on the same packed weights):
    equal up to the bf16 rounding of the first block's downsample branch, which the fused form keeps in fp32."""
    from pam import hrnet
    net = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False)
    hip = net.hip
    x = net.input_buffer(3)
    x.copy_(torch.randn(x.shape, device=x.device).to(x.dtype)); x[:, 3:] = 0
    hip.stop_after = 'layer1'
    try:
        hip.fuse_tail = True
        a = hip.features(x).float()
        hip.fuse_tail = False
        b = hip.features(x).float()
    finally:
        hip.stop_after = None; hip.fuse_tail = True
    torch.cuda.synchronize()
    assert a.shape == b.shape == (3, 256, 96, 72)
    rel = float((a - b).norm() / b.norm())
    assert rel < 6e-3, rel


@pytest.mark.parametrize('shape', [(1, 5, 3), (3, 7, 11), (2, 96, 72), (20, 96, 72)])
def test_pointwise64_vs_torch(eng, shape):
    e, hh = eng
    n, h, w = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n * 100 + h)
    conv = nn.Conv2d(64, 64, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2); conv.bias.copy_(torch.randn(64, generator=g))
    op = hh.PackedPointwise64(conv, dev)
    x = torch.randn((n, 64, h, w), generator=g)
    y = e.pointwise64(op, x.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(_bf(x), _bf(conv.weight), conv.bias)).detach()
    err = (y.float().cpu() - ref).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-2, float(err.max())
