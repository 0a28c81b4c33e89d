"""GPU: the fused stem (csrc/pam_stem.hip, k_stem_fused: conv1 8 -> 64 s2, conv2 64 -> 64 s2, layer1[0].conv1 64 -> 64 1x1 in one launch)
against (a) the three launches it replaces -- bit for bit -- and (b) a plain PyTorch fp32 reference of the same chain on the same
bf16-rounded inputs / weights (intermediates rounded to bf16 as the kernels store them), over the crop sizes of the two HRNet input
resolutions, ragged tiles in both directions, odd sizes and several rounds of work items."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu


def make_stem(seed):
    g = torch.Generator().manual_seed(seed)
    c1, c2, pw = nn.Conv2d(3, 64, 3, 2, 1, bias=True), nn.Conv2d(64, 64, 3, 2, 1, bias=True), nn.Conv2d(64, 64, 1, 1, 0, bias=True)
    with torch.no_grad():
        for cv in (c1, c2, pw):
            fan = cv.weight.shape[1] * cv.weight.shape[2] * cv.weight.shape[3]
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
            cv.bias.copy_(torch.randn(cv.weight.shape[0], generator=g) * 0.2)
    return c1, c2, pw


def torch_stem(x8, c1, c2, pw):
    dev = x8.device
    r = lambda t: t.to(torch.bfloat16).float()
    wb = lambda cv: (r(cv.weight).to(dev), cv.bias.to(dev))
    x = x8[:, :3].float()
    a = r(F.relu(F.conv2d(x, *wb(c1), 2, 1)))
    x0 = r(F.relu(F.conv2d(a, *wb(c2), 2, 1)))
    y1 = F.relu(F.conv2d(x0, *wb(pw)))
    return x0, y1


@pytest.fixture(scope='module')
def eng():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = torch.device('cuda:0'); e.tile_cfg = -1
    return e


CASES = [
    (1, 384, 288), (20, 384, 288),      # the bench workload: 54 items per crop, 1080 items on 256 workgroups
    (3, 256, 192),                      # the reference's other input resolution
    (2, 100, 60),                       # ragged tiles in both directions (x0 is 25 x 15)
    (2, 67, 45),                        # odd sizes: the last input row / column is a window's centre
    (1, 4, 4), (5, 9, 130),             # one pixel of x0; a single row of tiles
    (28, 384, 288),                     # a view-sharded Panoptic rank
]


@pytest.mark.parametrize('case', CASES)
def test_stem_fused_vs_three_launches_and_torch(eng, case):
    from pam import hrnet_hip
    n, h, w = case
    dev = eng.device
    c1, c2, pw = make_stem(100 + n + h)
    P1, P2, Pp = hrnet_hip.PackedConv(c1, dev, pad_cin_to=8), hrnet_hip.PackedConv(c2, dev), hrnet_hip.PackedPointwise64(pw, dev)
    op = hrnet_hip.PackedStem(P1, c2, Pp, dev)
    g = torch.Generator().manual_seed(7)
    x8 = torch.zeros((n, 8, h, w))
    x8[:, :3] = torch.randn((n, 3, h, w), generator=g)
    x8 = x8.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    x0, y1 = eng.stem_fused(op, x8)
    a = eng.conv(P1, x8, relu=True)
    b = eng.conv(P2, a, relu=True)
    c = eng.pointwise64(Pp, b)
    torch.cuda.synchronize()
    assert x0.shape == b.shape and y1.shape == c.shape
    assert torch.equal(x0, b), 'x0 differs from the two-launch stem: %d elements' % int((x0 != b).sum())
    assert torch.equal(y1, c), 'y1 differs from k_pw1: %d elements' % int((y1 != c).sum())
    rx0, ry1 = torch_stem(x8, c1, c2, pw)
    assert (x0.float() - rx0).abs().max().item() <= 2e-2 * max(1.0, rx0.abs().max().item())
    assert (y1.float() - ry1).abs().max().item() <= 2e-2 * max(1.0, ry1.abs().max().item())


def test_stem_fused_rejects_bad_arguments(eng):
    import ctypes as C
    lib = eng.lib
    z = C.c_void_p(0)
    assert lib.pam_stem_fused_nhwc_bf16(None, z, z, z, z, z, z, z, z, z, 1, 384, 288) != 0
    t = torch.zeros(16, device=eng.device)
    pp = C.c_void_p(t.data_ptr())
    assert lib.pam_stem_fused_nhwc_bf16(None, pp, pp, pp, pp, pp, pp, pp, pp, pp, 0, 384, 288) != 0
    assert lib.pam_stem_fused_nhwc_bf16(None, pp, pp, pp, pp, pp, pp, pp, pp, pp, 1, 2, 288) != 0
    assert lib.pam_stem_fused_nhwc_bf16(None, pp, pp, pp, pp, pp, pp, pp, pp, pp, 4096, 4096, 4096) != 0      # past the 2 GiB descriptor range
