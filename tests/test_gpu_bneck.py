"""GPU: the fused layer1 Bottleneck (csrc/pam_bneck.hip, k_bneck: 3x3 64 -> 64, conv3 64 -> 256 + residual, the next block's conv1
256 -> 64 in one launch) against (a) the two launches it replaces -- bit for bit -- and (b) a plain PyTorch fp32 reference of the same
chain on the same bf16-rounded operands, over the layer1 shapes of both input resolutions, ragged tiles, tiny maps and several rounds."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu


def make_convs(seed):
    g = torch.Generator().manual_seed(seed)
    c2, c3, c1n, down = nn.Conv2d(64, 64, 3, 1, 1, bias=True), nn.Conv2d(64, 256, 1, bias=True), nn.Conv2d(256, 64, 1, bias=True), nn.Conv2d(64, 256, 1, bias=True)
    with torch.no_grad():
        for cv in (c2, c3, c1n, down):
            fan = cv.weight.shape[1] * cv.weight.shape[2] * cv.weight.shape[3]
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
            cv.bias.copy_(torch.randn(cv.weight.shape[0], generator=g) * 0.2)
    return c2, c3, c1n, down


@pytest.fixture(scope='module')
def eng():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = torch.device('cuda:0'); e.tile_cfg = -1; e.c96_slab = 48
    return e


CASES = [(2, 96, 72, True, False), (20, 96, 72, True, False), (20, 96, 72, False, False), (3, 64, 48, True, False), (2, 50, 37, True, False),
         (2, 7, 5, False, False), (1, 2, 3, True, False), (28, 96, 72, True, False),
         # the first block: no residual, the downsample convolution over the block input as a second K source
         (2, 96, 72, True, True), (20, 96, 72, True, True), (3, 64, 48, False, True), (2, 50, 37, True, True), (1, 2, 3, True, True)]


@pytest.mark.parametrize('case', CASES)
def test_bneck_fused_vs_two_launches_and_torch(eng, case):
    from pam import hrnet_hip
    n, h, w, has2, first = case
    dev = eng.device
    c2, c3, c1n, down = make_convs(10 + n + h)
    tail = hrnet_hip.PackedTail(c3, down if first else None, c1n if has2 else None, dev)
    op = hrnet_hip.PackedBneck(c2, tail, dev)
    P2 = hrnet_hip.PackedConv(c2, dev)
    g = torch.Generator().manual_seed(3)
    cl = lambda t: t.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    y1 = cl(torch.relu(torch.randn((n, 64, h, w), generator=g)))
    res = cl(torch.relu(torch.randn((n, 256, h, w), generator=g))) if not first else None
    x0 = cl(torch.relu(torch.randn((n, 64, h, w), generator=g))) if first else None
    X, Y = eng.bottleneck_fused(op, y1, res, x0)
    y2 = eng.conv(P2, y1, relu=True)
    X2, Y2 = eng.bottleneck_tail(tail, y2, x0, res, 0)
    torch.cuda.synchronize()
    assert torch.equal(X, X2), 'X differs from the two-launch path: %d elements' % int((X != X2).sum())
    assert (Y is None) == (not has2) and (Y is None or torch.equal(Y, Y2))
    # fp32 reference on the bf16-rounded operands, intermediates rounded where the kernels round them
    r = lambda t: t.to(torch.bfloat16).float()
    wb = lambda cv: (r(cv.weight).to(dev), cv.bias.to(dev))
    t2 = r(F.relu(F.conv2d(y1.float(), *wb(c2), 1, 1)))
    tx = r(F.relu(F.conv2d(t2, *wb(c3)) + (res.float() if not first else F.conv2d(x0.float(), *wb(down)))))
    assert (X.float() - tx).abs().max().item() <= 2e-2 * max(1.0, tx.abs().max().item())
    if has2:
        ty = F.relu(F.conv2d(tx, *wb(c1n)))
        assert (Y.float() - ty).abs().max().item() <= 2e-2 * max(1.0, ty.abs().max().item())


def test_bneck_fused_rejects_bad_arguments(eng):
    import ctypes as C
    lib = eng.lib
    t = torch.zeros(16, device=eng.device)
    pp = C.c_void_p(t.data_ptr())
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, None, None, pp, pp, pp, pp, pp, pp, pp, pp, pp, 1, 96, 72) != 0
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, pp, pp, pp, pp, pp, pp, pp, pp, pp, pp, pp, 1, 96, 72) != 0        # both a residual and x0
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, pp, None, None, pp, pp, pp, pp, pp, pp, pp, pp, 1, 96, 72) != 0    # neither
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, pp, None, pp, pp, pp, pp, pp, pp, pp, pp, None, 1, 96, 72) != 0    # conv1 image without its output
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, pp, None, pp, pp, pp, pp, pp, None, None, pp, None, 0, 96, 72) != 0
    assert lib.pam_bottleneck_fused_nhwc_bf16(None, pp, None, pp, pp, pp, pp, pp, None, None, pp, None, 4096, 1024, 1024) != 0    # past the 2 GiB descriptor range
