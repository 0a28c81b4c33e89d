"""GPU: the fused BasicBlock kernels (csrc/pam_block2.hip: k_bblock2_48, k_bblock2_96) against (a) a plain PyTorch fp32 reference of the
same block on the same bf16-rounded inputs / weights (intermediate rounded to bf16 as the kernels store it) and (b) the two-launch path
through pam_conv2d_nhwc_bf16, over the HRNet-W48 branch shapes, ragged tiles in both directions, tiny images and two rounds of items."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu

def make_block(c, seed):
    g = torch.Generator().manual_seed(seed)
    convs = []
    for _ in range(2):
        cv = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        with torch.no_grad():
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=g) * (2.0 / (9 * c)) ** 0.5)
            cv.bias.copy_(torch.randn(c, generator=g) * 0.2)
        convs.append(cv)
    return convs


def torch_block(x_bf16, c1, c2):
    """fp32 reference on the bf16-rounded operands; the intermediate is rounded to bf16 like the kernel's LDS copy."""
    x = x_bf16.float()
    w1, w2 = c1.weight.to(torch.bfloat16).float().to(x.device), c2.weight.to(torch.bfloat16).float().to(x.device)
    y = F.relu(F.conv2d(x, w1, c1.bias.to(x.device), 1, 1)).to(torch.bfloat16).float()
    return F.relu(F.conv2d(y, w2, c2.bias.to(x.device), 1, 1) + x)


@pytest.fixture(scope='module')
def eng():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = torch.device('cuda:0'); e.tile_cfg = -1
    return e


# ---- the resident-weights form (csrc/pam_block2.hip, C = 48) ---------------------------------------------------------------------
CASES2 = [
    # n, h, w, tile (None = the library's choice)
    (2, 96, 72, None),
    (20, 96, 72, None),          # the bench workload: 16 x 36 tiles, 240 items
    (20, 96, 72, (24, 24)),
    (3, 96, 72, (8, 18)),        # small tiles: the 3- / 2-M-tile instantiations
    (2, 50, 72, (16, 36)),       # ragged last tile row
    (2, 96, 70, (16, 36)),       # ragged last tile column
    (1, 64, 48, None),           # 256 x 192 crops
    (3, 7, 5, None),             # tiny image: one item, mostly junk slots
    (2, 33, 41, (11, 13)),       # nothing divides anything
    (28, 96, 72, None),          # a view-sharded Panoptic rank: two rounds of workgroups
]


@pytest.mark.parametrize('case', CASES2)
def test_block2_48_vs_torch_and_two_launches(eng, case):
    from pam import hrnet_hip
    n, h, w, tile = case
    c, dev = 48, eng.device
    c1, c2 = make_block(c, 200 + h)
    g = torch.Generator().manual_seed(11 + n + w)
    x = torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    op = hrnet_hip.PackedBlock(c1, c2, dev)
    y = eng.basic_block2(op, x, tile)
    torch.cuda.synchronize()
    ref = torch_block(x, c1, c2)
    err = (y.float() - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()
    assert bool((err <= tol).all()), (case, float(err.max()), float((err / tol).max()))
    assert float(err.mean()) < 2e-3, float(err.mean())
    # the two-launch path (k_conv3x3<48>) walks K in the same order: bit-identical
    p1, p2 = hrnet_hip.PackedConv(c1, dev), hrnet_hip.PackedConv(c2, dev)
    u = eng.conv(p2, eng.conv(p1, x, relu=True), res=x, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(y, u)


def test_block2_tile_choice_and_limits(eng):
    import ctypes as C
    t = (C.c_int32 * 2)()
    assert eng.lib.pam_basic_block2_tile(48, 20, 96, 72, t) == 0
    tr, tc = int(t[0]), int(t[1])
    assert 20 * -(-96 // tr) * -(-72 // tc) <= 256 and (tr + 4) * (tc + 4) <= 800       # one round of workgroups at 20 crops
    assert eng.lib.pam_basic_block2_tile(96, 20, 48, 36, t) == 0 and (int(t[0]) + 2) * (int(t[1]) + 4) <= 384      # 96 channels: weights streamed, 3 M tiles per wave
    t20 = (int(t[0]), int(t[1]))
    assert eng.lib.pam_basic_block2_tile(96, 4, 48, 36, t) == 0 and (int(t[0]), int(t[1])) == ((t20[0] + 1) // 2, t20[1])   # a few crops: items of half the size
    assert eng.lib.pam_basic_block2_tile(96, 9, 48, 36, t) == 0 and (int(t[0]), int(t[1])) == t20
    x96 = torch.zeros((1, 96, 48, 36), dtype=torch.bfloat16, device=eng.device).contiguous(memory_format=torch.channels_last)
    w96 = torch.zeros(1024 + 2 * 27 * 96 * 64, dtype=torch.uint8, device=eng.device)
    rc = eng.lib.pam_basic_block2_nhwc_bf16(C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream), C.c_void_p(x96.data_ptr()), C.c_void_p(w96.data_ptr()),
                                            C.c_void_p(torch.empty_like(x96).data_ptr()), 1, 48, 36, 96, 12, 36)
    assert rc != 0                                                                         # a 12 x 36 item needs the instantiation that was dropped
    assert eng.lib.pam_basic_block2_tile(192, 20, 24, 18, t) != 0
    x = torch.zeros((1, 48, 8, 8), dtype=torch.bfloat16, device=eng.device).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    w = torch.zeros(1024 + 2 * 14 * 48 * 64, dtype=torch.uint8, device=eng.device)
    st = C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    rc = eng.lib.pam_basic_block2_nhwc_bf16(st, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()), 1, 8, 8, 48, 40, 40)
    assert rc != 0                                                                         # tile beyond the LDS budget
    rc = eng.lib.pam_basic_block2_nhwc_bf16(st, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(x.data_ptr()), 1, 8, 8, 48, 0, 0)
    assert rc != 0                                                                         # in place


CASES96 = [
    # n, h, w, tile
    (2, 48, 36, None),
    (20, 48, 36, None),          # the bench workload: 6 x 36 tiles, 160 items
    (20, 48, 36, (4, 36)),
    (3, 48, 36, (6, 36)),
    (2, 27, 36, (6, 36)),        # ragged last tile row
    (2, 48, 34, (6, 18)),        # ragged last tile column
    (40, 48, 36, None),          # two frames per replay: 320 items, two rounds of workgroups
    (2, 32, 24, None),           # 256 x 192 crops
    (3, 7, 5, None),             # tiny image
    (2, 21, 29, (7, 11)),
    (28, 48, 36, None),
    # (rows + 4)(cols + 4) in (448, 640]: the two-k-steps-per-barrier form with its six-slot ring (KPB = 2, NPW = 10) -- wider maps at
    # other input resolutions reach it through the tile picker
    (2, 40, 60, (4, 60)),        # 8 x 64 = 512 slots
    (2, 6, 124, (1, 124)),       # 5 x 128 = 640 slots: the largest tile the kernel takes
]


@pytest.mark.parametrize('case', CASES96)
def test_block2_96_vs_torch_and_unfused(eng, case):
    """The streamed-weights fused block of the 96-channel branch (k_bblock2_96) against the fp32 PyTorch block (intermediate rounded to
    bf16 as the kernel stores it) and the two-launch path (same products; the residual enters the sum first, as in k_conv3x3s)."""
    from pam import hrnet_hip
    n, h, w, tile = case
    c, dev = 96, eng.device
    c1, c2 = make_block(c, 300 + h)
    g = torch.Generator().manual_seed(13 + n + w)
    x = torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    op = hrnet_hip.PackedBlock(c1, c2, dev)
    y = eng.basic_block2(op, x, tile)
    torch.cuda.synchronize()
    ref = torch_block(x, c1, c2)
    err = (y.float() - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()
    assert bool((err <= tol).all()), (case, float(err.max()), float((err / tol).max()))
    assert float(err.mean()) < 2e-3, float(err.mean())
    p1, p2 = hrnet_hip.PackedConv(c1, dev), hrnet_hip.PackedConv(c2, dev)
    u = eng.conv(p2, eng.conv(p1, x, relu=True), res=x, relu=True)
    torch.cuda.synchronize()
    d = (y.float() - u.float()).abs()
    assert float(d.max()) <= 4e-2 + 1e-2 * float(u.float().abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05
