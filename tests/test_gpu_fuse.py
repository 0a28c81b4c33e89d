"""GPU: the fuse-layer kernels of round 5 -- k_down48 (3x3 stride-2 convolution of a 48-channel input, csrc/pam_down.hip) and k_fuse_sum
(one output of an HR module's fuse layer with the coarser branches' 1x1 products computed inside the launch, csrc/pam_fuse.hip) --
against a plain PyTorch fp32 reference of the same op on the same bf16-rounded inputs AND against the launches they replace, which
they must equal bit for bit (same K order per output element, same summation order of the terms)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from pam import _lib, hrnet_hip
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = torch.device('cuda:0'); e.tile_cfg = -1
    return e


def _conv(cin, cout, k, stride, seed):
    g = torch.Generator().manual_seed(seed)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
    return conv


def _cl(shape, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)


D48_CASES = [
    # n, h, w, cout, wide (channels of the tensor the 48-channel input is a slice of), relu, relu_from, residual, tile (rows, cols, groups)
    (2, 96, 72, 96, 48, False, 0, False, None),          # stage 2: 48 -> 96
    (20, 96, 72, 192, 48, True, 96, False, None),        # the merged head of branch 0 in a stage-4 module at 20 crops
    (20, 96, 72, 144, 48, True, 96, False, None),        # ... in a stage-3 module
    (20, 48, 36, 192, 192, False, 0, False, None),       # second level: the input is a channel slice of the merged head's output
    (20, 48, 36, 48, 192, True, 0, False, None),
    (20, 24, 18, 384, 48, False, 0, False, None),        # third level: 12 x 9 outputs, the slabs spread over workgroups
    (3, 96, 72, 192, 48, True, 0, False, (4, 36, 1)),    # full-width tiles, 9 M tiles (one wave carries two)
    (3, 96, 72, 192, 48, True, 0, False, (10, 18, 2)),   # 12 M tiles, slabs in two groups
    (2, 96, 72, 96, 48, True, 48, True, None),           # residual + ReLU from the second slab on
    (2, 31, 23, 96, 48, False, 0, False, None),          # odd sizes: ragged last tile row and column, right / bottom padding
    (2, 31, 23, 96, 144, True, 0, False, (5, 4, 1)),     # small ragged tiles of a slice
    (3, 7, 5, 48, 48, True, 0, True, None),              # tiny image
    (2, 64, 48, 192, 48, True, 96, False, None),         # 256 x 192 crops
    (40, 96, 72, 192, 48, True, 96, False, None),        # two frames per replay
    (2, 48, 36, 384, 96, False, 0, False, (12, 6, 2)),   # 8 slabs in two groups of 4
]


@pytest.mark.parametrize('case', D48_CASES)
def test_down48_vs_torch_and_the_generic_kernel(eng, case):
    from pam import hrnet_hip
    n, h, w, cout, wide, relu, relu_from, use_res, tile = case
    dev = eng.device
    conv = _conv(48, cout, 3, 2, 100 + h + cout)
    op = hrnet_hip.PackedConv(conv, dev)
    xw = _cl((n, wide, h, w), 7 + n + w, dev)
    off = (wide - 48) // 2 // 8 * 8
    x = xw[:, off:off + 48]
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    res = _cl((n, cout, ho, wo), 9, dev) if use_res else None
    eng._keep = []
    eng.d48_tile = tile
    try:
        y = eng.conv_down48(op, x, res=res, relu=relu, relu_from=relu_from)
    finally:
        eng.d48_tile = None
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev), 2, 1)
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = torch.cat([ref[:, :relu_from], torch.relu(ref[:, relu_from:])], 1)
    torch.cuda.synchronize()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.float() - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2            # one bf16 rounding of the result + fp32 accumulation-order slack
    assert bool((err <= tol).all()), (case, err.max().item())
    if not use_res:                               # the generic kernels fold the residual into the sum BEFORE the products: last-bit differences there
        eng.down48 = False
        try:
            y0 = eng.conv(op, x, relu=relu, relu_from=relu_from)
        finally:
            eng.down48 = True
        torch.cuda.synchronize()
        assert torch.equal(y0, y), (case, (y0.float() - y.float()).abs().max().item())


def test_down48_tile_choice_and_limits(eng):
    import ctypes as C
    t = (C.c_int32 * 3)()
    for n, h, w, cout in ((20, 96, 72, 192), (20, 48, 36, 192), (20, 24, 18, 384), (1, 96, 72, 96), (217, 96, 72, 144)):
        assert eng.lib.pam_conv3x3s2_c48_tile(n, h, w, cout, t) == 0
        tr, tc, gs = t[0], t[1], t[2]
        assert (2 * tr + 1) * (2 * tc + 1) <= 789 and (tr * tc + 15) // 16 <= 24 and (cout // 48) % gs == 0, (n, h, w, cout, tr, tc, gs)
    assert eng.lib.pam_conv3x3s2_c48_tile(20, 96, 72, 100, t) != 0
    x = _cl((1, 48, 16, 12), 1, eng.device)
    y = torch.empty((1, 96, 8, 6), dtype=torch.bfloat16, device=eng.device)
    wz = torch.zeros(2 * 43008, dtype=torch.uint8, device=eng.device)
    args = lambda tr, tc, g: (None, C.c_void_p(x.data_ptr()), 48, C.c_void_p(wz.data_ptr()), None, None, 0, C.c_void_p(y.data_ptr()), 96, 1, 16, 12, 96, 0, 0, tr, tc, g)
    assert eng.lib.pam_conv3x3s2_c48_nhwc_bf16(*args(30, 30, 1)) == 0      # clamped to the map
    assert eng.lib.pam_conv3x3s2_c48_nhwc_bf16(*args(4, 4, 3)) != 0        # 2 slabs do not split into 3 groups
    torch.cuda.synchronize()


DS_CASES = [
    # n, h, w, cin, cout, wide, relu, relu_from
    (20, 48, 36, 96, 192, 96, False, 0),           # stage 3 / transition 2 at 20 crops: 6-row tiles, 64-channel slabs
    (20, 48, 36, 96, 288, 96, True, 192),          # the merged head of branch 1 in a stage-4 module: 48-channel slabs, ReLU from channel 192 on
    (20, 24, 18, 96, 384, 288, False, 0),          # its second level: the input is a channel slice, whole-image tiles
    (20, 24, 18, 192, 384, 192, False, 0),         # branch 2 -> 3 / transition 3: six chunks
    (3, 48, 36, 192, 384, 192, True, 0),
    (2, 31, 23, 96, 192, 96, True, 0),             # odd sizes: padding on all sides, ragged last tile
    (2, 32, 24, 96, 192, 96, False, 0),            # 256 x 192 crops
    (3, 7, 5, 192, 192, 192, True, 64),            # tiny image
    (40, 48, 36, 96, 288, 96, True, 192),
]


@pytest.mark.parametrize('case', DS_CASES)
def test_down_s_vs_torch(eng, case):
    """The streamed stride-2 kernel (k_down_s) against fp32 PyTorch; against the generic kernel it differs by the summation order only
    (K chunk by chunk of 32 input channels instead of tap by tap)."""
    from pam import hrnet_hip
    n, h, w, cin, cout, wide, relu, relu_from = case
    dev = eng.device
    conv = _conv(cin, cout, 3, 2, 200 + h + cout)
    op = hrnet_hip.PackedConv(conv, dev)
    xw = _cl((n, wide, h, w), 17 + n + w, dev)
    off = (wide - cin) // 2 // 8 * 8
    x = xw[:, off:off + cin]
    eng._keep = []
    assert eng.lib.pam_conv3x3s2_slab(h, w, cin, cout) in (48, 64)
    y = eng.conv_down_s(op, x, relu=relu, relu_from=relu_from)
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev), 2, 1)
    if relu:
        ref = torch.cat([ref[:, :relu_from], torch.relu(ref[:, relu_from:])], 1)
    eng.down_s = False
    try:
        y0 = eng.conv(op, x, relu=relu, relu_from=relu_from)
    finally:
        eng.down_s = True
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err = (y.float() - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2
    assert bool((err <= tol).all()), (case, err.max().item())
    assert float((y.float() - y0.float()).abs().max()) <= float(2.0 ** -6 * ref.abs().max() + 2e-2)


def test_down_s_declines_what_it_does_not_take(eng):
    assert eng.lib.pam_conv3x3s2_slab(48, 36, 48, 96) == 0 and eng.lib.pam_conv3x3s2_slab(48, 36, 96, 100) == 0
    assert eng.lib.pam_conv3x3s2_slab(400, 300, 96, 192) == 0                         # rows too wide for the 512-slot patch


FS_CASES = [
    # n, h, w, c, shifts of the coarser sources, number of plain terms, tile (a, b) or None
    (2, 96, 72, 48, (1, 2, 3), 0, None),           # output 0 of a stage-4 module
    (20, 96, 72, 48, (1, 2, 3), 0, None),
    (20, 48, 36, 96, (1, 2), 1, None),             # output 1: one strided-chain term + two coarser sources
    (20, 24, 18, 192, (1,), 2, None),              # output 2
    (20, 96, 72, 48, (1, 2), 0, None),             # stage 3
    (20, 48, 36, 96, (1,), 1, None),
    (20, 96, 72, 48, (1,), 0, None),               # stage 2
    (3, 96, 72, 48, (1, 2, 3), 0, (1, 3)),
    (3, 96, 72, 48, (1, 2, 3), 0, (1, 1)),
    (2, 64, 48, 48, (1, 2, 3), 0, None),           # 256 x 192 crops: an 8 x 6 coarsest map
    (2, 40, 24, 48, (1, 2, 3), 0, (2, 2)),         # a 5 x 3 coarsest map under 2 x 2 tiles: ragged tiles in both directions
    (2, 24, 16, 96, (2,), 2, None),                # a lone source two levels down
    (40, 96, 72, 48, (1, 2, 3), 0, None),
]


@pytest.mark.parametrize('case', FS_CASES)
def test_fuse_sum_vs_torch_and_the_launches_it_replaces(eng, case):
    from pam import hrnet_hip
    n, h, w, c, shifts, nplain, tile = case
    dev = eng.device
    convs = [_conv(c << sh, c, 1, 1, 40 + sh + c) for sh in shifts]
    op = hrnet_hip.PackedUp(convs, shifts, dev)
    base = _cl((n, c, h, w), 3, dev)
    wide = _cl((n, 2 * c + 16, h, w), 4, dev)
    plain = [wide[:, 8:8 + c], wide[:, 8 + c:8 + 2 * c]][:nplain]              # channel slices, as the merged strided heads hand them over
    srcs = [_cl((n, c << sh, h >> sh, w >> sh), 5 + sh, dev) for sh in shifts]
    eng._keep = []
    y = eng.fuse_sum(op, base, plain, srcs, relu=True, tile=tile or (0, 0))
    # the launches it replaces: one 1x1 convolution per source, then the up-sampling sum with the terms in branch order
    packed = [hrnet_hip.PackedConv(cv, dev) for cv in convs]
    terms = [eng.conv(pk, s) for pk, s in zip(packed, srcs)]
    y0 = eng.upsample_add(base, plain + terms, [0] * nplain + list(shifts), relu=True)
    ref = base.float()
    for t in plain:
        ref = ref + t.float()
    for cv, s, sh in zip(convs, srcs, shifts):
        t = F.conv2d(s.float(), cv.weight.detach().to(torch.bfloat16).float().to(dev), cv.bias.detach().to(dev))
        ref = ref + F.interpolate(t.to(torch.bfloat16).float(), scale_factor=2 ** sh, mode='nearest')
    ref = torch.relu(ref)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 2.0 ** -6 * ref.abs() + 6e-2).all()), (case, err.max().item())      # a product rounded one bf16 step apart moves the sum by that step
    assert torch.equal(y0, y), (case, (y0.float() - y.float()).abs().max().item())


def test_fuse_sum_argument_checks(eng):
    from pam import hrnet_hip
    dev = eng.device
    op = hrnet_hip.PackedUp([_conv(96, 48, 1, 1, 1)], [1], dev)
    base = _cl((1, 48, 10, 8), 1, dev)
    with pytest.raises(Exception):                 # a map that is not a whole number of source pixels
        eng.fuse_sum(op, _cl((1, 48, 9, 8), 1, dev), [], [_cl((1, 96, 4, 4), 2, dev)])
    y = eng.fuse_sum(op, base, [], [_cl((1, 96, 5, 4), 2, dev)], relu=False)
    torch.cuda.synchronize()
    assert y.shape == base.shape


def test_forward_with_the_round5_kernels_equals_the_forward_without_them():
    """The whole conv stack with k_down48 (default) and, optionally, k_fuse_sum against the same network on the generic strided kernel +
    separate 1x1 launches + k_upsample_add: bit-identical features; 203 launches with the fused sums, 221 without."""
    from pam import hrnet
    net = hrnet.HRNetPose(48, 17, None, use_graph=False)
    x = net.input_buffer(3)
    x.copy_(torch.randn(x.shape, device=net.device).to(x.dtype)); x[:, 3:] = 0
    hip = net.hip
    feats, launches = {}, {}
    try:
        for name, fs, d48, ds in (('old', False, False, False), ('d48', False, True, False), ('both', True, True, False), ('default', False, True, True)):
            hip.fused_sums, hip.down48, hip.down_s = fs, d48, ds
            hip.count = dict(bytes=0, flops=0, launches=0)
            feats[name] = hip.features(x).clone()
            launches[name] = hip.count['launches']
    finally:
        del hip.fused_sums, hip.down48, hip.down_s
        hip.count = None
    torch.cuda.synchronize()
    assert launches == dict(old=221, d48=221, both=203, default=221), launches
    assert torch.equal(feats['old'], feats['d48']) and torch.equal(feats['old'], feats['both'])
    # k_down_s walks K in another order: the features move in the last bf16 bits only
    rel = float((feats['default'].float() - feats['old'].float()).norm() / feats['old'].float().norm())
    # two bf16 evaluations of the same network in different summation orders are as far from each other as each is from fp32 (6e-3:
    # test_gpu_image.py::test_bf16_stack_vs_fp32, which runs this default path): last-bit differences of twelve layers, amplified downstream
    assert 0 < rel < 2e-2, rel
