"""GPU: the fused BasicBlock kernel (csrc/pam_block.hip) against (a) a plain PyTorch fp32 reference of the same block on the same
bf16-rounded inputs / weights (intermediate rounded to bf16 as the kernel stores it) and (b) the two-launch path through
pam_conv2d_nhwc_bf16, over the HRNet-W48 branch shapes, ragged tiles, tiny images and grouped launches."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu

CASES = [
    # n, c, h, w
    (2, 48, 96, 72),
    (3, 96, 48, 36),
    (3, 192, 24, 18),
    (20, 48, 96, 72),        # full 20-crop shapes of the bench workload
    (20, 96, 48, 36),
    (20, 192, 24, 18),
    (2, 48, 50, 72),         # ragged last tile (50 = 6 * 8 + 2)
    (2, 96, 27, 36),
    (2, 192, 13, 18),
    (1, 48, 64, 48),         # 256x192 crops: 64 x 48 heat-maps
    (2, 96, 32, 24),
    (2, 192, 16, 12),
    (3, 48, 7, 5),           # tiny images: one item, mostly junk tiles
    (1, 192, 3, 4),
]


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


@pytest.mark.parametrize('waves', [4, 8])
@pytest.mark.parametrize('case', CASES)
def test_block_vs_torch_and_unfused(eng, case, waves):
    from pam import hrnet_hip
    n, c, h, w = case
    dev = eng.device
    if eng.lib.pam_basic_block_rows(c, h, w, waves) <= 0:
        pytest.skip('shape not supported with %d waves' % waves)
    c1, c2 = make_block(c, 100 + c + h)
    g = torch.Generator().manual_seed(7 + n + h)
    x = torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    op = hrnet_hip.PackedBlock(c1, c2, dev)
    y = eng.basic_blocks([op], [x], waves)[0]
    torch.cuda.synchronize()
    ref = torch_block(x, c1, c2)
    err = (y.float() - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()                      # one bf16 rounding of the output + the intermediate's rounding boundary cases
    assert bool((err <= tol).all()), (case, float(err.max()), float((err / tol).max()))
    assert float(err.mean()) < 2e-3, float(err.mean())
    # two-launch path: same MFMA products; the accumulation order over K may differ (chunking), so equal to within rounding
    p1, p2 = hrnet_hip.PackedConv(c1, dev), hrnet_hip.PackedConv(c2, dev)
    t = eng.conv(p1, x, relu=True)
    u = eng.conv(p2, t, res=x, relu=True)
    torch.cuda.synchronize()
    d = (y.float() - u.float()).abs()
    assert float(d.max()) <= 4e-2 + 1e-2 * float(u.float().abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05       # nearly every element is bit-equal
    if c == 48:                                        # same K order as k_conv3x3<48>: bit-exact
        assert torch.equal(y, u)


def test_grouped_launch_equals_single(eng):
    from pam import hrnet_hip
    dev = eng.device
    shapes = [(5, 48, 96, 72), (5, 96, 48, 36), (5, 192, 24, 18)]
    ops, xs = [], []
    for k, (n, c, h, w) in enumerate(shapes):
        c1, c2 = make_block(c, 31 + k)
        ops.append(hrnet_hip.PackedBlock(c1, c2, dev))
        xs.append(torch.randn((n, c, h, w), generator=torch.Generator().manual_seed(k)).to(torch.bfloat16).to(dev)
                  .contiguous(memory_format=torch.channels_last))
    single = [eng.basic_blocks([o], [x], 8)[0] for o, x in zip(ops, xs)]
    grouped = eng.basic_blocks(ops, xs)
    rev = eng.basic_blocks(ops[::-1], xs[::-1])[::-1]
    two4 = eng.basic_blocks(ops[:2], xs[:2], 4)                          # two workgroups per CU: same values (same K order per element)
    torch.cuda.synchronize()
    for a, b, c_ in zip(single, grouped, rev):
        assert torch.equal(a, b) and torch.equal(a, c_)
    assert torch.equal(two4[0], single[0]) and torch.equal(two4[1], single[1])


@pytest.mark.parametrize('shape', [(20, 96, 72), (3, 50, 72), (2, 7, 5)])
def test_short_items_equal_normal_items(eng, shape):
    """The "short" 4-row items (bits 4-7 of `waves`: a packing choice for grouped launches) compute exactly what the normal items do:
    same K order per output element -> bit-equal, for either branch of a grouped 48 + 96 launch and for single launches."""
    from pam import hrnet_hip
    dev = eng.device
    n, h, w = shape
    ops, xs = [], []
    for k, (c, hh, ww) in enumerate([(48, h, w), (96, max(1, h // 2), max(1, w // 2))]):
        c1, c2 = make_block(c, 51 + k)
        ops.append(hrnet_hip.PackedBlock(c1, c2, dev))
        xs.append(torch.randn((n, c, hh, ww), generator=torch.Generator().manual_seed(3 + k)).to(torch.bfloat16).to(dev)
                  .contiguous(memory_format=torch.channels_last))
    ref = eng.basic_blocks(ops, xs, 8)
    for mask in (1, 2, 3):
        got = eng.basic_blocks(ops, xs, 8 | (mask << 4))
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), mask
    one = eng.basic_blocks([ops[1]], [xs[1]], 8 | (1 << 4))[0]
    torch.cuda.synchronize()
    assert torch.equal(one, ref[1])


def test_block_rejects_unsupported(eng):
    assert eng.lib.pam_basic_block_rows(384, 12, 9, 0) == 0
    assert eng.lib.pam_basic_block_rows(48, 96, 200, 0) == 0
    assert eng.lib.pam_basic_block_rows(64, 96, 72, 0) == 0
    assert eng.lib.pam_basic_block_rows(192, 24, 18, 4) == 0          # the 192-wide tile does not fit two workgroups per CU


def test_grouped_fuse_launches_equal_single(eng):
    """pam_conv2d_group_nhwc_bf16 / pam_upsample_add_group_nhwc_bf16 against one launch per member (same arithmetic, same K order)."""
    from pam import hrnet_hip
    dev = eng.device
    g = torch.Generator().manual_seed(5)
    mk = lambda n, c, h, w: torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    x0, x1, x2 = mk(3, 48, 24, 18), mk(3, 96, 12, 9), mk(3, 192, 6, 5)
    convs = [nn.Conv2d(48, 144, 3, 2, 1), nn.Conv2d(96, 48, 1, 1, 0), nn.Conv2d(192, 144, 1, 1, 0), nn.Conv2d(96, 192, 3, 2, 1)]
    ops = [hrnet_hip.PackedConv(c, dev) for c in convs]
    specs = [(ops[0], x0, True, 96), (ops[1], x1, False, 0), (ops[2], x2, False, 0), (ops[3], x1, True, 0)]
    ys = eng.conv_group(specs)
    torch.cuda.synchronize()
    for (op, x, relu, rf), y in zip(specs, ys):
        ref = eng.conv(op, x, relu=relu, relu_from=rf)
        torch.cuda.synchronize()
        assert torch.equal(y, ref), (op.cin, op.cout)
    # a channel slice as input (the chains continue from slices of a merged convolution's output)
    sl = ys[0][:, 96:144]
    op5 = hrnet_hip.PackedConv(nn.Conv2d(48, 192, 3, 2, 1), dev)
    y5 = eng.conv_group([(op5, sl, False, 0)])[0]
    ref5 = eng.conv(op5, sl)
    torch.cuda.synchronize()
    assert torch.equal(y5, ref5)
    # sums
    t1 = mk(3, 48, 12, 9); t2 = mk(3, 96, 6, 5)[:, :48]
    b1 = mk(3, 96, 12, 9); u1 = mk(3, 96, 6, 5)
    sums = [(x0, [t1, t2], [1, 2], True), (b1, [u1], [1], True), (x2, [], [], True)]
    outs = eng.upsample_add_group(sums)
    torch.cuda.synchronize()
    for (base, terms, sh, relu), o in zip(sums, outs):
        ref = eng.upsample_add(base, terms, sh, relu) if terms else torch.relu(base)
        torch.cuda.synchronize()
        assert torch.equal(o, ref)


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
def test_block2_vs_torch_and_ring_kernel(eng, case):
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
    # the ring kernel and the two-launch path walk K in the same order: bit-identical
    if eng.lib.pam_basic_block_rows(c, h, w, 8) > 0:
        assert torch.equal(y, eng.basic_blocks([op], [x], 8)[0])
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
    assert eng.lib.pam_basic_block2_tile(96, 20, 48, 36, t) == 0 and (int(t[0]) + 4) * (int(t[1]) + 4) <= 640      # 96 channels: weights streamed
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
    (20, 48, 36, None),          # the bench workload: 12 x 36 tiles, 80 items
    (20, 48, 36, (12, 18)),
    (3, 48, 36, (6, 36)),
    (2, 27, 36, (12, 36)),       # ragged last tile row
    (2, 48, 34, (12, 18)),       # ragged last tile column
    (2, 32, 24, None),           # 256 x 192 crops
    (3, 7, 5, None),             # tiny image
    (2, 21, 29, (7, 11)),
    (28, 48, 36, None),
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
