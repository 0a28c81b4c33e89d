"""GPU: the hand-written MFMA convolution (csrc/pam_conv.hip) against a plain PyTorch fp32 reference of the same op on the
same bf16-rounded inputs, over every (Cin, Cout, kernel, stride, tile) family HRNet-W48 uses, incl. ragged M / K tails."""
import ctypes as C

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pam

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, k, stride, residual, relu
    (2, 24, 18, 48, 48, 3, 1, True, True),
    (1, 96, 72, 48, 48, 3, 1, False, True),
    (3, 12, 9, 96, 96, 3, 1, True, True),
    (2, 12, 10, 192, 192, 3, 1, True, False),
    (1, 12, 9, 384, 384, 3, 1, True, True),
    (2, 24, 18, 64, 64, 3, 1, True, True),        # layer1 bottleneck 3x3 (64-wide N tiles)
    (20, 96, 72, 48, 48, 3, 1, True, True),       # full-size branch-0 layer
    (2, 25, 18, 48, 48, 3, 1, True, True),        # ragged last row tile (25 = 12 + 12 + 1)
    (180, 25, 18, 48, 48, 3, 1, True, True),      # 540 tiles on 512 persistent workgroups, ragged tiles among the second ones
    (50, 96, 72, 48, 48, 3, 1, True, True),       # 1600 tiles: every persistent workgroup walks 3-4 tiles
    (1, 7, 5, 96, 96, 3, 1, True, True),          # tiny image: one 2-wave workgroup per slab
    (3, 31, 72, 48, 96, 3, 1, False, True),       # ragged, wide rows, Cout != Cin
    (2, 9, 12, 384, 192, 3, 1, True, False),      # 64-channel chunks, narrow slabs
    (2, 13, 11, 8, 64, 3, 2, False, True),       # stem conv1 (3 -> 8 padded channels), odd sizes
    (2, 20, 16, 64, 64, 3, 2, False, True),
    (2, 16, 12, 64, 64, 1, 1, False, True),
    (2, 16, 12, 64, 256, 1, 1, True, True),
    (2, 16, 12, 256, 64, 1, 1, False, True),
    (2, 16, 12, 256, 48, 3, 1, False, True),
    (2, 16, 12, 256, 96, 3, 2, False, True),
    (2, 12, 8, 384, 48, 1, 1, False, False),     # fuse 1x1
    (2, 16, 12, 48, 96, 3, 2, True, False),      # fuse stride-2 with accumulate
    (1, 7, 5, 96, 192, 3, 2, False, True),
    (2, 20, 16, 48, 144, 3, 2, False, True),     # merged fuse-layer heads: 3 / 4 / 6 N tiles in one workgroup (tiles 5, 6, 7)
    (2, 20, 16, 48, 192, 3, 2, False, True),
    (2, 13, 11, 96, 288, 3, 2, False, True),
    (20, 48, 36, 96, 288, 3, 2, False, True),    # the real merged heads of branch 1 at 20 crops (96-channel slabs: tile 9)
    (3, 24, 18, 192, 384, 3, 2, False, False),
    (2, 12, 9, 384, 96, 1, 1, False, False),
    # streamed kernel k_conv3x3s (Cin 192 / 384): its three tile shapes, full-size layers, ragged last tile, Cout != Cin
    (20, 24, 18, 192, 192, 3, 1, True, True),    # 4 M tiles per wave, 320-slot patch
    (20, 12, 9, 384, 384, 3, 1, True, True),     # 3 M tiles, 192-slot patch
    (3, 16, 12, 192, 192, 3, 1, False, True),
    (2, 40, 6, 192, 64, 3, 1, True, False),      # 5 M tiles, 384-slot patch
    (2, 25, 18, 384, 128, 3, 1, True, True),     # ragged last row tile
    (1, 8, 6, 384, 384, 3, 1, False, False),
    (5, 3, 3, 192, 192, 3, 1, True, True),       # image smaller than one M tile
]


@pytest.fixture(scope='module')
def eng():
    from pam import hrnet, hrnet_hip
    m = hrnet.PoseHighResolutionNet()
    return hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet), hrnet_hip


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_conv_vs_torch(eng, case, tile):
    from pam import _lib, hrnet_hip
    n, h, w, cin, cout, k, stride, use_res, relu = case
    if tile in (10, 11, 12) and (cout % 48 or use_res or cin == 8):
        pytest.skip('128- / 64-pixel-tile streamed implicit GEMM: 48-channel slabs, no residual')
    if tile == 9 and (cout % 96 or use_res or cin == 8 or (k == 3 and stride == 1)):
        pytest.skip('96-channel-slab streamed implicit GEMM: fuse-layer shapes (Cout % 96 == 0, no residual)')
    nb = cout // (48 if cout % 48 == 0 else 64)
    if (tile in (1, 3) and nb % 2) or (tile in (5, 7) and nb % 3) or (tile == 6 and nb % 4):
        pytest.skip('tile needs a matching number of N tiles')
    if 5 <= tile <= 7 and (k != 3 or stride != 2 or cout % 48):
        pytest.skip('wide-N tiles are exercised on the strided fuse-layer shapes')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(hash(case) % 1000)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((n, cin, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    res = torch.randn((n, cout, ho, wo), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last) if use_res else None
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = dev; e.tile_cfg = tile
    e.down48 = tile == -1                        # a stated tile keeps the strided 48-channel layers on the generic kernels (k_down48: tests/test_gpu_fuse.py)
    y = e.conv(op, x, res=res, relu=relu)
    wq = conv.weight.detach().to(torch.bfloat16).float().to(dev)
    ref = F.conv2d(x.float(), wq, conv.bias.detach().to(dev), stride, k // 2)
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = torch.relu(ref)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err = (y.float() - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2          # one bf16 rounding of the result + fp32 accumulation-order slack
    assert bool((err <= tol).all()), (case, tile, err.max().item())


@pytest.mark.parametrize('shape', [(20, 24, 18, 192, 192), (20, 12, 9, 384, 384), (2, 25, 18, 384, 128), (2, 40, 6, 192, 64),
                                   (4, 96, 72, 64, 64), (3, 96, 72, 256, 48), (2, 50, 30, 64, 64)])
def test_streamed_kernel_is_the_default_and_matches_the_classic_one(eng, shape):
    """pam_conv3x3_layout() announces the streamed kernel for these layers; tile_cfg = -2 forces k_conv3x3 on the classic weight image.
    Same products, different summation order (the streamed kernel starts from bias + residual): equal up to a bf16 rounding here and there."""
    from pam import _lib, hrnet_hip
    n, h, w, cin, cout = shape
    lib = _lib.load()
    bn = 48 if cout == 48 else 64
    assert lib.pam_conv3x3_layout(h, w, cin, cout) == bn
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * 9)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((n, cin, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((n, cout, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = lib; e.device = dev
    e.tile_cfg = -1; y = e.conv(op, x, res=res, relu=True)
    e.tile_cfg = -2; u = e.conv(op, x, res=res, relu=True)
    torch.cuda.synchronize()
    assert op._images[(bn, True)].shape == (cout // bn, cin // 32, 9, bn, 4, 8)
    d = (y.float() - u.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(u.float().abs().max()) + 1e-6
    assert float((d > 0).float().mean()) < 0.02


def test_upsample_add_vs_torch():
    from pam import _lib, hrnet_hip
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1
    base = torch.randn((2, 48, 24, 16), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    terms = [torch.randn((2, 48, 24 >> s, 16 >> s), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last) for s in (1, 2, 3)]
    for nt in (1, 2, 3):
        for relu in (False, True):
            y = e.upsample_add(base, terms[:nt], [1, 2, 3][:nt], relu)
            ref = base.float()
            for s, t in zip((1, 2, 3), terms[:nt]):
                ref = ref + F.interpolate(t.float(), scale_factor=2 ** s, mode='nearest')
            if relu:
                ref = torch.relu(ref)
            torch.cuda.synchronize()
            assert torch.equal(y, ref.to(torch.bfloat16))


def test_channel_sliced_operands_and_partial_relu():
    """pam_conv2d_nhwc_bf16_ex / pam_upsample_add_nhwc_bf16_ex: input = channel slice of a wider tensor, ReLU only from a channel on,
    sum terms = channel slices (the merged fuse-layer convolutions of HipHRNet)."""
    from pam import _lib, hrnet_hip
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(9)
    e = hrnet_hip.ConvEngine(); e.lib = _lib.load(); e.device = dev
    wide = torch.randn((2, 192, 24, 18), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    for (off, cin, cout, k, stride, relu_from) in [(96, 48, 192, 3, 2, 0), (144, 48, 48, 3, 2, 0), (0, 96, 144, 3, 2, 96), (48, 96, 96, 1, 1, 48)]:
        conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5); conv.bias.copy_(torch.randn(cout, generator=g))
        op = hrnet_hip.PackedConv(conv, dev)
        x = wide[:, off:off + cin]
        y = e.conv(op, x, relu=True, relu_from=relu_from)
        ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev), stride, k // 2)
        ref = torch.cat([ref[:, :relu_from], torch.relu(ref[:, relu_from:])], 1)
        torch.cuda.synchronize()
        err = (y.float() - ref).abs()
        assert bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all()), ((off, cin, cout, k, stride, relu_from), err.max().item())
    base = torch.randn((2, 48, 24, 18), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    coarse = torch.randn((2, 96, 12, 9), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    out = e.upsample_add(base, [wide[:, 48:96], coarse[:, 48:96]], [0, 1], True)
    ref = torch.relu(base.float() + wide[:, 48:96].float() + F.interpolate(coarse[:, 48:96].float(), scale_factor=2, mode='nearest'))
    torch.cuda.synchronize()
    assert torch.equal(out, ref.to(torch.bfloat16))


@pytest.mark.gpu
def test_leaky_layer_of_a_streamed_shape_takes_the_classic_kernel():
    """A 3x3 layer whose shape the streamed kernel would take (Cin 384, e.g. YOLOv3-tiny's 384 -> 256 after its route) but with a leaky
    activation: the host asks for the classic kernel and image (tile_cfg -2), the result matches torch."""
    from pam import _lib, hrnet_hip
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    conv = nn.Conv2d(384, 256, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (384 * 9)) ** 0.5)
        conv.bias.copy_(torch.randn(256, generator=g))
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((2, 384, 13, 13), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1
    assert e.lib.pam_conv3x3_layout(13, 13, 384, 256) == 64
    y = e.conv(op, x, relu='leaky')
    # not the streamed 3x3 kernel (2), whose Cin = 384 instantiations take codes 0 / 1 only: a classic kernel or -- round 5 -- the streamed
    # implicit GEMM, whose epilogue has the leaky activation since then (k_conv3x3<384> has no leaky epilogue)
    assert e.lib.pam_conv_last_kernel() in (0, 1, 3)
    wq = conv.weight.detach().to(torch.bfloat16).float().to(dev)
    ref = F.leaky_relu(F.conv2d(x.float(), wq, conv.bias.detach().to(dev), 1, 1), 0.1)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all()), float(err.max())
    y2 = e.conv(op, x, relu=True)
    assert e.lib.pam_conv_last_kernel() == 2                           # k_conv3x3s


@pytest.mark.parametrize('slab', [48])
@pytest.mark.parametrize('shape', [(20, 48, 36), (2, 25, 18), (3, 7, 5), (2, 50, 30)])
def test_streamed_kernel_for_the_96_channel_branch(eng, shape, slab):
    """96 -> 96 3x3 layers on k_conv3x3s with 48-channel slabs (the executor's c96_slab) vs torch fp32."""
    from pam import _lib, hrnet_hip
    n, h, w = shape
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    conv = nn.Conv2d(96, 96, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (96 * 9)) ** 0.5)
        conv.bias.copy_(torch.randn(96, generator=g))
    x = torch.randn((n, 96, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((n, 96, h, w), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    e.lib = lib; e.device = dev; e.tile_cfg = -1
    assert lib.pam_conv3x3_layout(h, w, 96, 96) == 0 and lib.pam_conv3x3_layout_ex(h, w, 96, 96, slab) == slab
    op = hrnet_hip.PackedConv(conv, dev)
    e.c96_slab = 0
    y0 = e.conv(op, x, res=res, relu=True)
    assert lib.pam_conv_last_kernel() == 1                # c96_slab = 0: k_conv3x3
    e.c96_slab = slab
    y = e.conv(op, x, res=res, relu=True)
    assert lib.pam_conv_last_kernel() == 2                # PAM_CONV_KERNEL_3X3S
    ref = torch.relu(F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float().to(dev), conv.bias.detach().to(dev), 1, 1) + res.float())
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all()), float(err.max())


@pytest.mark.parametrize('shape', [(3, 24, 18, 192), (2, 12, 9, 384), (1, 48, 36, 192), (5, 6, 5, 384)])
def test_deep_3x3_layers_with_32_channel_slabs_equal_the_64_channel_form(shape):
    """tile_cfg -8 (HipHRNet.slab32, forwards of a few crops): the 192- / 384-channel 3x3 layers with 32-channel slabs -- twice the
    workgroups, the same K order per output channel -> bit-identical to the 64-channel-slab launch, with and without the residual."""
    from pam import _lib, hrnet_hip
    n, h, w, c = shape
    DEV = torch.device('cuda:0')
    e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = DEV; e.tile_cfg = -1
    torch.manual_seed(3)
    conv = nn.Conv2d(c, c, 3, 1, 1, bias=True)
    op = hrnet_hip.PackedConv(conv, DEV)
    g = torch.Generator().manual_seed(11)
    x = torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    r = torch.randn((n, c, h, w), generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    assert e.lib.pam_conv3x3_layout_small(h, w, c, c) == 32
    for res, relu in ((None, True), (r, True), (None, False)):
        e.slab32 = False
        a = e.conv(op, x, res=res, relu=relu).clone()
        e.slab32 = True
        b = e.conv(op, x, res=res, relu=relu).clone()
        e.slab32 = False
        torch.cuda.synchronize()
        assert e.lib.pam_conv_last_kernel() == 2 and ('s32', 32) in op._images
        assert torch.equal(a, b)
    assert e.lib.pam_conv3x3_layout_small(h, w, 96, 96) == 0
