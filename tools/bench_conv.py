#!/usr/bin/env python3
"""Per-layer timing of the MFMA conv kernel vs MIOpen (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
import pam
from pam import _lib, hrnet_hip

ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--tiles', default='-1'); ap.add_argument('--generic-all', action='store_true'); ap.add_argument('--no-miopen', action='store_true'); ap.add_argument('--fuse', action='store_true', help='the strided / 1x1 convolutions of the fuse layers (incl. merged ones)')
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev


def timeit(fn, iters=30):
    """us per call on the GPU: `iters` calls captured into one hipGraph and replayed between two events (a Python launch costs ~20 us:
    eager timing cannot resolve these kernels)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        for _ in range(iters): keep.append(fn())
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


LAYERS = [(96, 72, 48, 48, 3, 1), (48, 36, 96, 96, 3, 1), (24, 18, 192, 192, 3, 1), (12, 9, 384, 384, 3, 1),
          (96, 72, 64, 64, 3, 1), (96, 72, 256, 64, 1, 1), (96, 72, 64, 256, 1, 1), (192, 144, 64, 64, 3, 2), (384, 288, 8, 64, 3, 2),
          (48, 36, 96, 48, 1, 1), (96, 72, 48, 96, 3, 2), (12, 9, 384, 48, 1, 1)]
if args.generic_all:          # every layer shape of HRNet-W48 that runs on k_conv_igemm (h, w = input size)
    LAYERS = [(48, 36, 96, 192, 3, 2), (96, 72, 48, 96, 3, 2), (96, 72, 64, 256, 1, 1), (96, 72, 256, 96, 3, 2), (96, 72, 256, 64, 1, 1),
              (96, 72, 48, 48, 3, 2), (192, 144, 64, 64, 3, 2), (24, 18, 192, 384, 3, 2), (48, 36, 48, 192, 3, 2), (24, 18, 96, 384, 3, 2),
              (48, 36, 96, 96, 3, 2), (48, 36, 96, 48, 1, 1), (24, 18, 192, 96, 1, 1), (24, 18, 48, 384, 3, 2), (96, 72, 64, 64, 1, 1),
              (24, 18, 192, 48, 1, 1), (48, 36, 48, 48, 3, 2), (12, 9, 384, 192, 1, 1), (12, 9, 384, 96, 1, 1), (12, 9, 384, 48, 1, 1)]
if args.fuse:
    LAYERS = [(96, 72, 48, 144, 3, 2), (96, 72, 48, 192, 3, 2), (96, 72, 48, 96, 3, 2), (48, 36, 96, 192, 3, 2), (48, 36, 96, 288, 3, 2), (48, 36, 48, 192, 3, 2), (48, 36, 48, 48, 3, 2),
              (24, 18, 192, 384, 3, 2), (24, 18, 96, 384, 3, 2), (24, 18, 48, 384, 3, 2), (48, 36, 96, 48, 1, 1), (24, 18, 192, 144, 1, 1), (12, 9, 384, 336, 1, 1),
              (192, 144, 64, 64, 3, 2), (96, 72, 256, 96, 3, 2), (96, 72, 64, 256, 1, 1), (96, 72, 256, 64, 1, 1), (96, 72, 64, 64, 1, 1)]
for (h, w, cin, cout, k, s) in LAYERS:
    conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=True)
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((args.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * args.n * ho * wo * cout * cin * k * k
    mi = conv.to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    with torch.no_grad():
        t_mi = 1.0 if args.no_miopen else timeit(lambda: torch.relu(mi(x)))
    line = '%3dx%-3d %3d->%-3d k%d s%d  %6.2f GF  miopen+relu %7.1f us (%6.1f TF/s) |' % (h, w, cin, cout, k, s, fl / 1e9, t_mi, fl / t_mi / 1e6)
    for t in [int(q) for q in args.tiles.split(',')]:
        e.tile_cfg = t
        try:
            us = timeit(lambda: e.conv(op, x, relu=True))
            line += ' cfg%d %7.1f us (%6.1f TF/s)' % (t, us, fl / us / 1e6)
        except Exception as ex:
            line += ' cfg%d n/a' % t
    print(line, flush=True)
