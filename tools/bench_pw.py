#!/usr/bin/env python3
"""k_pw2 (the fused pointwise tail of a layer1 Bottleneck) alone on the chip, per wave-tile size, against the launches it replaces
(conv3 1x1 + residual, next conv1 1x1 [, downsample 1x1]).  Development tool."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.ConvEngine(); e.lib = _lib.load(); e.device = dev


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


n, h, w = args.n, 96, 72
cl = lambda c: torch.randn((n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
y2, x0, res = cl(64), cl(64), cl(256)
conv3, down, conv1 = nn.Conv2d(64, 256, 1), nn.Conv2d(64, 256, 1), nn.Conv2d(256, 64, 1)
M = n * h * w
for name, first, use_res, second in (('middle block (res + next conv1)', False, True, True), ('first block (downsample + next conv1)', True, False, True),
                                     ('last block (res only)', False, True, False)):
    op = hrnet_hip.PackedTail(conv3, down if first else None, conv1 if second else None, dev)
    by = 2 * M * (64 + (64 if first else 0) + (256 if use_res else 0) + 256 + (64 if second else 0))
    for cfg in (1, 21, 11, 2):
        e._keep = []
        us = timeit(lambda: e.bottleneck_tail(op, y2, x0 if first else None, res if use_res else None, cfg), args.iters)
        print('%-40s cfg=%2d  %7.1f us  %6.0f GB/s algorithmic' % (name, cfg, us, by / us / 1e3), flush=True)
    # what it replaces
    c3, c1, cd = hrnet_hip.PackedConv(conv3, dev), hrnet_hip.PackedConv(conv1, dev), hrnet_hip.PackedConv(down, dev)
    e._keep = []
    e.tile_cfg = -1
    def old():
        r = e.conv(cd, x0) if first else (res if use_res else None)
        x = e.conv(c3, y2, res=r, relu=True)
        if second:
            e.conv(c1, x, relu=True)
    print('%-40s separate launches  %7.1f us' % (name, timeit(old, args.iters)), flush=True)
