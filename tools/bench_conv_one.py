#!/usr/bin/env python3
"""Run ONE conv layer shape repeatedly (for rocprofv3 --pmc passes)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='96,72,48,48,3,1'); ap.add_argument('--n', type=int, default=20)
ap.add_argument('--iters', type=int, default=20); ap.add_argument('--tile', type=int, default=-1); ap.add_argument('--res', type=int, default=1)
ap.add_argument('--block', type=int, default=0, help='1: the fused BasicBlock kernel on (h, w, cin) instead of one convolution')
ap.add_argument('--tail', type=int, default=0, help='1: the fused pointwise tail of a layer1 Bottleneck (k_pw2: conv3 + residual + next conv1) on (h, w); 2: k_pw1 (64 -> 64 pointwise); 3: k_bneck (3x3 + tail + next conv1 in one launch); 4: k_stem_fused on 4h x 4w crops')
a = ap.parse_args()
h, w, cin, cout, k, s = [int(x) for x in a.shape.split(',')]
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = a.tile
if a.block:
    bop = hrnet_hip.PackedBlock(nn.Conv2d(cin, cin, 3, 1, 1), nn.Conv2d(cin, cin, 3, 1, 1), dev)
    xb = torch.randn((a.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    for _ in range(a.iters):
        e._keep = []
        yb = e.basic_block2(bop, xb)
    torch.cuda.synchronize()
    sys.exit(0)
if a.tail:
    cl = lambda c: torch.randn((a.n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    if a.tail == 1:
        top = hrnet_hip.PackedTail(nn.Conv2d(64, 256, 1), None, nn.Conv2d(256, 64, 1), dev)
        y2, r = cl(64), cl(256)
        for _ in range(a.iters):
            e._keep = []
            e.bottleneck_tail(top, y2, None, r, 0)
    elif a.tail == 3:
        top = hrnet_hip.PackedTail(nn.Conv2d(64, 256, 1), None, nn.Conv2d(256, 64, 1), dev)
        bop = hrnet_hip.PackedBneck(nn.Conv2d(64, 64, 3, 1, 1), top, dev)
        y1, r = cl(64), cl(256)
        for _ in range(a.iters):
            e._keep = []
            e.bottleneck_fused(bop, y1, r)
    elif a.tail == 4:
        sop = hrnet_hip.PackedStem(hrnet_hip.PackedConv(nn.Conv2d(3, 64, 3, 2, 1), dev, pad_cin_to=8), nn.Conv2d(64, 64, 3, 2, 1),
                                   hrnet_hip.PackedPointwise64(nn.Conv2d(64, 64, 1), dev), dev)
        x8 = torch.zeros((a.n, 8, 4 * h, 4 * w)); x8[:, :3] = torch.randn((a.n, 3, 4 * h, 4 * w))
        x8 = x8.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
        for _ in range(a.iters):
            e._keep = []
            e.stem_fused(sop, x8)
    else:
        pop = hrnet_hip.PackedPointwise64(nn.Conv2d(64, 64, 1), dev)
        x0 = cl(64)
        for _ in range(a.iters):
            e._keep = []
            e.pointwise64(pop, x0)
    torch.cuda.synchronize()
    sys.exit(0)
conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=True)
op = hrnet_hip.PackedConv(conv, dev)
x = torch.randn((a.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
res = torch.randn((a.n, cout, h // s, w // s)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last) if a.res else None
for _ in range(a.iters):
    y = e.conv(op, x, res=res, relu=True)
torch.cuda.synchronize()
