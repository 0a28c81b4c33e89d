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
a = ap.parse_args()
h, w, cin, cout, k, s = [int(x) for x in a.shape.split(',')]
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = a.tile
if a.block:
    bop = hrnet_hip.PackedBlock(nn.Conv2d(cin, cin, 3, 1, 1), nn.Conv2d(cin, cin, 3, 1, 1), dev)
    xb = torch.randn((a.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    for _ in range(a.iters):
        yb = e.basic_blocks([bop], [xb], 8)
    torch.cuda.synchronize()
    sys.exit(0)
conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=True)
op = hrnet_hip.PackedConv(conv, dev)
x = torch.randn((a.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
res = torch.randn((a.n, cout, h // s, w // s)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last) if a.res else None
for _ in range(a.iters):
    y = e.conv(op, x, res=res, relu=True)
torch.cuda.synchronize()
