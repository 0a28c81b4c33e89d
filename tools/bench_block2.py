#!/usr/bin/env python3
"""Fused BasicBlock kernels (k_bblock2_48 / k_bblock2_96) vs the two-launch path, over crop counts and item tiles (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip

ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, nargs='+', default=[20]); ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--tiles', type=str, default='')          # e.g. 16x36,24x24
ap.add_argument('--c', type=int, default=48)
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1


def timeit(fn, iters=args.iters):
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
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


c = args.c
h, w = {48: (96, 72), 96: (48, 36)}[c]
c1, c2 = nn.Conv2d(c, c, 3, 1, 1), nn.Conv2d(c, c, 3, 1, 1)
op = hrnet_hip.PackedBlock(c1, c2, dev)
p1, p2 = hrnet_hip.PackedConv(c1, dev), hrnet_hip.PackedConv(c2, dev)
tiles = [None] + [tuple(int(v) for v in t.split('x')) for t in args.tiles.split(',') if t]
for n in args.n:
    x = torch.randn((n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    fl = 2 * 2.0 * n * h * w * c * c * 9
    e.c96_slab = 48
    tu = timeit(lambda: e.conv(p2, e.conv(p1, x, relu=True), res=x, relu=True))
    print('n=%3d  two launches %6.1f us (%4.0f TF/s)' % (n, tu, fl / tu / 1e6), flush=True)
    for t in tiles:
        try:
            tt = timeit(lambda: e.basic_block2(op, x, t))
        except Exception as ex:
            print('   tile', t, 'failed:', ex); continue
        print('   resident, tile %-10s %6.1f us (%4.0f TF/s)' % (t if t else e._bb2_tiles.get((c, n, h, w)), tt, fl / tt / 1e6), flush=True)
