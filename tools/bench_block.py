#!/usr/bin/env python3
"""Fused BasicBlock kernel vs the two-launch path, per HRNet-W48 branch and grouped (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip

ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1


def timeit(fn, iters=args.iters):
    """us per call on the GPU (graph replay: a Python launch costs ~20 us, eager timing cannot resolve these kernels)."""
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


SH = [(48, 96, 72), (96, 48, 36), (192, 24, 18)]
ops, xs = [], []
for c, h, w in SH:
    c1, c2 = nn.Conv2d(c, c, 3, 1, 1), nn.Conv2d(c, c, 3, 1, 1)
    x = torch.randn((args.n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    op = hrnet_hip.PackedBlock(c1, c2, dev)
    p1, p2 = hrnet_hip.PackedConv(c1, dev), hrnet_hip.PackedConv(c2, dev)
    fl = 2 * 2.0 * args.n * h * w * c * c * 9
    tf = timeit(lambda: e.basic_blocks([op], [x], 8))
    t4 = timeit(lambda: e.basic_blocks([op], [x], 4)) if e.lib.pam_basic_block_rows(c, h, w, 4) > 0 else float('nan')
    tu = timeit(lambda: e.conv(p2, e.conv(p1, x, relu=True), res=x, relu=True))
    print('C=%3d %2dx%-2d  fused 8 waves %6.1f us (%5.0f TF/s)  4 waves x 2/CU %6.1f us (%5.0f TF/s)   two launches %6.1f us (%5.0f TF/s)' % (c, h, w, tf, fl / tf / 1e6, t4, fl / t4 / 1e6, tu, fl / tu / 1e6), flush=True)
    ops.append(op); xs.append(x)
for k, wv in ((2, 4), (2, 8), (3, 8)):
    fl = sum(2 * 2.0 * args.n * h * w * c * c * 9 for c, h, w in SH[:k])
    tg = timeit(lambda: e.basic_blocks(ops[:k], xs[:k], wv))
    print('grouped %d branches, %d waves: %6.1f us (%5.0f TF/s)' % (k, wv, tg, fl / tg / 1e6), flush=True)
for mask in (1, 2, 3):
    fl = sum(2 * 2.0 * args.n * h * w * c * c * 9 for c, h, w in SH[:2])
    tg = timeit(lambda: e.basic_blocks(ops[:2], xs[:2], 8 | (mask << 4)))
    print('grouped 2 branches, 8 waves, short-item mask %d: %6.1f us (%5.0f TF/s)' % (mask, tg, fl / tg / 1e6), flush=True)
