#!/usr/bin/env python3
"""us per call: the fused stem (k_stem_fused) against the three launches it replaces, captured 20x into one graph each (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import _lib, hrnet_hip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_stem import make_stem

ap = argparse.ArgumentParser(); ap.add_argument('--n', default='20'); ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1
c1, c2, pw = make_stem(1)
P1, P2, Pp = hrnet_hip.PackedConv(c1, dev, pad_cin_to=8), hrnet_hip.PackedConv(c2, dev), hrnet_hip.PackedPointwise64(pw, dev)
op = hrnet_hip.PackedStem(P1, c2, Pp, dev)


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph(); keep = []
    with torch.cuda.graph(g):
        for _ in range(iters): keep.append(fn())
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


for n in [int(v) for v in args.n.split(',')]:
    x8 = torch.zeros((n, 8, 384, 288)); x8[:, :3] = torch.randn((n, 3, 384, 288))
    x8 = x8.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    three = lambda: e.pointwise64(Pp, e.conv(P2, e.conv(P1, x8, relu=True), relu=True))
    print('n=%3d  three launches %7.1f us   fused %7.1f us' % (n, timeit(three, args.iters), timeit(lambda: e.stem_fused(op, x8), args.iters)), flush=True)
