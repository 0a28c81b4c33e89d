#!/usr/bin/env python3
"""The 96 -> 96 3x3 layer at 48 x 36 on k_conv3x3 and on the streamed kernel with 48- / 96-channel slabs (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device('cuda:0')
ns = [int(v) for v in sys.argv[1:]] or [20]
conv = nn.Conv2d(96, 96, 3, 1, 1)
op = hrnet_hip.PackedConv(conv, dev)


def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev); keep = []
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters): keep.append(fn())
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


for n in ns:
    x = torch.randn((n, 96, 48, 36)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn((n, 96, 48, 36)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * n * 48 * 36 * 96 * 96 * 9
    for slab in (0, 48, 96):
        e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1; e.c96_slab = slab
        t = timeit(lambda: e.conv(op, x, res=r, relu=True))
        print('n=%3d slab %2d: %6.1f us (%4.0f TF/s)' % (n, slab, t, fl / t / 1e6), flush=True)
