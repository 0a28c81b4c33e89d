#!/usr/bin/env python3
"""Graph-replayed timing of small generic-kernel convolutions vs K (fixed cost vs per-chunk cost; development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device('cuda:0')
e = hrnet_hip.ConvEngine(); e.lib = _lib.load(); e.device = dev


def graph_time(fn, reps=40):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record(); g.replay(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * reps) * 1e3


for (h, w, k, s, cout) in [(12, 9, 1, 1, 48), (24, 18, 1, 1, 48), (48, 36, 3, 2, 48), (96, 72, 3, 2, 48), (12, 9, 3, 1, 48)]:
    line = '%3dx%-3d k%d s%d ->%d:' % (h, w, k, s, cout)
    for cin in (64, 128, 256, 384):
        conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=True); op = hrnet_hip.PackedConv(conv, dev)
        op._w_ohwi = None                                   # force the generic kernel
        x = torch.randn((20, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
        t = graph_time(lambda: e.conv(op, x, relu=True))
        line += '  Cin %3d (%2d chunks) %5.1f us' % (cin, (k * k * cin + 63) // 64, t)
    print(line, flush=True)
