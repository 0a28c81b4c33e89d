#!/usr/bin/env python3
"""Sweep (rows-per-tile, MT, waves) of k_conv3x3 per HRNet layer shape; graph-replayed timing (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--quick', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record(); g.replay(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * reps) * 1e3


for (h, w, cin, cout) in [(96, 72, 48, 48), (48, 36, 96, 96), (24, 18, 192, 192), (12, 9, 384, 384), (96, 72, 64, 64)]:
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((args.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((args.n, cout, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * args.n * h * w * cout * cin * 9
    e.tile_cfg = -1
    t_auto = graph_time(lambda: e.conv(op, x, res=res, relu=True))
    out = []
    for cfg in (42, 43, 44, 82, 83):
        cap = 16 * (cfg // 10) * (cfg % 10)
        for th in range(1, h + 1):
            if th * (w + 2) > cap:
                break
            if args.quick and th * (w + 2) < cap * 0.6:
                continue
            e.tile_cfg = 1000 + th * 100 + cfg
            try:
                out.append((graph_time(lambda: e.conv(op, x, res=res, relu=True)), th, cfg))
            except Exception as ex:
                pass
    out.sort()
    print('%3dx%-3d C%-3d auto %.1f us (%.0f TF/s) | best: %s' % (h, w, cin, t_auto, fl / t_auto / 1e6,
          '  '.join('th%d/cfg%d %.1f' % (th, cfg, t) for t, th, cfg in out[:6])), flush=True)
