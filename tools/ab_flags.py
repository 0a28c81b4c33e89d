#!/usr/bin/env python3
"""Interleaved A/B of executor variants in ONE process on one device (guide rule 24): each variant = HipHRNet attribute overrides,
replayed as a captured hipGraph; R rounds of `iters` replays each, medians and minima.
usage: ab_flags.py [--n 20] name:attr=val,attr=val ...     e.g.  base: r48:block2=1 no1:knock_out=2"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import hrnet
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=20); ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('variants', nargs='+')
args = ap.parse_args()
dev = torch.device('cuda:0')
runs = []
for v in args.variants:
    name, _, spec = v.partition(':')
    net = hrnet.HRNetPose(48, 17, None, use_graph=False)
    hip = net.hip
    for kv in [q for q in spec.split(',') if q]:
        k, _, val = kv.partition('=')
        if k in ('fused_sums', 'slab32'):                                                 # fused_sums: 1 = every output, else a bit mask over the outputs; slab32: 1 | 16
            setattr(hip, k, True if val == '1' else int(val))
        elif k in ('b96_tile', 'b48_tile', 'fs_cap', 'd48_tile'):                            # e.g. b96_tile=12x36
            setattr(hip, k, tuple(int(q) for q in val.split('x')))
        else:
            cur = getattr(hip, k)
            if isinstance(cur, tuple):
                setattr(hip, k, tuple(int(c) for c in val))          # e.g. order=3210, lane_of=0121
            elif cur is None or isinstance(cur, (bool, int)):
                setattr(hip, k, (type(cur)(int(val)) if cur is not None else int(val)))
            else:
                setattr(hip, k, val)
    x = net.input_buffer(args.n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
    hip.features(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = hip.features(x)
    runs.append((name, g.replay, net, (g, x, out)))          # keep the static input / output alive: the graph reads / writes them
for _, run, _, _ in runs:
    for _ in range(5): run()
torch.cuda.synchronize()
t = {name: [] for name, _, _, _ in runs}
for r in range(args.rounds):
    for name, run, _, _ in runs:
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(args.iters): run()
        e1.record(); torch.cuda.synchronize()
        t[name].append(e0.elapsed_time(e1) / args.iters)
base = np.median(t[runs[0][0]])
for name, _, _, _ in runs:
    a = np.array(t[name])
    print('%-14s median %.3f ms  min %.3f  max %.3f   %+.1f %% vs %s' % (name, np.median(a), a.min(), a.max(), 100 * (np.median(a) / base - 1), runs[0][0]), flush=True)
