#!/usr/bin/env python3
"""Replays the n-crop HRNet forward (captured hipGraph) in one executor variant: the program to put behind rocprofv3 --kernel-trace
(tools/fwd_trace.sh).  usage: fwd_run.py [--n 20] [--replays 12] attr=val,attr=val   (HipHRNet attributes, as tools/ab_flags.py)"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20); ap.add_argument('--replays', type=int, default=12); ap.add_argument('spec', nargs='?', default='')
args = ap.parse_args()
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
hip = net.hip
for kv in [q for q in args.spec.split(',') if q]:
    k, _, val = kv.partition('=')
    if k == 'b96_tile':
        hip.b96_tile = tuple(int(q) for q in val.split('x')); continue
    cur = getattr(hip, k)
    setattr(hip, k, tuple(int(c) for c in val) if isinstance(cur, tuple) else (int(val) if cur is None else type(cur)(int(val))))
x = net.input_buffer(args.n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
hip.features(x); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        out = hip.features(x)
for _ in range(args.replays):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print('last replay %.3f ms' % e0.elapsed_time(e1))
