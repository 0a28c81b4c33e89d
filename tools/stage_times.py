#!/usr/bin/env python3
"""Wall time of the HRNet forward cut after the stem / layer1 / stage2 / stage3 / whole (hipGraph replay of each prefix; the
differences are the stages' wall times inside the multi-stream plan).  Development tool."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--config', default=''); ap.add_argument('--no-branch-streams', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False, backend='hip')
hip = net.hip
if args.config:
    hip.apply_config(args.config)
hip.multi_stream = not args.no_branch_streams
x = net.input_buffer(args.n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
prev = 0.0
for stop in ('stem', 'layer1', 'stage2', 'stage3', None):
    hip.stop_after = stop
    hip.prof = None
    hip.features(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y = hip.features(x)
    run, info = g.replay, 'captured hipGraph'
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(args.iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print('%-8s prefix %.3f ms   (+%.3f)   %s' % (stop or 'all', ms, ms - prev, info), flush=True)
    prev = ms
