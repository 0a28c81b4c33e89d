#!/usr/bin/env python3
"""us per call of the head + arg-max decode (pam_head_decode: k_head_argmax + k_argmax_finish) on random features, captured 20x into one
graph (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
ap = argparse.ArgumentParser(); ap.add_argument('--n', default='20'); ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
for n in [int(v) for v in args.n.split(',')]:
    f = torch.randn((n, 48, 96, 72), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    view_of = torch.zeros(n, dtype=torch.int32, device=dev); slot_of = torch.arange(n, dtype=torch.int32, device=dev)
    boxes = torch.tensor([[10.0, 20.0, 100.0, 200.0]] * n, dtype=torch.float32, device=dev)
    det = torch.zeros((1, max(n, 8), 17, 3), dtype=torch.float64, device=dev)
    fn = lambda: net.head_decode(f, view_of, slot_of, boxes, det)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(args.iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / args.iters * 1e3)
    print('n=%3d  head + arg-max decode %6.1f us' % (n, best), flush=True)
