#!/usr/bin/env python3
"""GPU micro-benchmark of the HRNet-W48 conv stack variants (development tool)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--modes', default='graph')
ap.add_argument('--backends', default='hip')
ap.add_argument('--no-branch-streams', action='store_true')
ap.add_argument('--order', default='0,1,2,3')
ap.add_argument('--config', default='', help='executor configuration (HipHRNet.CONFIGS); empty = the flags below'); ap.add_argument('--merge', type=int, default=2, help='0: no merged fuse convs, 1: strided chains only, 2: + up-convs')
args = ap.parse_args()
dev = torch.device('cuda:0')
flops = hrnet.count_flops() * args.n


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for backend in args.backends.split(','):
    for mode in args.modes.split(','):
        torch.backends.cudnn.benchmark = (mode == 'bench')
        t0 = time.time()
        net = hrnet.HRNetPose(48, 17, None, use_graph=(mode == 'graph'), backend=backend)
        if backend == 'hip':
            net.hip.multi_stream = not args.no_branch_streams
            net.hip.order = tuple(int(q) for q in args.order.split(','))
            net.hip.merge_fuse = args.merge >= 1; net.hip.merge_up = args.merge >= 2
            if args.config:
                net.hip.apply_config(args.config)
        x = net.input_buffer(args.n)
        x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
        net.heatmaps(x); torch.cuda.synchronize()
        t1 = time.time()
        ms = timeit(lambda: net.heatmaps(x), args.iters)
        print(backend, '%-6s N=%d  first-call %.1fs  %.3f ms/forward  %.1f TFLOP/s' % (mode, args.n, t1 - t0, ms, flops / ms / 1e9), flush=True)
