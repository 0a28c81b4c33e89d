#!/usr/bin/env python3
"""Throughput of K independent HRNet forwards in flight on K streams (frame-level pipelining experiment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
N = 20
flops = hrnet.count_flops() * N
for K in (1, 2, 3):
    nets = [hrnet.HRNetPose(48, 17, None, use_graph=True) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    xs = []
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            x = nets[k].input_buffer(N); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
            nets[k].heatmaps(x); x = nets[k].input_buffer(N); xs.append(x)
    torch.cuda.synchronize()
    iters = 30
    t0 = time.perf_counter()
    for it in range(iters):
        k = it % K
        with torch.cuda.stream(streams[k]):
            nets[k].heatmaps(xs[k])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('K=%d streams: %.3f ms per forward (throughput), %.1f TFLOP/s' % (K, dt * 1e3, flops / dt / 1e12), flush=True)
