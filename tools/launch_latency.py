#!/usr/bin/env python3
"""What a SYNCHRONOUS caller pays for a hipGraph launch (development tool): wall time of [replay; synchronize] from an idle GPU against
the GPU time of the same replay (HIP events), for the whole forward as one graph and as two (stem + layer1 | stages 2-4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
hip = net.hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = net.input_buffer(n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
hip.features(x); torch.cuda.synchronize()


def cap(fn):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = fn()
    return g, out


g_all, out_all = cap(lambda: hip.features(x))
hip._keep = []
g_head, y = cap(lambda: hip._head(x))
g_body, out_b = cap(lambda: hip._body(y))


def measure(run, reps=30):
    for _ in range(3): run(); torch.cuda.synchronize()
    wall, gpu = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        t0 = time.perf_counter()
        a.record(); run(); b.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        wall.append((t2 - t0) * 1e6); gpu.append(a.elapsed_time(b) * 1e3)
    return np.median(wall), np.median(gpu)


for name, run in (('one graph', g_all.replay), ('two graphs (head | body)', lambda: (g_head.replay(), g_body.replay()))):
    w, g = measure(run)
    print('%-26s wall %7.1f us   GPU (event to event) %7.1f us   difference %6.1f us' % (name, w, g, w - g))
