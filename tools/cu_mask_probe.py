#!/usr/bin/env python3
"""Experiment (round 5): static spatial partitioning of the chip between the four branch chains of the HRNet forward with CU-masked
streams (hipExtStreamCreateWithCUMask), EAGER multi-stream execution (a hipGraph replay does not carry stream CU masks).
usage: cu_mask_probe.py [--n 20] [--shares 74,67,51,64] [--layout block|stride]"""
import os, sys, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import hrnet
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20); ap.add_argument('--shares', default='74,67,51,64'); ap.add_argument('--layout', default='stride')
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL('libamdhip64.so')          # torch's runtime is already mapped: the loader hands back the same library by SONAME
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(cus):
    m = np.zeros(8, dtype=np.uint32)
    for c in cus:
        m[c >> 5] |= np.uint32(1 << (c & 31))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def timeit(net, x, streams, iters):
    hipn = net.hip
    side_saved = hipn.side
    if streams is not None:
        hipn.side = streams[1:]
    cur = streams[0] if streams is not None else torch.cuda.current_stream(dev)
    with torch.cuda.stream(cur):
        for _ in range(3):
            hipn.features(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record(cur)
            for _ in range(iters):
                hipn.features(x)
            e1.record(cur); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
    hipn.side = side_saved
    return float(np.median(ts)), float(np.min(ts))


net = hrnet.HRNetPose(48, 17, None, use_graph=False)
x = net.input_buffer(a.n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
print('eager, plain streams          : median %.3f ms  min %.3f' % timeit(net, x, None, a.iters), flush=True)
allcu = list(range(256))
full = [masked_stream(allcu) for _ in range(4)]
print('eager, 4 streams masked to ALL: median %.3f ms  min %.3f' % timeit(net, x, full, a.iters), flush=True)
shares = [int(q) for q in a.shares.split(',')]
assert sum(shares) <= 256
if a.layout == 'block':
    bounds = np.cumsum([0] + shares)
    parts = [list(range(bounds[i], bounds[i + 1])) for i in range(4)]
else:                                   # interleave: CU c goes to the branch whose cumulative share it falls into, spread over the whole index range
    order = []
    acc = [0.0] * 4
    for c in range(sum(shares)):
        k = int(np.argmin([acc[i] / shares[i] for i in range(4)]))
        acc[k] += 1; order.append(k)
    parts = [[c for c, k in enumerate(order) if k == i] for i in range(4)]
part = [masked_stream(p) for p in parts]
print('eager, partition %s (%s): median %.3f ms  min %.3f' % ((a.shares, a.layout) + timeit(net, x, part, a.iters)), flush=True)
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    net.hip.features(x)
    with torch.cuda.graph(g, stream=s):
        out = net.hip.features(x)
for _ in range(5): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(a.iters): g.replay()
e1.record(); torch.cuda.synchronize()
print('hipGraph replay, plain streams: %.3f ms' % (e0.elapsed_time(e1) / a.iters), flush=True)


def graph_time(streams, tag):
    hipn = net.hip
    side_saved = hipn.side
    hipn.side = streams[1:]
    g2 = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(streams[0]):
            hipn.features(x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g2, stream=streams[0]):
                o2 = hipn.features(x)
        for _ in range(5): g2.replay()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(True), torch.cuda.Event(True)
        a0.record()
        for _ in range(a.iters): g2.replay()
        a1.record(); torch.cuda.synchronize()
        print('hipGraph captured on %s: %.3f ms' % (tag, a0.elapsed_time(a1) / a.iters), flush=True)
    except Exception as ex:
        print('hipGraph captured on %s: failed: %s' % (tag, str(ex)[:200]), flush=True)
    hipn.side = side_saved
    return g2


keep = [graph_time(full, 'streams masked to ALL CUs'), graph_time(part, 'the partition %s (%s)' % (a.shares, a.layout))]
