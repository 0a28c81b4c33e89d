#!/usr/bin/env python3
"""Does a pinned host -> device copy on its own stream overlap the conv-stack replay on this platform?  (development tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=True)
x = net.input_buffer(20); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
for _ in range(5): net.features(x)
torch.cuda.synchronize()
host = torch.empty(12 * 1024 * 1024, dtype=torch.uint8).pin_memory()
dbuf = torch.empty_like(host, device=dev)
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cs = torch.cuda.Stream(dev, priority=prio)


def run(copy, iters=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if copy == 'side':
            with torch.cuda.stream(cs):
                dbuf.copy_(host, non_blocking=True)
        elif copy == 'same':
            dbuf.copy_(host, non_blocking=True)
        net.features(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def copy_only(iters=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for _ in range(iters):
            dbuf.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for name in ('none', 'side', 'same', 'none', 'side'):
    print('%-5s %.3f ms per iteration' % (name, run(name)), flush=True)
print('copy alone: %.3f ms per 12 MiB copy (priority %d stream)' % (copy_only(), prio))
