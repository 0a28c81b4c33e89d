#!/usr/bin/env python3
"""Does the replay time of the captured forward vary between captures (same process) or between processes?  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=True, backend='hip')
n = 20
x = torch.randn((n, 8, 384, 288), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if False else None
def timeit(iters=20):
    xb = net.input_buffer(n)
    for _ in range(3): net.features(xb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(iters): net.features(xb)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
xb = net.input_buffer(n); xb.copy_(torch.randn(xb.shape, device=dev).to(xb.dtype))
res = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    net._graphs.clear()
    xb = net.input_buffer(n)
    t = [timeit() for _ in range(3)]
    res.append(t)
    print('capture %2d: %s ms' % (k, ' '.join('%.3f' % v for v in t)), flush=True)
