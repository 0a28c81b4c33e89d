#!/usr/bin/env python3
"""Do four captured branch streams really start together?  Each branch = a chain of tiny convolutions (a handful of workgroups, so
nothing competes for CU resources); run as a hipGraph replay under rocprofv3 --kernel-trace and look at the first start per queue
(development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device('cuda:0')
e = hrnet_hip.ConvEngine(); e.lib = _lib.load(); e.device = dev
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ops, xs = [], []
for cin in (48, 96, 192, 384):
    conv = nn.Conv2d(cin, cin, 1, 1, 0, bias=True); ops.append(hrnet_hip.PackedConv(conv, dev))
    xs.append(torch.randn((1, cin, 12, 9)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last))
side = [torch.cuda.Stream(dev) for _ in range(3)]
keep = []


def run():
    cur = torch.cuda.current_stream(dev)
    for st in side:
        st.wait_stream(cur)
    for b in range(4):
        with torch.cuda.stream(None if b == 0 else side[b - 1]):
            x = xs[b]
            for _ in range(nchain):
                x = e.conv(ops[b], x, relu=True); keep.append(x)
    for st in side:
        cur.wait_stream(st)


run(); torch.cuda.synchronize()
s = torch.cuda.Stream(dev); s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream(dev).wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('done')
