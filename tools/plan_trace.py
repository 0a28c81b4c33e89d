#!/usr/bin/env python3
"""A few replays of the 20-crop forward in one of its forms, for rocprofv3 --kernel-trace + tools/timeline.py (development tool):
  joined = the one-join-per-module schedule captured as a hipGraph (round-2 form); plan0 / plan1 = the dependency-precise launch plan,
  eager on real streams / as one explicit hipGraph.  Prints GPU ms per forward (HIP events) and host us per replay call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
form = sys.argv[1] if len(sys.argv) > 1 else 'plan0'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
hip = net.hip
x = net.input_buffer(n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
if form == 'joined':
    hip.dag = False
    hip.features(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = hip.features(x)
    run = g.replay
else:
    hip.features(x); torch.cuda.synchronize()
    plan = hip.make_plan(x)
    mode = int(form[-1])
    run = lambda: plan.replay(mode)
s2 = torch.cuda.Stream(dev)
with torch.cuda.stream(s2):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(s2)
    t0 = time.perf_counter()
    for _ in range(iters): run()
    host = (time.perf_counter() - t0) / iters
    e1.record(s2); torch.cuda.synchronize()
print('%s n=%d: %.3f ms per forward (GPU, back to back), host %.0f us per replay call' % (form, n, e0.elapsed_time(e1) / iters, host * 1e6), flush=True)
