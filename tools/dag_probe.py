#!/usr/bin/env python3
"""Where does the event-based schedule break?  eager dag vs joined schedule, then hipGraph capture per stage (development tool)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
stop = sys.argv[2] if len(sys.argv) > 2 else 'stage3'
mode = sys.argv[3] if len(sys.argv) > 3 else 'capture'
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
hip = net.hip
x = net.input_buffer(n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
hip.stop_after = None if stop == 'all' else stop
var = os.environ.get('VAR', '')
if 'nomerge' in var:
    hip.merge_fuse = False
if 'one' in var:                      # only the first module of stage 3
    hip.stage3 = hip.stage3[:1]
if 'two' in var:
    hip.stage3 = hip.stage3[:2]
if 'nofuseblk' in var:
    hip.fuse_blocks = False
if 'lazymark' in var:                 # record events only for tensors that another stream will read
    _m = hip._mark
    hip._mark_all = False
hip.dag = False
ref = hip.features(x).clone(); torch.cuda.synchronize()
hip.dag = True
y = hip.features(x).clone(); torch.cuda.synchronize()
print('eager dag == joined:', torch.equal(ref, y), flush=True)
if mode == 'capture':
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = hip.features(x)
    print('captured', flush=True)
    for _ in range(3):
        g.replay(); torch.cuda.synchronize()
        print('replay == joined:', torch.equal(ref, out), flush=True)
else:                                     # 'plan0' / 'plan1': the launch plan, eager replay or explicit hipGraph
    plan = hip.make_plan(x)
    print('plan: %d launches, %d events, %d streams' % (plan.launches, plan.events, plan.streams), flush=True)
    for _ in range(3):
        out = plan.replay(int(mode[-1])); torch.cuda.synchronize()
        print('replay == joined:', torch.equal(ref, out), flush=True)
