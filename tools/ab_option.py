#!/usr/bin/env python3
"""Interleaved A/B of a pam_conv_option value on the whole conv stack (development tool): the n-crop forward is captured once per
option value (the choice is made at launch time, so a capture bakes it in) in the given executor configuration, then the captures
are replayed alternately.  usage: tools/ab_option.py <option key> [crops=20] [config=streamed96] [rounds=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet, _lib
key = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
config = sys.argv[3] if len(sys.argv) > 3 else 'streamed96'; rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
lib, hip = net.lib, net.hip
hip.apply_config(config)
x = net.input_buffer(n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
caps, outs = {}, {}
for v in (0, 1):
    old = lib.pam_conv_option(key, v)
    with torch.no_grad():
        hip.features(x); torch.cuda.synchronize()
        g = _lib.new_graph(); s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                outs[v] = hip.features(x)
    lib.pam_conv_option(key, old)
    caps[v] = g
for v in (0, 1):
    caps[v].replay()
torch.cuda.synchronize()
d = (outs[0].float() - outs[1].float()).abs().max().item(); ref = outs[0].float().abs().max().item()
print('max |difference| of the stack output between the two forms: %.4g (max |value| %.4g)' % (d, ref))
ms = {0: [], 1: []}
for r in range(rounds):
    for v in (0, 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            caps[v].replay()
        b.record(); b.synchronize()
        ms[v].append(a.elapsed_time(b) / 20)
for v in (0, 1):
    print('option %d = %d: %s  median %.4f ms' % (key, v, ' '.join('%.4f' % t for t in ms[v]), sorted(ms[v])[len(ms[v]) // 2]))
