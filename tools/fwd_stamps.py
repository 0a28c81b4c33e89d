#!/usr/bin/env python3
"""UN-profiled timeline of one replay of the captured forward: one-wave stamp kernels (tools/micro/stamp_kernel.hip, built into /tmp) at
the schedule's points of interest -- start / end of every branch's blocks, end of its fuse tail, the join, every sum -- store the 100 MHz
clock.  Prints per module the times relative to the module's first stamp.  usage: fwd_stamps.py [--n 20] attr=val,attr=val"""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('spec', nargs='?', default='')
args = ap.parse_args()
so = '/tmp/libstamp_%d.so' % os.getpid()
subprocess.check_call(['hipcc', '-O2', '-fPIC', '--offload-arch=gfx950', '-shared', os.path.join(ROOT, 'tools', 'micro', 'stamp_kernel.hip'), '-o', so])
import numpy as np, torch
import pam
from pam import hrnet
lib = C.CDLL(so); lib.stamp_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
dev = torch.device('cuda:0')
net = hrnet.HRNetPose(48, 17, None, use_graph=False)
hip = net.hip
for kv in [q for q in args.spec.split(',') if q]:
    k, _, val = kv.partition('=')
    if k in ('b96_tile', 'b48_tile'):
        setattr(hip, k, tuple(int(q) for q in val.split('x'))); continue
    cur = getattr(hip, k)
    setattr(hip, k, tuple(int(c) for c in val) if isinstance(cur, tuple) else (int(val) if cur is None else type(cur)(int(val))))
x = net.input_buffer(args.n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
hip.features(x); torch.cuda.synchronize()
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
tags = []


def stamp(tag):
    lib.stamp_launch(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.c_void_p(buf.data_ptr()), len(tags)); tags.append(tag)


hip.stamp = stamp
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        stamp('begin'); out = hip.features(x); stamp('end')
for _ in range(10): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
t = buf.cpu().numpy()[:len(tags)].astype(np.float64) / 100.0          # us
t -= t[0]
print('replay %.3f ms (with %d stamp kernels); begin -> end %.1f us' % (e0.elapsed_time(e1), len(tags), t[-1]))
mod, m0 = 0, None
line = []
for tag, v in zip(tags, t):
    if tag == 'b0 start':
        if line: print('  module %d (starts %.1f): ' % (mod, m0) + '  '.join(line)); mod += 1
        m0, line = v, []
    if m0 is None:
        print('  %-10s %8.1f' % (tag, v)); continue
    if tag == 'end':
        print('  module %d (starts %.1f): ' % (mod, m0) + '  '.join(line)); print('  end %.1f' % v); break
    line.append('%s %.0f' % (tag, v - m0))
os.remove(so)
