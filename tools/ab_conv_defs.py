#!/usr/bin/env python3
"""Interleaved A/B of csrc/pam_conv.hip built with extra -D flags against the library, per HRNet layer shape (development tool).
usage: ab_conv_defs.py --defs PAM_SWAP_ROLES [--n 20]"""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--defs', required=True); ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
csrc = os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc')
so = '/tmp/libconv_var_%d.so' % os.getpid()
subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off'] + ['-D' + d for d in args.defs.split(',') if d] +
                      ['-shared', os.path.join(csrc, 'pam_conv.hip'), '-o', so])
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device('cuda:0')
base = _lib.load()
var = C.CDLL(so)
for name, (res, at) in _lib._SIGS.items():
    if hasattr(var, name):
        f = getattr(var, name); f.restype = res; f.argtypes = at
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = base; e.device = dev; e.tile_cfg = -1; e.c96_slab = 48


class Both(object):
    """the variant's conv entry points, everything else from the library"""
    def __init__(self, var, base): self.var, self.base = var, base
    def __getattr__(self, k): return getattr(self.var, k) if hasattr(self.var, k) else getattr(self.base, k)


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph(); keep = []
    with torch.cuda.graph(g):
        for _ in range(iters): keep.append(fn())
    g.replay(); torch.cuda.synchronize()
    return g


LAYERS = [(24, 18, 192, 192, 3, 1, 1), (12, 9, 384, 384, 3, 1, 1), (96, 72, 256, 48, 3, 1, 0), (96, 72, 48, 144, 3, 2, 0), (48, 36, 96, 288, 3, 2, 0),
          (24, 18, 192, 384, 3, 2, 0), (96, 72, 256, 96, 3, 2, 0), (12, 9, 384, 336, 1, 1, 0), (24, 18, 192, 144, 1, 1, 0)]
for (h, w, cin, cout, k, s, with_res) in LAYERS:
    conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=True)
    op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((args.n, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((args.n, cout, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last) if with_res else None
    graphs = []
    for lib in (base, Both(var, base)):
        e.lib = lib; op.layout_lib = lib
        graphs.append(timeit(lambda: e.conv(op, x, res=res, relu=True), args.iters))
    t = [[], []]
    for r in range(7):
        for i, g in enumerate(graphs):
            a, b = torch.cuda.Event(True), torch.cuda.Event(True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            t[i].append(a.elapsed_time(b) / args.iters * 1e3)
    m = [sorted(v)[len(v) // 2] for v in t]
    print('%3dx%-3d %3d->%-3d k%d s%d   library %6.2f us   [%s] %6.2f us   %+.1f %%' % (h, w, cin, cout, k, s, m[0], args.defs, m[1], 100 * (m[1] / m[0] - 1)), flush=True)
os.remove(so)
