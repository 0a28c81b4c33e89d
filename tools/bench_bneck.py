#!/usr/bin/env python3
"""us per call: the fused layer1 Bottleneck (k_bneck) against the two launches it replaces, captured 20x into one graph each (development tool)."""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
_ap = argparse.ArgumentParser(add_help=False); _ap.add_argument('--defs', default=''); _known, _ = _ap.parse_known_args()
variants = []
for i, grp in enumerate([g for g in _known.defs.split(':') if g]):       # --defs A=1,B=2[:C=3 ...]: private builds of csrc/pam_bneck.hip, timed beside the library's
    so = '/tmp/libbneck_var_%d_%d.so' % (os.getpid(), i)
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off'] + ['-D' + d for d in grp.split(',') if d] +
                          ['-shared', os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc', 'pam_bneck.hip'), '-o', so])
    variants.append((grp, so))
import torch
import pam
from pam import _lib, hrnet_hip
from test_gpu_bneck import make_convs

ap = argparse.ArgumentParser(); ap.add_argument('--n', default='20'); ap.add_argument('--iters', type=int, default=20); ap.add_argument('--defs', default='')
args = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1; e.c96_slab = 48
c2, c3, c1n, down = make_convs(1)
P2 = hrnet_hip.PackedConv(c2, dev)
libs = []
for grp, so in variants:
    l = C.CDLL(so); l.pam_bottleneck_fused_nhwc_bf16.argtypes = [C.c_void_p] * 12 + [C.c_int] * 3
    libs.append((grp, l))


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
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


for has2, first in ((True, False), (False, False), (True, True)):
    tail = hrnet_hip.PackedTail(c3, down if first else None, c1n if has2 else None, dev)
    op = hrnet_hip.PackedBneck(c2, tail, dev)
    for n in [int(v) for v in args.n.split(',')]:
        cl = lambda t: t.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
        y1 = cl(torch.relu(torch.randn((n, 64, 96, 72))))
        res = cl(torch.relu(torch.randn((n, 256, 96, 72)))) if not first else None
        x0 = cl(torch.relu(torch.randn((n, 64, 96, 72)))) if first else None
        two = lambda: e.bottleneck_tail(tail, e.conv(P2, y1, relu=True), x0, res, 0)
        t3 = timeit(lambda: e.conv(P2, y1, relu=True), args.iters)
        extra = ''
        for grp, l in libs:
            lib0 = e.lib; e.lib = l
            try:
                extra += '   [%s] %7.1f us' % (grp, timeit(lambda: e.bottleneck_fused(op, y1, res, x0), args.iters))
            finally:
                e.lib = lib0
        print('n=%3d next-conv1=%d first=%d  3x3 alone %6.1f us, two launches %7.1f us   fused %7.1f us' % (n, has2, first, t3, timeit(two, args.iters), timeit(lambda: e.bottleneck_fused(op, y1, res, x0), args.iters)) + extra, flush=True)
