#!/usr/bin/env python3
"""us per call: the fused stem (k_stem_fused) against the three launches it replaces, captured 20x into one graph each (development tool).
--defs A=1,B=2[:C=3 ...] (e.g. STEM_LOOK=2:STEM_NRES=0): also builds private copies of csrc/pam_stem.hip with those -D flags (one per ':'-separated group) and times them."""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser(); ap.add_argument('--n', default='20'); ap.add_argument('--iters', type=int, default=20); ap.add_argument('--defs', default='')
args = ap.parse_args()
csrc = os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc')
variants = []
for i, grp in enumerate([g for g in args.defs.split(':') if g]):
    so = '/tmp/libstem_var_%d_%d.so' % (os.getpid(), i)
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off'] + ['-D' + d for d in grp.split(',') if d] +
                          ['-shared', os.path.join(csrc, 'pam_stem.hip'), '-o', so])
    variants.append((grp, so))
import torch
import pam
from pam import _lib, hrnet_hip
from test_gpu_stem import make_stem
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1
c1, c2, pw = make_stem(1)
P1, P2, Pp = hrnet_hip.PackedConv(c1, dev, pad_cin_to=8), hrnet_hip.PackedConv(c2, dev), hrnet_hip.PackedPointwise64(pw, dev)
op = hrnet_hip.PackedStem(P1, c2, Pp, dev)
libs = []
for grp, so in variants:
    l = C.CDLL(so); l.pam_stem_fused_nhwc_bf16.argtypes = [C.c_void_p] * 10 + [C.c_int] * 3
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


P = lambda t: C.c_void_p(t.data_ptr())
for n in [int(v) for v in args.n.split(',')]:
    x8 = torch.zeros((n, 8, 384, 288)); x8[:, :3] = torch.randn((n, 3, 384, 288))
    x8 = x8.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    three = lambda: e.pointwise64(Pp, e.conv(P2, e.conv(P1, x8, relu=True), relu=True))
    line = 'n=%3d  three launches %7.1f us   fused %7.1f us' % (n, timeit(three, args.iters), timeit(lambda: e.stem_fused(op, x8), args.iters))
    x0 = torch.empty((n, 64, 96, 72), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last); y1 = torch.empty_like(x0)
    for grp, l in libs:
        def run(l=l):
            rc = l.pam_stem_fused_nhwc_bf16(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(x8), P(op.c1._stem), P(op.c1.bias), P(op.w2), P(op.b2), P(op.pw.w), P(op.pw.b),
                                            P(x0), P(y1), n, 384, 288)
            assert rc == 0, rc
        line += '   [%s] %7.1f us' % (grp, timeit(run, args.iters))
    print(line, flush=True)
for _, so in variants:
    os.remove(so)
