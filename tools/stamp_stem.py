#!/usr/bin/env python3
"""Phase breakdown of the fused stem kernel (csrc/pam_stem.hip) from per-wave in-kernel s_memtime stamps (development tool).  Builds a
DIAGNOSTIC copy (-DPAM_DIAG [+ extra -D flags]) into /tmp and loads it beside the product library."""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--defs', default='')
args = ap.parse_args()
csrc = os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc')
so = '/tmp/libstem_diag_%d.so' % os.getpid()
subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-DPAM_DIAG'] +
                      ['-D' + d for d in args.defs.split(',') if d] + ['-shared', os.path.join(csrc, 'pam_stem.hip'), '-o', so])
import numpy as np, torch
import pam
from pam import _lib, hrnet_hip
from test_gpu_stem import make_stem
lib = C.CDLL(so)
lib.pam_stem_fused_nhwc_bf16.argtypes = [C.c_void_p] * 10 + [C.c_int] * 3
dev = torch.device('cuda:0')
c1, c2, pw = make_stem(1)
P1, Pp = hrnet_hip.PackedConv(c1, dev, pad_cin_to=8), hrnet_hip.PackedPointwise64(pw, dev)
op = hrnet_hip.PackedStem(P1, c2, Pp, dev)
x8 = torch.zeros((args.n, 8, 384, 288)); x8[:, :3] = torch.randn((args.n, 3, 384, 288))
x8 = x8.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
x0 = torch.empty((args.n, 64, 96, 72), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last); y1 = torch.empty_like(x0)
P = lambda t: C.c_void_p(t.data_ptr())


def run():
    rc = lib.pam_stem_fused_nhwc_bf16(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(x8), P(op.c1._stem), P(op.c1.bias), P(op.w2), P(op.b2), P(op.pw.w), P(op.pw.b),
                                      P(x0), P(y1), args.n, 384, 288)
    assert rc == 0, rc


stamps = torch.zeros((256, 8, 6, 8), dtype=torch.int64, device=dev)
lib.pam_stem_debug_stamps(None)
for _ in range(5): run()
torch.cuda.synchronize()
lib.pam_stem_debug_stamps(C.c_void_p(stamps.data_ptr()))
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record(); run(); b.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy()
t0 = s[:, :, 0, 0][s[:, :, 0, 0] != 0].min()
print('launch %.1f us (event); 100 MHz ticks' % (a.elapsed_time(b) * 1e3))
names = ['conv1 -> LDS', 'wait + barrier', 'conv2 K loop', 'epilogue + pointwise + stores', 'end barrier']
for it in range(6):
    v = s[:, :, it]
    v = v[v[:, :, 0] != 0]
    if not len(v): break
    print('item %d of a workgroup (%d waves): starts at %d .. %d' % (it, len(v), v[:, 0].min() - t0, v[:, 0].max() - t0))
    for k, nm in enumerate(names):
        d = v[:, k + 1] - v[:, k]
        print('   %-32s median %6.0f  p10 %6.0f  p90 %6.0f' % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print('last end %d ticks after first start' % (s[:, :, :, 5].max() - t0))
os.remove(so)
