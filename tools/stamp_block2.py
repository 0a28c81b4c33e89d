#!/usr/bin/env python3
"""Phase breakdown of the resident-weights BasicBlock kernel (csrc/pam_block2.hip) from per-wave in-kernel s_memtime stamps (development
tool).  Builds a DIAGNOSTIC copy (-DPAM_DIAG [+ extra -D flags]) into /tmp and loads it beside the product library."""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--defs', default=''); ap.add_argument('--tile', default=''); ap.add_argument('--c', type=int, default=48)
args = ap.parse_args()
csrc = os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc')
so = '/tmp/libbb2_diag_%d.so' % os.getpid()
subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-DPAM_DIAG'] +
                      ['-D' + d for d in args.defs.split(',') if d] + ['-shared', os.path.join(csrc, 'pam_block2.hip'), '-o', so])
import numpy as np, torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
lib = C.CDLL(so)
lib.pam_basic_block2_nhwc_bf16.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6
dev = torch.device('cuda:0')
c = args.c
h, w = {48: (96, 72), 96: (48, 36)}[c]
op = hrnet_hip.PackedBlock(nn.Conv2d(c, c, 3, 1, 1), nn.Conv2d(c, c, 3, 1, 1), dev)
x = torch.randn((args.n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
y = torch.empty_like(x)
tr, tc = [int(v) for v in args.tile.split('x')] if args.tile else (0, 0)


def run():
    rc = lib.pam_basic_block2_nhwc_bf16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(op.wpack.data_ptr()),
                                        C.c_void_p(y.data_ptr()), args.n, h, w, c, tr, tc)
    assert rc == 0, rc


stamps = torch.zeros((8192, 8, 8), dtype=torch.int64, device=dev)
lib.pam_block2_debug_stamps(None)
for _ in range(5): run()
torch.cuda.synchronize()
lib.pam_block2_debug_stamps(C.c_void_p(stamps.data_ptr()))
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record(); run(); b.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy()
s = s[s[:, 0, 0] != 0]
t0 = s[:, :, 0].min()
print('%d items, launch %.1f us (event); first start -> last end %d ticks; start spread %d' % (len(s), a.elapsed_time(b) * 1e3, s[:, :, 7].max() - t0, s[:, :, 0].max() - t0))
names = ['DMA issue', 'wait X + W1a', 'conv1', 'residual + barrier', 'mid write + barrier', 'conv2', 'epilogue']
for k, nm in enumerate(names):
    d = s[:, :, k + 1] - s[:, :, k]
    print('   %-20s median %6.0f  p10 %6.0f  p90 %6.0f   | wave medians %s' % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), ' '.join('%5.0f' % v for v in np.median(d, axis=0))))
tot = s[:, :, 7].max(axis=1) - s[:, :, 0].min(axis=1)
print('   %-20s median %6.0f  p10 %6.0f  p90 %6.0f' % ('item', np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
os.remove(so)
