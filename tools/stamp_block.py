#!/usr/bin/env python3
"""Phase breakdown of the fused BasicBlock kernel from in-kernel s_memtime stamps (development tool).  Builds a DIAGNOSTIC copy of
csrc/pam_block.hip (-DPAM_DIAG [+ extra -D flags]) into /tmp and loads it beside the product library; the shipped library has no stamps."""
import os, sys, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=20); ap.add_argument('--defs', default=''); ap.add_argument('--group', action='store_true'); ap.add_argument('--waves', type=int, default=8)
args = ap.parse_args()
src = os.path.join(ROOT, 'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd', 'csrc', 'pam_block.hip')
so = '/tmp/libbb_diag_%d.so' % os.getpid()
subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-DPAM_DIAG'] +
                      ['-D' + d for d in args.defs.split(',') if d] + ['-shared', src, '-o', so])
import numpy as np, torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
lib = C.CDLL(so)
lib.pam_basic_block_nhwc_bf16_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
dev = torch.device('cuda:0')
SH = [(48, 96, 72), (96, 48, 36), (192, 24, 18)]
ops, xs, ys = [], [], []
for c, h, w in SH:
    ops.append(hrnet_hip.PackedBlock(nn.Conv2d(c, c, 3, 1, 1), nn.Conv2d(c, c, 3, 1, 1), dev))
    xs.append(torch.randn((args.n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last))
    ys.append(torch.empty_like(xs[-1]))


def run(idx):
    d = (_lib.PamBlockDesc * len(idx))()
    for k, i in enumerate(idx):
        c, h, w = SH[i]
        d[k].in_ = xs[i].data_ptr(); d[k].w_img = ops[i].w_img.data_ptr(); d[k].bias = ops[i].bias.data_ptr(); d[k].out = ys[i].data_ptr()
        d[k].N = args.n; d[k].H = h; d[k].W = w; d[k].C = c
    rc = lib.pam_basic_block_nhwc_bf16_ex(C.c_void_p(torch.cuda.current_stream().cuda_stream), len(idx), C.cast(d, C.c_void_p), args.waves)
    assert rc == 0, rc


stamps = torch.zeros((4096 * 3, 8), dtype=torch.int64, device=dev)
for idx in (([[0], [1]] if args.waves == 4 else [[0], [1], [2]]) + ([[0, 1] if args.waves == 4 else [0, 1, 2]] if args.group else [])):
    lib.pam_block_debug_stamps(None)
    for _ in range(5): run(idx)
    torch.cuda.synchronize()
    lib.pam_block_debug_stamps(C.c_void_p(stamps.data_ptr()))
    stamps.zero_()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record(); run(idx); b.record(); torch.cuda.synchronize()
    sall = stamps.cpu().numpy()
    s = sall[:4096]
    s = s[s[:, 0] != 0]
    pr = sall[4096:]
    t0 = s[:, 0].min()
    names = ['X load+store', 'conv1', 'resid+mid', 'conv2', 'epilogue']
    print('branches %s: %d items, launch %.1f us (event), first start -> last end %.0f cycles (100 MHz memtime units? see ratio)' % (
        [SH[i][0] for i in idx], len(s), a.elapsed_time(b) * 1e3, s[:, 5].max() - t0))
    for k, nm in enumerate(names):
        d = s[:, k + 1] - s[:, k]
        print('   %-14s median %7.0f  p10 %7.0f  p90 %7.0f' % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    if 'BB_PROBE' in args.defs:
        for half, nm in ((0, 'wave 0 (A)'), (1, 'wave 4 (B)')):
            q = pr[half::2][:len(s)]
            q = q[q[:, 0] != 0]
            if len(q):
                d = np.median(q[:, 1:] - q[:, :1], axis=0)
                print('   probe %s: before-barrier %+d  after-barrier %+d  P3 %+d  P4 %+d  P5 %+d   | A vs B after-barrier skew %d' % (
                    nm, d[0], d[1], d[2], d[3], d[4], np.median(pr[1::2][:len(q), 2] - pr[0::2][:len(q), 2])))
    tot = s[:, 5] - s[:, 0]
    print('   %-14s median %7.0f  p10 %7.0f  p90 %7.0f ; start spread %.0f' % ('item', np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90), s[:, 0].max() - t0))
os.remove(so)
