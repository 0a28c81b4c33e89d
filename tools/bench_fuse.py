#!/usr/bin/env python3
"""Per-launch times of the fuse-layer kernels at N crops: every 3x3 stride-2 layer shape of HRNet-W48 on the generic kernels vs k_down48
(optionally a sweep over its tiles), and every fuse-layer sum as separate launches (1x1 convolutions + k_upsample_add) vs k_fuse_sum.
usage: bench_fuse.py [--n 20] [--sweep] [--only down|sum]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=20); ap.add_argument('--iters', type=int, default=30); ap.add_argument('--sweep', action='store_true')
ap.add_argument('--only', default='')
a = ap.parse_args()
dev = torch.device('cuda:0')
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev; e.tile_cfg = -1


def timeit(fn, iters=a.iters):
    """us per launch: `iters` launches captured into ONE hipGraph and replayed (an eager Python loop is host-bound at ~15 us per launch)"""
    e._keep = []
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    keep.append((g, e._keep))
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


keep = []
cl = lambda c, h, w: torch.randn((a.n, c, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
if a.only in ('', 'down'):
    # (h, w, cin, cout, channels of the tensor the input is a slice of, relu_from, launches per forward)
    for h, w, cin, cout, wide, rf, cnt in ((96, 72, 48, 96, 48, 0, 1), (96, 72, 48, 144, 48, 96, 4), (96, 72, 48, 192, 48, 96, 2), (48, 36, 48, 192, 144, 0, 4),
                                           (48, 36, 48, 192, 192, 0, 2), (48, 36, 48, 48, 192, 0, 2), (24, 18, 48, 384, 48, 0, 2),
                                           (48, 36, 96, 192, 96, 0, 5), (48, 36, 96, 288, 96, 192, 2), (24, 18, 96, 384, 288, 0, 2), (24, 18, 192, 384, 192, 0, 3)):
        op = hrnet_hip.PackedConv(nn.Conv2d(cin, cout, 3, 2, 1), dev)
        x = cl(wide, h, w)[:, :cin]
        e.down48 = e.down_s = False
        t0 = timeit(lambda: e.conv(op, x, relu=True, relu_from=rf))
        if cin != 48 and e.lib.pam_conv3x3s2_slab(h, w, cin, cout):
            e.down_s = True
            t1 = timeit(lambda: e.conv(op, x, relu=True, relu_from=rf))
            print('%3dx%-3d %3d->%-3d (of %3d) x%d  generic %6.1f us   k_down_s %6.1f us (slab %d)' % (h, w, cin, cout, wide, cnt, t0, t1, e.lib.pam_conv3x3s2_slab(h, w, cin, cout)), flush=True)
            continue
        line = '%3dx%-3d %3d->%-3d (of %3d) x%d  generic %6.1f us' % (h, w, cin, cout, wide, cnt, t0)
        if cin == 48:
            e.down48 = True
            t1 = timeit(lambda: e.conv(op, x, relu=True, relu_from=rf))
            import ctypes as C
            t3 = (C.c_int32 * 3)(); e.lib.pam_conv3x3s2_c48_tile(a.n, h, w, cout, t3)
            line += '   k_down48 %6.1f us (tile %d x %d, %d groups)' % (t1, t3[0], t3[1], t3[2])
            if a.sweep:
                ho, wo, ns = h // 2, w // 2, cout // 48
                res = []
                for tr in range(1, ho + 1):
                    for tc in sorted({wo, (wo + 1) // 2, (wo + 2) // 3, (wo + 3) // 4}):
                        if (2 * tr + 1) * (2 * tc + 1) > 789 or (tr * tc + 15) // 16 > 24:
                            continue
                        for g in [q for q in range(1, ns + 1) if ns % q == 0]:
                            e.d48_tile = (tr, tc, g)
                            res.append((timeit(lambda: e.conv(op, x, relu=True, relu_from=rf), 10), tr, tc, g))
                e.d48_tile = None
                res.sort()
                line += '   best: ' + ', '.join('%.1f us @ %dx%d/%d' % r for r in res[:5])
        print(line, flush=True)
if a.only in ('', 'sum'):
    # (h, w, c, shifts, plain terms, launches per forward)
    for h, w, c, shifts, npl, cnt in ((96, 72, 48, (1,), 0, 1), (96, 72, 48, (1, 2), 0, 4), (96, 72, 48, (1, 2, 3), 0, 3), (48, 36, 96, (1,), 1, 4),
                                      (48, 36, 96, (1, 2), 1, 2), (24, 18, 192, (1,), 2, 2)):
        convs = [nn.Conv2d(c << sh, c, 1) for sh in shifts]
        op = hrnet_hip.PackedUp(convs, shifts, dev)
        packed = [hrnet_hip.PackedConv(cv, dev) for cv in convs]
        base, plain, srcs = cl(c, h, w), [cl(c, h, w) for _ in range(npl)], [cl(c << sh, h >> sh, w >> sh) for sh in shifts]
        tc_ = [timeit(lambda pk=pk, s=s: e.conv(pk, s)) for pk, s in zip(packed, srcs)]
        terms = [e.conv(pk, s) for pk, s in zip(packed, srcs)]
        tu = timeit(lambda: e.upsample_add(base, plain + terms, [0] * npl + list(shifts), relu=True))
        tf = timeit(lambda: e.fuse_sum(op, base, plain, srcs))
        line = '%3dx%-3d C=%-3d up %s plain %d x%d   1x1 %s + k_upsample_add %5.1f = %5.1f us   k_fuse_sum %5.1f us' % (
            h, w, c, shifts, npl, cnt, ' + '.join('%.1f' % t for t in tc_), tu, sum(tc_) + tu, tf)
        if a.sweep:
            res = []
            for ta, tb in ((1, 3), (2, 3), (4, 3), (4, 6), (8, 12), (2, 9), (4, 9), (8, 6)):
                try:
                    res.append((timeit(lambda: e.fuse_sum(op, base, plain, srcs, tile=(ta, tb)), 10), ta, tb))
                except Exception:
                    pass
            res.sort()
            line += '   best: ' + ', '.join('%.1f us @ %dx%d' % r for r in res[:4])
            line += '   capped: ' + ', '.join('%d wg %.1f us' % (q, timeit(lambda: e.fuse_sum(op, base, plain, srcs, max_wg=q), 10)) for q in (64, 128, 192))
        print(line, flush=True)
