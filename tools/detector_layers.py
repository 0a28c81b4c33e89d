#!/usr/bin/env python3
"""Per-launch times of the YOLOv3 detector's conv stack (Darknet-53 on the MFMA kernels), every distinct launch alone on the chip
(20 back-to-back repetitions between HIP events), largest first (development tool).  usage: detector_layers.py [--n 5]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import yolov3
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=5)
args = ap.parse_args()
det = yolov3.YOLOv3(None, None, None, score_thresh=0.5, nms_thresh=0.4, use_graph=False)
H, W = det.size
x8 = torch.randn((args.n, 8, H, W), device='cuda:0').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
net = det.net
net.forward(x8); torch.cuda.synchronize()
net.prof = []
net.forward(x8); torch.cuda.synchronize()
rec, net.prof = net.prof, None
t = {}
for r in rec:
    if r['sig'] in t:
        continue
    r['fn'](); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        r['fn']()
    b.record(); torch.cuda.synchronize()
    t[r['sig']] = a.elapsed_time(b) / 20.0 * 1e3
rows = {}
for r in rec:
    k = (r['family'], r['sig'])
    e = rows.setdefault(k, dict(n=0, us=t[r['sig']], bytes=r['bytes'], flops=r['flops']))
    e['n'] += 1
tot = sum(e['n'] * e['us'] for e in rows.values())
print('launches %d, sum of stand-alone times %.1f us' % (len(rec), tot))
for (fam, sig), e in sorted(rows.items(), key=lambda kv: -kv[1]['n'] * kv[1]['us']):
    print('%-46s x%2d  %6.1f us each  %6.1f total  %5.2f TB/s  %6.1f TFLOP/s  %s' % (fam[:46], e['n'], e['us'], e['n'] * e['us'], e['bytes'] / e['us'] / 1e6,
                                                                                 e['flops'] / e['us'] / 1e6, str(sig)[:60]))
