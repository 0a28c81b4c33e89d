#!/usr/bin/env python3
"""GPU timing of the YOLOv3 person detector (resize -> Darknet-53 -> decode + NMS) on n views (development tool)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import yolov3

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=5); ap.add_argument('--h', type=int, default=776); ap.add_argument('--w', type=int, default=1032)
ap.add_argument('--iters', type=int, default=20); ap.add_argument('--no-graph', action='store_true')
args = ap.parse_args()
det = yolov3.YOLOv3(None, None, None, score_thresh=0.5, nms_thresh=0.4, use_graph=not args.no_graph)
frames = torch.randint(0, 256, (args.n, args.h, args.w, 3), dtype=torch.uint8, device='cuda:0')
det.net.count = dict(bytes=0, flops=0, launches=0)
det.use_graph, g = False, det.use_graph
det.detect_dev(frames); torch.cuda.synchronize()
work = dict(det.net.count); det.net.count = None
det.use_graph = g; det._graphs.clear()
buf = det.frame_buffer(args.n, args.h, args.w); buf.copy_(frames)
det.detect_dev(buf); buf = det.frame_buffer(args.n, args.h, args.w); buf.copy_(frames)
for _ in range(3): det.detect_dev(buf)
torch.cuda.synchronize()
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record()
for _ in range(args.iters): boxes, count = det.detect_dev(buf)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / args.iters
print('views=%d %dx%d  %.3f ms/frame-set  %.1f TFLOP/s  %.2f TB/s algorithmic  launches=%d  boxes=%s' % (
    args.n, args.w, args.h, ms, work['flops'] / ms / 1e9, work['bytes'] / ms / 1e9, work['launches'] + 2, count[:args.n].tolist()))
