#!/usr/bin/env python3
"""Ingest throughput of pam.ingest.FrameLoader on synthetic Shelf-sized JPEGs (5 cameras x 1032x776)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import pam
from pam.ingest import FrameLoader
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
base = (rng.integers(0, 256, (97, 129, 3), dtype=np.uint8))
img = np.asarray(Image.fromarray(base).resize((1032, 776), Image.BILINEAR))
frames = []
for t in range(40):
    fs = []
    for c in range(5):
        p = os.path.join(tmp, 'c%d_%03d.jpg' % (c, t)); Image.fromarray(np.roll(img, t * 7 + c, axis=1)).save(p, quality=92); fs.append(p)
    frames.append(fs)
dev = None
if len(sys.argv) > 1 and sys.argv[1] == 'cuda':
    import torch; dev = torch.device('cuda:0')
frames = frames * 6                       # 240 frame sets; the first 40 (pinned staging buffers are allocated on the way) are not timed
for workers in (1, 4, 8, 16):
    ld = FrameLoader('Shelf', frames, workers=workers, depth=6, device=dev)
    n = 0
    for idx, imgs, ts in ld:
        n += 1
        if n == 40:
            if dev is not None:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
    if dev is not None:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ld.close()
    print('workers=%2d  %.1f frame sets/s (5 x 1032x776 JPEG per set, steady state over %d sets)%s' % (workers, (n - 40) / dt, n - 40, ' -> device' if dev is not None else ''), flush=True)
