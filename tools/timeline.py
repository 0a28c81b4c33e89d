#!/usr/bin/env python3
"""Timeline analysis of one HRNet graph replay from a rocprofv3 --kernel-trace database (development tool):
wall time, chip-idle gaps, concurrency histogram and per-phase breakdown."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id, grid_x/workgroup_x, grid_y from kernels order by start"))
# the last replay = the last 315 conv/upsample kernels
ks = [r for r in rows if r[0].startswith('void k_conv') or r[0].startswith('k_upsample') or 'k_bblock' in r[0] or 'k_fuse' in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 315
ks = ks[-n:]
t0 = ks[0][1]; t1 = max(k[2] for k in ks)
print('kernels %d  wall %.1f us  sum of durations %.1f us' % (len(ks), (t1 - t0) / 1e3, sum(k[2] - k[1] for k in ks) / 1e3))
ev = sorted([(k[1], 1) for k in ks] + [(k[2], -1) for k in ks])
hist = {}; cur = 0; last = t0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
for c in sorted(hist): print('  concurrency %d: %.1f us (%.1f%%)' % (c, hist[c] / 1e3, 100 * hist[c] / (t1 - t0)))
# per-kernel gap to previous kernel end on the same queue
byq = {}
for k in ks: byq.setdefault(k[4], []).append(k)
for q, lst in byq.items():
    gaps = [b[1] - a[2] for a, b in zip(lst, lst[1:])]
    print('  queue %s: %d kernels, busy %.1f us, median gap %.2f us, sum gaps %.1f us' % (q, len(lst), sum(k[2] - k[1] for k in lst) / 1e3, np.median(gaps) / 1e3 if gaps else 0, sum(gaps) / 1e3))
# first 40 kernels: name, start offset, duration
if len(sys.argv) > 3:
    for k in ks[:int(sys.argv[3])]:
        print('  %8.1f +%6.1f q%s %s grid %dx%d' % ((k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[4], k[0][:40], k[5], k[6]))
