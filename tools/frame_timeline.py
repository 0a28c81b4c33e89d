#!/usr/bin/env python3
"""Where a frame's time goes outside the conv stack (development tool).  Input: the *_kernel_trace.csv of
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batched
            --no-families --no-surface --no-h2d --no-drift --no-pair
Frames are delimited by k_preprocess_crops; for the steady-state frames prints the median of: preprocess, gap to the first conv-stack
kernel, conv-stack span, gap to the head kernel, head + finish, the rest up to the next preprocess, chip-idle time per frame, and
the span of k_frame relative to its frame."""
import csv, sys
import numpy as np
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort(key=lambda r: r[1])
pre = [i for i, r in enumerate(rows) if 'k_preprocess_crops' in r[0]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
frames = list(zip(pre[skip:-1], pre[skip + 1:]))
is_conv = lambda n: any(s in n for s in ('k_conv', 'k_bblock', 'k_upsample', 'k_pw', 'k_bneck', 'k_stem_fused', 'k_down', 'k_fuse_sum', 'k_flag'))
acc = {}
def add(k, v): acc.setdefault(k, []).append(v / 1e3)
for a, b in frames:
    ks = rows[a:b]
    t0, t1 = ks[0][1], rows[b][1]
    conv = [k for k in ks if is_conv(k[0])]
    head = [k for k in ks if 'k_head_argmax' in k[0] or 'k_argmax_finish' in k[0]]
    frm = [k for k in ks if 'k_frame' in k[0]]
    if not conv or not head:
        continue
    c0, c1 = conv[0][1], max(k[2] for k in conv)
    h0, h1 = head[0][1], max(k[2] for k in head)
    add('frame', t1 - t0); add('preprocess', ks[0][2] - ks[0][1]); add('gap pre->conv', c0 - ks[0][2]); add('conv span', c1 - c0)
    add('gap conv->head', h0 - c1); add('head+finish', h1 - h0); add('head end -> next preprocess', t1 - h1)
    ev = sorted([(k[1], 1) for k in ks] + [(min(k[2], t1), -1) for k in ks])
    cur, last, idle = 0, t0, 0
    for t, d in ev:
        if cur == 0: idle += t - last
        last = t; cur += d
    add('chip idle', idle + max(0, t1 - last if cur == 0 else 0))
    if frm:
        add('k_frame start after frame start', frm[0][1] - t0); add('k_frame', frm[0][2] - frm[0][1])
    others = [k for k in ks if not is_conv(k[0]) and k not in head and k not in frm and k is not ks[0]]
    add('other kernels (count)', len(others) * 1e3); add('other kernels (sum us)', sum(k[2] - k[1] for k in others))
print('%d frames' % len(acc.get('frame', [])))
for k, v in acc.items():
    print('  %-36s median %8.1f   min %8.1f   max %8.1f' % (k, np.median(v), np.min(v), np.max(v)))
good = [(a, b) for a, b in frames if any('k_frame' in k[0] for k in rows[a:b]) and any(is_conv(k[0]) for k in rows[a:b])]
a, b = good[len(good) // 2]
qs = {}
for k in rows[a:b]:
    if is_conv(k[0]): qs[k[3]] = qs.get(k[3], 0) + 1
print('a steady-state frame, kernels outside the conv stack (conv-stack kernels per queue: %s):' % qs)
t0 = rows[a][1]
for k in rows[a:b + 1]:
    if not is_conv(k[0]):
        print('  %8.1f +%6.1f q%s s%s %s' % ((k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[3], k[4], k[0][:70]))
