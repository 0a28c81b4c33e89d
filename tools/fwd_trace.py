#!/usr/bin/env python3
"""One replay of the HRNet forward from a rocprofv3 --kernel-trace CSV (tools/fwd_trace.sh): wall, CU-time = sum(workgroups x duration)
/ 256 per kernel family, concurrency histogram, and the kernels of a time window.  usage: fwd_trace.py trace.csv [t0_us t1_us]"""
import csv, sys, re
import numpy as np
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(1, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * max(1, int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * max(1, int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    rows.append(dict(name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']), q=r.get('Queue_Id', ''), wg=wg,
                     lds=int(r.get('LDS_Block_Size', 0) or 0), vgpr=int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0)))
rows.sort(key=lambda r: r['s'])
stem = [i for i, r in enumerate(rows) if 'k_stem_fused' in r['name'] or 'k_conv_stem' in r['name']]
a = stem[-2]; b = stem[-1]                       # the second-to-last replay: complete
ks = rows[a:b]
t0 = ks[0]['s']; t1 = max(k['e'] for k in ks)
def fam(n):
    m = re.match(r'(?:void )?(?:\(anonymous namespace\)::)?(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:40]
print('%d kernels, wall %.1f us, sum of durations %.1f us' % (len(ks), (t1 - t0) / 1e3, sum(k['e'] - k['s'] for k in ks) / 1e3))
byf = {}
for k in ks:
    f = byf.setdefault(fam(k['name']), [0, 0.0, 0.0, k['lds'], k['vgpr']])
    f[0] += 1; f[1] += (k['e'] - k['s']) / 1e3; f[2] += min(k['wg'], 256) * (k['e'] - k['s']) / 1e3 / 256
print('%-44s %5s %9s %9s %7s %5s' % ('family', 'n', 'sum us', 'CU-time', 'LDS', 'VGPR'))
for f, v in sorted(byf.items(), key=lambda t: -t[1][2]):
    print('%-44s %5d %9.1f %9.1f %7d %5d' % (f[:44], v[0], v[1], v[2], v[3], v[4]))
print('%-44s %5d %9.1f %9.1f' % ('total', len(ks), sum(v[1] for v in byf.values()), sum(v[2] for v in byf.values())))
ev = sorted([(k['s'], 1) for k in ks] + [(k['e'], -1) for k in ks])
hist = {}; cur = 0; last = t0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
print('concurrency: ' + '  '.join('%d: %.0f us' % (c, hist[c] / 1e3) for c in sorted(hist)))
if len(sys.argv) > 3:
    w0, w1 = float(sys.argv[2]) * 1e3 + t0, float(sys.argv[3]) * 1e3 + t0
    for k in ks:
        if k['e'] >= w0 and k['s'] <= w1:
            print('  %8.1f +%6.1f q%-3s wg %5d lds %6d  %s' % ((k['s'] - t0) / 1e3, (k['e'] - k['s']) / 1e3, k['q'], k['wg'], k['lds'], fam(k['name'])[:60]))
