#!/usr/bin/env python3
"""Markdown table over the per-configuration bench lines of a round (profiles/README.md): frames/s, conv-stack fraction of the bf16 MFMA
peak, k_frame time, full-pipeline rate, CPU baseline.  usage: profiles_table.py <dir> <tag>"""
import json, os, sys
d, tag = sys.argv[1], sys.argv[2]
pm = {}
try:
    for c in json.load(open(os.path.join(d, '%s_pmc_k_frame.json' % tag)))['cases']:
        if c.get('scenes') == 1:
            pm[c.get('workload')] = c
except Exception:
    pass
print('| BASELINE config | workload | crops / frame | frames/s (boxes given) | ms / frame | conv stack: ms, fraction of 2.5 PFLOP/s | full pipeline frames/s (detector in the loop) | `k_frame` us (one scene), counter HBM bytes vs algorithmic | CPU baseline frames/s (cores) |')
print('|---|---|---|---|---|---|---|---|---|')
for cfgno, wl in (('#2', 'S1'), ('#3', 'S2'), ('#4', 'S3'), ('#5 (single-GPU leg)', 'S4')):
    try:
        j = json.loads(open(os.path.join(d, '%s_bench_%s_n1.json' % (tag, wl))).read().strip().splitlines()[-1])
    except Exception as e:
        print('| %s | %s | (no line: %s) |' % (cfgno, wl, e)); continue
    r, fp, cb, k = j['roofline'], j.get('full_pipeline') or {}, j.get('cpu_baseline') or {}, pm.get(wl, {})
    print('| %s | %s | %s | %.1f | %.3f | %.3f ms, %.3f | %s | %s | %s |' % (
        cfgno, j['config']['workload'], (j['config']['crops_per_rank'] or ['?'])[0], j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'],
        ('%.1f (serial %.1f; detector alone %.2f ms = %.3f of the MFMA peak)' % (fp['value'], fp['serial']['value'], fp['detector']['ms'], fp['detector']['frac'])) if fp else '-',
        ('%.0f us, %.0f KB vs %.0f KB' % (k['us_per_launch'], k.get('hbm_bytes_per_launch', 0) / 1024, k.get('algorithmic_bytes_per_scene', 0) / 1024)) if k else '-',
        ('%.2f (%s)' % (cb['value'], cb.get('cores'))) if cb else '-'))
