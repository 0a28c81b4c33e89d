#!/usr/bin/env python3
"""Print a few fields of bench.py's JSON line (stdin); a label may be given as argv[1]."""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else '', 'value %.1f  ms/step %.4f  conv stack %.4f ms  h2d %s  pair %s' % (
    j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j.get('value_with_h2d'), j.get('value_2frames_per_forward')))
