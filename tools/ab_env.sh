#!/bin/bash
# alternate two settings of an environment variable over bench.py runs on ONE box: tools/ab_env.sh VAR A B "<bench args>" [rounds]
VAR=$1; A=$2; B=$3; ARGS=$4; R=${5:-2}
for r in $(seq 1 $R); do for v in $A $B; do
env $VAR=$v python3 bench.py $ARGS 2>/tmp/err.txt | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); fp=j.get('full_pipeline') or {}
print('[$VAR=$v] value %.1f ms %.3f conv %.3f | full %.1f | cfg %s | tracks %s' % (j['value'], j['ms_per_step'], j['conv_stack_ms']['median'], j.get('value_full_pipeline') or 0, sorted(set(c['choice'] for c in j['config']['conv_executor'].values())), j.get('final_tracks')))" || tail -5 /tmp/err.txt
done; done
