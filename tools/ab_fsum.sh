B="--steps 200 --warmup 20 --no-cpu-baseline --no-families --no-surface --no-batched --no-drift --no-pair --no-h2d"
for r in 1 2; do for m in 6 20; do
PAM_FSUM_MAX=$m python3 bench.py $B 2>/tmp/err.txt | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); fp=j.get('full_pipeline') or {}
print('[fsum_max=$m] value %.1f ms %.3f conv %.3f | full %.1f | cfg %s | traffic %s | tracks %s' % (j['value'], j['ms_per_step'], j['conv_stack_ms']['median'], j.get('value_full_pipeline') or 0, j['config']['conv_executor'], j['roofline'].get('traffic'), j.get('final_tracks')))" || tail -5 /tmp/err.txt
done; done
