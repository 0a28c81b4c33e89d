# process-to-process variance of the headline on ONE box: N fresh processes of the plain timed run, value + conv stack + shader clock probes.
# usage (through gpurun): bash tools/variance.sh [N]
for i in $(seq 1 ${1:-12}); do
  python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-families --no-surface --no-batched --no-drift --no-pair --no-h2d 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=j['conv_stack_ms']; k=j['clock_mhz']
print('run $i value %.1f ms/step %.4f conv min %.3f med %.3f max %.3f clock %.0f -> %.0f MHz' % (j['value'], j['ms_per_step'], c['min'], c['median'], c['max'], k['before'], k['after']))"
done
