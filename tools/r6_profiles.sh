#!/bin/bash
# Round-6 artefacts (GPU box, through gpurun), every step under a hard time limit; rocprofv3 steps LAST in their call (on this pool a
# process started after a rocprofv3 run has been seen to hang).  usage: tools/r6_profiles.sh <git commit> <step>
#   step bench   : the bench line of every BASELINE configuration (S2 full; S1 / S3 / S4 without the extra legs) + N = 2 on one device
#   step trace   : bench.py under rocprofv3 --kernel-trace --stats -> kernel stats of the same command
#   step kframe  : tools/pmc_frame.sh (PMC passes over k_frame, S1-S4 + 2 048 S2 scenes)
#   step memory  : replay cache + arena of prewarmed S2 and S4 pipelines
COMMIT=${1:-unknown}; STEP=${2:-bench}; TAG=r06
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
T="timeout -s KILL"
case $STEP in
bench)
  $T 400 python3 bench.py > $OUT/${TAG}_bench_S2_n1.json 2> $OUT/bench.err; echo "S2 rc=$?"
  for WL in S1 S3 S4; do
    $T 500 python3 bench.py --workload $WL --steps 40 --warmup 6 --no-pair --no-h2d --no-drift --no-driver-loop > $OUT/${TAG}_bench_${WL}_n1.json 2> $OUT/bench_$WL.err; echo "$WL rc=$?"
  done
  $T 500 python3 bench.py --gpus 2 --steps 20 --warmup 4 > $OUT/${TAG}_bench_n2_one_device.json 2> $OUT/bench2.err; echo "n2 rc=$?"
  for f in $OUT/${TAG}_bench_S*_n1.json; do python3 -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], json.dumps(j['summary'])[:900])"; done
  ;;
trace)
  cd /tmp; export TMPDIR=/tmp
  $T 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-driver-loop --no-ab > $OUT/${TAG}_bench_S2_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
  cp $(ls -t $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_S2_kernel_stats.csv; rm -rf $OUT/stats
  head -12 $OUT/${TAG}_bench_S2_kernel_stats.csv | cut -c1-160
  ;;
kframe)
  $T 1200 bash tools/pmc_frame.sh $COMMIT $TAG > $OUT/pmc_frame.log 2>&1; echo "pmc_frame rc=$?"
  cp gpurun_out/pmc_frame/${TAG}_pmc_k_frame.json $OUT/ 2>/dev/null; tail -3 $OUT/pmc_frame.log | cut -c1-600
  ;;
memory)
  $T 200 python3 tests/flag_child.py memory 2>/dev/null | grep MEMORY-OK | tee $OUT/${TAG}_replay_cache_memory.txt
  $T 900 python3 tests/flag_child.py memory:S4 2>/dev/null | grep MEMORY-OK | tee -a $OUT/${TAG}_replay_cache_memory.txt
  ;;
esac
ls $OUT | head -40
