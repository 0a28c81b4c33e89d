#!/bin/bash
# Round-end artefacts (GPU box, through gpurun): bench line, the same command under rocprofv3 --kernel-trace --stats, PMC passes
# (whole-forward HBM traffic, per-family counters), the N = 2 bench on one device.  usage: tools/refresh_profiles.sh <git commit> <tag> [light]
# light: only the bench line, the same command under rocprofv3 and the N = 2 run (kernels unchanged since the last PMC passes)
COMMIT=${1:-unknown}; TAG=${2:-r05}; MODE=${3:-full}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd $R
# the executor configuration the autotuner picks for 20 crops on THIS box (a short bench run), then the whole-forward HBM counters in it
CONFIG=$(python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-families --no-surface --no-batched --no-drift --no-pair --no-h2d 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['config']['conv_executor']['20']['choice'])")
echo "executor configuration for 20 crops: $CONFIG"
if [ $MODE = full ]; then
bash tools/pmc_hrnet.sh $COMMIT $TAG ${CONFIG:-fused48_fused96} > $OUT/pmc_hrnet.log 2>&1
cp gpurun_out/pmc_hrnet/${TAG}_hrnet_hbm_traffic.json profiles/ 2>/dev/null
fi       # bench.py stamps roofline.traffic from the newest one
python3 bench.py > $OUT/${TAG}_bench_S2_n1.json 2> $OUT/bench.err
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 30 --warmup 5 > $OUT/${TAG}_bench_S2_under_rocprof.json 2> $OUT/rocprof.err )
cp $(ls -t $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_S2_kernel_stats.csv
python3 bench.py --gpus 2 --steps 20 --warmup 4 > $OUT/${TAG}_bench_n2_one_device.json 2> $OUT/bench2.err
# one bench line per BASELINE.json configuration besides S2 (N = 1): Campus-like S1, Panoptic-like 5 HD cams S3, Panoptic 31 cams S4 (single-GPU leg)
for WL in S1 S3 S4; do
  python3 bench.py --workload $WL --steps 40 --warmup 6 --no-pair --no-h2d --no-drift > $OUT/${TAG}_bench_${WL}_n1.json 2> $OUT/bench_$WL.err
done
if [ $MODE = full ]; then
bash tools/pmc_families.sh $COMMIT $TAG > $OUT/pmc_families.log 2>&1
bash tools/pmc_frame.sh $COMMIT $TAG > $OUT/pmc_frame.log 2>&1
fi
cp gpurun_out/pmc_fam/${TAG}_pmc_families.json gpurun_out/pmc_hrnet/${TAG}_hrnet_hbm_traffic.json gpurun_out/pmc_frame/${TAG}_pmc_k_frame.json $OUT/ 2>/dev/null
rm -rf $OUT/stats
python3 tools/profiles_table.py $OUT $TAG > $OUT/${TAG}_configs_table.md 2>/dev/null
ls -la $OUT
