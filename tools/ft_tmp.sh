cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/crash
for i in 1 2 3; do timeout 300 python3 tools/graph_destroy_stress.py 40 > gpurun_out/crash/gi_$i.txt 2>&1; echo immortal run $i rc=$?; done
timeout 900 python -m pytest tests -m gpu -x -q -k "image or pipeline or detect" 2>&1 | tail -2; echo pytest rc=${PIPESTATUS[0]}
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/crash/bench.json 2> gpurun_out/crash/bench.err; echo bench rc=$?; python3 tools/brief.py < gpurun_out/crash/bench.json
