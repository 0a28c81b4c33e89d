#!/bin/bash
# kernel trace of the captured forward in one executor variant (GPU box, through gpurun): tools/fwd_trace.sh <tag> <spec> [n] [t0_us t1_us]
TAG=$1; SPEC=$2; N=${3:-20}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fwd_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/fwd_run.py --n $N "$SPEC" > $OUT/run.log 2>&1 )
CSV=$(ls $OUT/t/*/*kernel_trace.csv | head -1)
cp $CSV $OUT/trace.csv; rm -rf $OUT/t
tail -1 $OUT/run.log
python3 $R/tools/fwd_trace.py $OUT/trace.csv $4 $5
