#!/bin/bash
# rocprofv3 PMC passes over one conv layer (run on the GPU box through gpurun). usage: tools/pmc_conv.sh <shape> <tag>
SHAPE=${1:-96,72,48,48,3,1}; TAG=${2:-c48}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
P5="FETCH_SIZE"
P6="WRITE_SIZE GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_one.py --shape $SHAPE --iters 6 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
tot=collections.OrderedDict()
for f in sorted(glob.glob('$OUT/p*/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        if 'k_conv' not in r['Kernel_Name']: continue
        tot.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k,v in tot.items():
    v=v[1:] if len(v)>1 else v
    print('%-28s per-launch %.4g'%(k, sum(v)/len(v)))
PY
