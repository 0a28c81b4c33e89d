#!/bin/bash
# rocprofv3 PMC passes over ONE layer of every kernel family of the HRNet conv stack (20 crops), summarised to one JSON
# (profiles/rNN_pmc_families.json).  Counters in their own runs (no tracing), several passes: SQ has 8 slots, TCC 4.
# usage (GPU box, through gpurun): tools/pmc_families.sh <git commit> [round tag]
COMMIT=${1:-unknown}; TAG=${2:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fam; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="FETCH_SIZE"
P4="WRITE_SIZE GRBM_GUI_ACTIVE"
# name | bench_conv_one arguments
FAMS=(
 "k_bblock2_48_96x72|--shape 96,72,48,48,3,1 --block 1"
 "k_bblock2_96_48x36|--shape 48,36,96,96,3,1 --block 1"
 "k_conv3x3s_C96_48x36|--shape 48,36,96,96,3,1"
 "k_conv3x3s_C192_24x18|--shape 24,18,192,192,3,1"
 "k_conv3x3s_C384_12x9|--shape 12,9,384,384,3,1"
 "k_down48_3x3s2_48to192_96x72|--shape 96,72,48,192,3,2 --res 0"
 "k_down48_3x3s2_48to144_96x72|--shape 96,72,48,144,3,2 --res 0"
 "k_down48_3x3s2_48to192_48x36|--shape 48,36,48,192,3,2 --res 0"
 "k_down48_3x3s2_48to384_24x18|--shape 24,18,48,384,3,2 --res 0"
 "k_down_s_3x3s2_96to192_48x36|--shape 48,36,96,192,3,2 --res 0"
 "k_down_s_3x3s2_96to288_48x36|--shape 48,36,96,288,3,2 --res 0"
 "k_down_s_3x3s2_192to384_24x18|--shape 24,18,192,384,3,2 --res 0"
 "k_conv_gs_3x3s2_64to64_192x144|--shape 192,144,64,64,3,2 --res 0"
 "k_conv_gs_1x1_96to48_48x36|--shape 48,36,96,48,1,1 --res 0"
 "k_conv_gs64_1x1_384to336_12x9|--shape 12,9,384,336,1,1 --res 0"
 "k_bneck_3x3_tail_96x72|--shape 96,72,64,256,1,1 --tail 3"
 "k_stem_fused_384x288|--shape 96,72,8,64,3,2 --tail 4"
)
for F in "${FAMS[@]}"; do
  NAME=${F%%|*}; ARGS=${F#*|}
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/$NAME/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_one.py $ARGS --iters 6 > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, json, collections
out={'git_commit':'$COMMIT','crops':20,'how':'tools/pmc_families.sh: rocprofv3 --pmc, 4 passes per layer (SQ x2, FETCH_SIZE, WRITE_SIZE+GRBM), 6 launches each, first launch dropped, per-launch means','families':{}}
for d in sorted(glob.glob('$OUT/*/')):
    name=d.rstrip('/').split('/')[-1]
    tot=collections.OrderedDict()
    for f in sorted(glob.glob(d+'p*/*/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'k_conv' not in k and 'k_bblock' not in k and 'k_pw' not in k and 'k_bneck' not in k and 'k_stem_fused' not in k and 'k_down' not in k: continue
            tot.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    c={k:(sum(v[1:])/len(v[1:]) if len(v)>1 else v[0]) for k,v in tot.items()}
    if not c: continue
    d2=dict(c)
    # derived (guide: SQ_BUSY_CYCLES etc. count quad-cycles summed over SEs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles)
    if c.get('SQ_BUSY_CYCLES'): d2['mfma_busy_over_sq_busy']=c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/c['SQ_BUSY_CYCLES']
    if c.get('SQ_WAVE_CYCLES'):
        d2['wait_any_frac_of_wave_cycles']=c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES']
        d2['active_inst_frac_of_wave_cycles']=c.get('SQ_ACTIVE_INST_ANY',0)/c['SQ_WAVE_CYCLES']
    if c.get('SQ_LDS_IDX_ACTIVE'): d2['lds_bank_conflict_frac']=c.get('SQ_LDS_BANK_CONFLICT',0)/c['SQ_LDS_IDX_ACTIVE']
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c: d2['hbm_bytes']=(2*c['FETCH_SIZE']+c['WRITE_SIZE'])*1024
    out['families'][name]=d2
print(json.dumps(out)[:2000])
open('$OUT/${TAG}_pmc_families.json','w').write(json.dumps(out, indent=1))
PY
