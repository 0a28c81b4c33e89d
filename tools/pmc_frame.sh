#!/bin/bash
# rocprofv3 PMC passes over k_frame (the fused matching / triangulation / tracking kernel): one scene of every BASELINE workload (S1 Campus-like
# 3 cams, S2 Shelf-like 5 cams, S3 Panoptic-like 5 HD cams, S4 Panoptic 31 cams) and S2 x 2048 scenes.
# Counters in their own runs (no tracing), the program directly behind `--`.  -> gpurun_out/pmc_frame/<TAG>_pmc_k_frame.json
# usage (GPU box, through gpurun): tools/pmc_frame.sh <git commit> [round tag]
COMMIT=${1:-unknown}; TAG=${2:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_frame; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE GRBM_GUI_ACTIVE"
P5="TCC_HIT_sum TCC_MISS_sum"
CASES=("S1 1" "S2 1" "S3 1" "S4 1" "S2 2048")
for CASE in "${CASES[@]}"; do
  set -- $CASE; SZ=$1; SC=$2; NAME=${SZ}_x${SC}
  python3 $GRAFT_REPO_ROOT/tools/frame_one.py --size $SZ --scenes $SC > $OUT/$NAME.plain.json 2> $OUT/$NAME.plain.err
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/$NAME/p$i -- python3 $GRAFT_REPO_ROOT/tools/frame_one.py --size $SZ --scenes $SC > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, json, collections
out={'git_commit':'$COMMIT','how':'tools/pmc_frame.sh: rocprofv3 --pmc in 5 separate passes per case (SQ x2, FETCH_SIZE, WRITE_SIZE+GRBM, TCC hit/miss), program = tools/frame_one.py directly behind --; 32 launches per pass, the first 20 (track build-up) dropped, per-launch means over the last 12; launch time and phase table from an un-profiled run of the same program','cases':[]}
for name in ('S1_x1','S2_x1','S3_x1','S4_x1','S2_x2048'):
    try: plain=json.loads(open('$OUT/%s.plain.json'%name).read().strip().splitlines()[-1])
    except Exception as e: plain={'error':str(e)}
    tot=collections.OrderedDict()
    for f in sorted(glob.glob('$OUT/%s/p*/*/*counter_collection.csv'%name)):
        for r in csv.DictReader(open(f)):
            if 'k_frame' not in r['Kernel_Name']: continue
            tot.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    # a frame of a wide rig is three k_frame launches (round 6): 32 frames per pass -> per FRAME means over the last 12 frames
    c={k:(sum(v[-12*max(1,len(v)//32):])/12.0) for k,v in tot.items()}
    e=dict(plain); e['counters_per_launch']=c
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        # guide (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reads HALF of a wide coalesced stream on gfx950.  k_frame's
        # reads are 8-byte-per-lane f64 loads, an access width the guide calls uncalibrated: both readings are given
        e['hbm_bytes_per_launch_fetch_as_is']=(c['FETCH_SIZE']+c['WRITE_SIZE'])*1024
        e['hbm_bytes_per_launch']=(2*c['FETCH_SIZE']+c['WRITE_SIZE'])*1024
        if plain.get('us_per_launch'):
            e['hbm_GBs']=e['hbm_bytes_per_launch']/plain['us_per_launch']/1e3
            e['traffic_over_algorithmic']=e['hbm_bytes_per_launch']/max(1,plain.get('algorithmic_bytes_per_scene',0)*plain.get('scenes',1))
    if c.get('SQ_WAVE_CYCLES'):
        e['wait_any_frac_of_wave_cycles']=c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES']
        e['active_inst_frac_of_wave_cycles']=c.get('SQ_ACTIVE_INST_ANY',0)/c['SQ_WAVE_CYCLES']
        e['valu_active_frac_of_wave_cycles']=c.get('SQ_ACTIVE_INST_VALU',0)/c['SQ_WAVE_CYCLES']
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_WAVE_CYCLES'):
        e['mean_resident_waves_per_busy_cycle']=c['SQ_WAVE_CYCLES']/c['SQ_BUSY_CYCLES']
    if c.get('SQ_LDS_IDX_ACTIVE'): e['lds_bank_conflict_frac']=c.get('SQ_LDS_BANK_CONFLICT',0)/c['SQ_LDS_IDX_ACTIVE']
    if c.get('TCC_HIT_sum') is not None and (c.get('TCC_HIT_sum',0)+c.get('TCC_MISS_sum',0))>0:
        e['l2_hit_rate']=c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum'])
    out['cases'].append(e)
print(json.dumps(out)[:1500])
open('$OUT/${TAG}_pmc_k_frame.json','w').write(json.dumps(out, indent=1))
PY
