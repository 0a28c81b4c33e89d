#!/bin/bash
# HBM traffic of one HRNet forward (20 crops) from rocprofv3 PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_hrnet; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/$P -- python3 $GRAFT_REPO_ROOT/tools/bench_hrnet.py --backends hip --modes eager --iters 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json
res={}
for P in ('FETCH_SIZE','WRITE_SIZE'):
    f=sorted(glob.glob('$OUT/%s/*/*counter_collection.csv'%P))[-1]
    rows=[r for r in csv.DictReader(open(f)) if r['Counter_Name']==P]
    conv=[r for r in rows if 'k_conv' in r['Kernel_Name'] or 'k_upsample' in r['Kernel_Name']]
    # forwards in this run: 3 warm-up + 1 + 2 timed ... count via the stem conv (grid of conv1: 8->64) occurrences
    nfw=sum(1 for r in conv if 'k_conv_igemm<4' in r['Kernel_Name'] and int(r['Grid_Size_X'])==552960 and int(r['Workgroup_Size_X'])==256)
    nfw=max(nfw,1)
    tot=sum(float(r['Counter_Value']) for r in conv)
    res[P]={'sum_kb':tot,'forwards':nfw,'kb_per_forward':tot/nfw}
# guide: FETCH_SIZE (KB) reads exactly half of a wide coalesced stream on gfx950 -> double it; WRITE_SIZE is exact
res['hbm_bytes_per_forward']=(2*res['FETCH_SIZE']['kb_per_forward']+res['WRITE_SIZE']['kb_per_forward'])*1024
res['note']='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over k_conv*/k_upsample_add kernels of one 20-crop HRNet-W48 forward; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section)'
print(json.dumps(res))
open('$OUT/traffic.json','w').write(json.dumps(res))
PY
