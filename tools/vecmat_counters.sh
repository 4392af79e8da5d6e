#!/bin/bash
# SQ counters of the few-rows product kernel (M = 8 and M = 1, 4096 x 4096 over 2^61 - 1): three passes of four counters.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  (cd /tmp && VECMAT_SHAPES=${VECMAT_SHAPES:-8x4096x4096,1x4096x4096} timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/ctr_vecmat/p$i -o pmc -- python $R/tools/vecmat_probe.py) > $O/ctr_vecmat_p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/ctr_vecmat/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if 'vecmat' in name:
            acc[name + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
