#!/bin/bash
# exponentiation-bound kernels (batched inverse, square root, Legendre, pow) after the round-3 changes: timings and SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/alu_r03; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
python $R/tools/inv_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/plain.log
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  PROBE_P61_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -o pmc -- python $R/tools/inv_probe.py > $O/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python - <<PY | tee $O/pmc_summary.txt
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/pmc*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if 'k_inv_batch' in name or 'k_pow' in name:
            acc[name + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
