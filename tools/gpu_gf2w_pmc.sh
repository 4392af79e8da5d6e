#!/bin/bash
# kernel trace + SQ counters of the wide binary-field kernels (tools/gf2w_probe.py); counters in their own passes (kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03}; O=$R/gpurun_out/gf2w_$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
python $R/tools/gf2w_probe.py > $O/plain.log 2>&1; cat $O/plain.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/gf2w_probe.py > $O/trace.log 2>&1
python $R/tools/trace_summary.py $O/trace/t_kernel_trace.csv gf2w k_ew2 k_recombine | tee $O/trace_summary.txt
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -o pmc -- python $R/tools/gf2w_probe.py > $O/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python - <<PY | tee $O/pmc_summary.txt
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/pmc*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if 'gf2w' in name or 'GF2W' in name:
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
