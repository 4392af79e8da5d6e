#!/bin/bash
# per-kernel time + SQ counters of the 4096^3 matrix-core product (tools/mm_probe.py); separate passes, kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mm_trace -o mm -- python $R/tools/mm_probe.py > $O/mm_trace.log 2>&1
python $R/tools/trace_summary.py $O/mm_trace/mm_kernel_trace.csv k_limb k_split
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/mm_pmc$i -o pmc -- python $R/tools/mm_probe.py > $O/mm_pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
  python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for r in csv.DictReader(open('$O/mm_pmc$i/pmc_counter_collection.csv')):
        if 'k_limb_gemm' in r['Kernel_Name']:
            acc['gemm'][r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    for k, v in acc['gemm'].items():
        print(f'  {k:32s} {v / cnt[k]:.4g} per launch ({cnt[k]} launches)')
except Exception as e:
    print('no counters:', e)
PY
done
