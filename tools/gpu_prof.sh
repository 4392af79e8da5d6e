#!/bin/bash
# rocprofv3: kernel trace + stats (csv) of the bench, then PMC passes (separate runs, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=${TAG:-r01}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/rocprof_$T.log 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${T}_$C -o $C -- python $R/tools/pmc_probe.py > $O/pmc_${T}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
find $O/prof_$T $O/pmc_${T}_* -type f | head -40
