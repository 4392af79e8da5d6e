#!/bin/bash
# First GPU contact: smoke, parity tests, bench, rocprof kernel trace.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > $O/smi.txt
nproc > $O/nproc.txt
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 50 --warmup 5 ) > $O/bench.log 2>&1
echo "bench rc=$?" >> $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r1 -o r1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/rocprof.log 2>&1
echo "rocprof rc=$?" >> $O/rocprof.log
ls -R $O/prof_r1 | head -30
tail -5 $O/smoke.log $O/pytest_gpu.log $O/bench.log
