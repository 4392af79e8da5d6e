#!/bin/bash
# tests + bench (+ optional rocprof) on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
(time timeout 900 python bench.py --steps 50 --warmup 5 $BENCH_ARGS) > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$PROF -o $PROF -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/rocprof_$PROF.log 2>&1
  echo "rocprof rc=$?" >> $O/rocprof_$PROF.log
fi
tail -3 $O/pytest_gpu.log; tail -4 $O/bench.log | cut -c1-400
