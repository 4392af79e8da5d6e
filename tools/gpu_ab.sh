#!/bin/bash
# A/B of launch geometry and cache policy inside the real bench, plus tuning probe and tests.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 ./build/tune_stream > $O/tune2.log 2>&1
(timeout 600 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for cfg in "NT1_BPC0:FFGPU_NT=1 FFGPU_BLOCKS_PER_CU=0" "NT0_BPC0:FFGPU_NT=0 FFGPU_BLOCKS_PER_CU=0" "NT1_BPC8:FFGPU_NT=1 FFGPU_BLOCKS_PER_CU=8" "NT0_BPC8:FFGPU_NT=0 FFGPU_BLOCKS_PER_CU=8"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_$name.log 2>&1
done
tail -2 $O/pytest_gpu.log
