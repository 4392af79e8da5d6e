#!/bin/bash
# device-side wire (MPYC_AMD_IPC_WIRE=1) for co-located parties: parity against the reference and timings at m = 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_api_path.py -q -x -m gpu 2>&1 | tail -15) | tee $O/ipc_tests.log
export PYTHONPATH=$R/tests:$R:$R/_refstage
cd /tmp
for cfg in "0 1000000 5 1" "1 1000000 10 1" "0 10000000 3 1" "1 10000000 10 1" "1 10000000 5 8"; do
  set -- $cfg
  echo "== ipc=$1 n=$2 reps=$3 chain=$4"
  MPYC_AMD_IPC_WIRE=$1 API_MODE=gpu API_N=$2 API_REPS=$3 API_WARMUP=2 API_CHAIN=$4 timeout 600 python $R/tests/api_program.py --no-log -M3 2>&1 | grep -E "API_RESULT|Error|error" | cut -c1-700
done 2>&1 | tee $O/ipc_times.log
