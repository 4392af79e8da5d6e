#!/bin/bash
# GPU box: API-level parity tests + exploratory timing of the API path (cProfile of party 0).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD/tests:$PWD:$PWD/_refstage
timeout 1500 python -m pytest tests/test_api_path.py -m gpu -x -q > gpurun_out/api_tests.log 2>&1
tail -5 gpurun_out/api_tests.log
for cfg in "1 10000000 5" "1 100000000 3" "3 10000000 3" "3 1000000 3"; do
  set -- $cfg
  M=$1; N=$2; R=$3
  ARGS="--no-log"; [ "$M" -gt 1 ] && ARGS="$ARGS -M$M"
  API_MODE=gpu API_N=$N API_REPS=$R API_CPROFILE=gpurun_out/api_m${M}_n${N}.prof timeout 900 python tests/api_program.py $ARGS 2>&1 | grep API_RESULT | tee gpurun_out/api_m${M}_n${N}.log
  python - <<PY
import pstats
p = pstats.Stats('gpurun_out/api_m${M}_n${N}.prof')
p.sort_stats('cumulative').print_stats(45)
PY
done > gpurun_out/api_profile.txt 2>&1
grep API_RESULT gpurun_out/api_profile.txt
