#!/bin/bash
# GPU box: new round-3 tests, the full bench line (timed), and bench --gpus 2 on one GPU (gloo, validation of the N>1 control flow)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo skip-tests
tail -5 gpurun_out/r03b_tests.log
echo skip
tail -3 gpurun_out/r03b_tests.log
SECONDS=0; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
echo "bench wall: ${SECONDS}s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03b_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('roofline'))
print(json.dumps(d.get('cpu_baseline'), indent=1))
print(json.dumps(d.get('api'), indent=1))
print(d.get('extras_error'))
print(json.dumps(d.get('multi_gpu'), indent=1)[:3000])
PY
FFGPU_BENCH_DEVICE=0 FFGPU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --n 2000000 --no-extras --no-cpu-baseline > gpurun_out/r03b_bench2.json 2> gpurun_out/r03b_bench2.err
echo "bench2 rc=$?"; tail -c 3000 gpurun_out/r03b_bench2.json; tail -5 gpurun_out/r03b_bench2.err
