#!/bin/bash
# exponentiation-bound kernels after the round-3 Mersenne product / addition-chain / shared-exponentiation changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "pow_and_inverse or elementwise or sqrt or fullsize" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_api.py -q -x -m gpu 2>&1 | tail -3
python tools/inv_probe.py
} > gpurun_out/alu.log 2>&1
tail -30 gpurun_out/alu.log
