#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binary or gf2 or golden or recomb" > gpurun_out/r03c_tests.log 2>&1; tail -4 gpurun_out/r03c_tests.log
python tools/gf2w_probe.py
GF2W_ROT=1 python tools/gf2w_probe.py
