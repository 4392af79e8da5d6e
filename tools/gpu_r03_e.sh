#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_protocols.py -m gpu -x -q > gpurun_out/r03e_tests.log 2>&1; tail -6 gpurun_out/r03e_tests.log
python tools/sbox_layer_time.py
