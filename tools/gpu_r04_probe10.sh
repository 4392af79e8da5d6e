cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_protocols.py -k "sbox or aes or layer or graph" 2>&1 | tail -5
echo "== sbox continuous"; timeout 300 python tools/sbox_clock_probe.py 2>&1 | grep -v amdgpu.ids
echo "== sbox time"; timeout 300 python tools/sbox_layer_time.py 2>&1 | grep -v amdgpu.ids | grep "fused=True"
