cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_parity.py -k "bitsliced or full_batch or wide_binary or pow_and_inverse" 2>&1 | tail -5
echo "== gf2w"; timeout 200 python tools/gf2w_probe.py 2>&1 | grep -v amdgpu.ids | grep mul
FFGPU_GF2W_BITSLICED=0 timeout 200 python tools/gf2w_probe.py 2>&1 | grep -v amdgpu.ids | grep "gf2_64 mul"
