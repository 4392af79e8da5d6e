cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_protocols.py tests/test_gpu_parity.py -k "sbox or bitsliced or aes or graph or layer" 2>&1 | tail -5
echo "== gf2w"; timeout 200 python tools/gf2w_probe.py 2>&1 | grep -v amdgpu.ids | grep "mul"
echo "== sbox W=2"; timeout 300 python tools/sbox_clock_probe.py 2>&1 | grep -v amdgpu.ids
echo "== sbox W=1"; FFGPU_SBL_WPT=1 timeout 300 python tools/sbox_clock_probe.py 2>&1 | grep -v amdgpu.ids
