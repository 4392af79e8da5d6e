cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest -m gpu -x -q tests/test_gpu_parity.py -k "matmul or matrix_core or skinny or full_batch" 2>&1 | tail -8
echo "== mm A/B"; timeout 300 python tools/mm_ab.py 2>&1 | grep -v amdgpu.ids
FFGPU_MM_GLDS=0 timeout 300 python tools/mm_ab.py 2>&1 | grep -v amdgpu.ids
