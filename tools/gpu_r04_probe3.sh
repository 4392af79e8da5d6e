cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== bitslice"; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bitslice_probe.hip -o /tmp/bsp 2>/dev/null && timeout 120 /tmp/bsp
echo "== sbox clock"; timeout 300 python tools/sbox_clock_probe.py 2>&1 | grep -v amdgpu.ids
echo "== inverse"; INV_VARIANTS=1,0 timeout 300 python tools/inv_ab.py 2>&1 | grep -v amdgpu.ids | head -3
PROBE_P61_ONLY= timeout 300 python tools/inv_probe.py 2>&1 | grep -v amdgpu.ids
echo "== tests"; timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_protocols.py tests/test_gpu_api.py tests/test_gpu_sweep.py 2>&1 | tail -5
