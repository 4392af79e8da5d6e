cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== bitslice"; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bitslice_probe.hip -o /tmp/bsp 2>/dev/null && timeout 120 /tmp/bsp
echo "== inverse A/B"; timeout 600 python tools/inv_ab.py 2>&1 | grep -v amdgpu.ids
echo "== sbox interleaved"; FFGPU_SBL_BURST=0 timeout 300 python tools/sbox_layer_time.py 2>&1 | grep -v amdgpu.ids
echo "== sbox burst"; FFGPU_SBL_BURST=1 timeout 300 python tools/sbox_layer_time.py 2>&1 | grep -v amdgpu.ids
echo "== tests"; timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_protocols.py tests/test_gpu_parity.py -k "sbox or inverse or pow or inv" 2>&1 | tail -5
echo "== valu pmc"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/valu_r04 -o valu -- python $GRAFT_REPO_ROOT/tools/valu_probe.py) > $O/valu_r04.log 2>&1; tail -2 $O/valu_r04.log
F=$(find $O/valu_r04 -name '*counter_collection.csv' | head -1); echo $F
python tools/valu_summary.py $F $O/valu_order.json $O/r04_valu.json $O/r04_valu.md | tail -30
