#!/bin/bash
# GPU box: the integer algebra of HostView under the real runtime: full -m gpu suite, then np_bnnmnist -b 32 (the demo whose
# host conversions round 2 profiled) with MPYC_AMD_LAZY_INTS=0 (round-2 behaviour) and =1 (default), cProfile of the latter
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu_lazy.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_lazy.log
tail -5 $O/pytest_gpu_lazy.log
export PYTHONPATH=$R/mpyc_amd/autoinstall:$R:$R/_refstage MPYC_GPU=1
cd $R/_refstage/demos
for L in 0 1; do
  for run in 1 2; do
    t0=$(date +%s.%N); MPYC_AMD_LAZY_INTS=$L python np_bnnmnist.py -d0 -o 1234 -b 32 --no-log 2>&1 | grep -v amdgpu.ids | md5sum; t1=$(date +%s.%N)
    python -c "print('np_bnnmnist -b 32  LAZY_INTS=$L  %.2f s' % ($t1-$t0))"
  done
done
for L in 0 1; do
  t0=$(date +%s.%N); MPYC_AMD_LAZY_INTS=$L python np_lpsolver.py -i5 --no-log 2>&1 | grep -v amdgpu.ids | md5sum; t1=$(date +%s.%N)
  python -c "print('np_lpsolver -i5  LAZY_INTS=$L  %.2f s' % ($t1-$t0))"
done
MPYC_AMD_LAZY_INTS=1 python -m cProfile -o $O/cprof_bnn32_r03.prof np_bnnmnist.py -d0 -o 1234 -b 32 --no-log > $O/cprof_bnn32_r03.log 2>&1
python - <<PY
import pstats
st = pstats.Stats('$O/cprof_bnn32_r03.prof')
st.sort_stats('tottime').print_stats(22)
PY
