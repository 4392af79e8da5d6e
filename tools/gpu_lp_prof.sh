cd /root/repo/_refstage/demos
export PYTHONPATH=/root/repo/mpyc_amd/autoinstall:/root/repo:/root/repo/_refstage MPYC_GPU=1
for i in 1 2; do for L in 0 1; do t0=$(date +%s.%N); MPYC_AMD_LAZY_INTS=$L python np_lpsolver.py -i5 --no-log > /dev/null 2>&1; t1=$(date +%s.%N); python -c "print('np_lpsolver -i5 LAZY=$L %.2f s' % ($t1-$t0))"; done; done
MPYC_AMD_LAZY_INTS=1 python -m cProfile -o /root/repo/gpurun_out/cprof_lp5_r03.prof np_lpsolver.py -i5 --no-log > /dev/null 2>&1
python - <<PY
import pstats
pstats.Stats('/root/repo/gpurun_out/cprof_lp5_r03.prof').sort_stats('tottime').print_stats(25)
PY
for L in 0 1; do t0=$(date +%s.%N); MPYC_AMD_LAZY_INTS=$L python np_bnnmnist.py -d0 -o 1234 -b 32 --no-log > /dev/null 2>&1; t1=$(date +%s.%N); python -c "print('np_bnnmnist -b 32 LAZY=$L %.2f s' % ($t1-$t0))"; done
