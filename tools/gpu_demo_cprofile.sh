#!/bin/bash
# cProfile of one reference demo under install() on the GPU box: where the HOST time goes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
export PYTHONPATH=$R/mpyc_amd/autoinstall:$R:$R/_refstage MPYC_GPU=1
cd $R/_refstage/demos
python -m cProfile -o $O/cprof_$TAG.prof "$@" --no-log > $O/cprof_$TAG.log 2>&1
python - <<PY
import pstats
st = pstats.Stats('$O/cprof_$TAG.prof')
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(45)
PY
