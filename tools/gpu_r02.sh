#!/bin/bash
# Round-2 GPU pass: parity tests (incl. the reference's own suite under install(), needs _refstage/),
# bench line, rocprofv3 kernel trace of the bench and of the reference's np_aes demo under install().
#   STAGES="tests bench prof aes" (default all)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r02}
STAGES=${STAGES:-"tests bench prof aes"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  (time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -4 $O/pytest_gpu.log
fi
if has bench; then
  (time timeout 900 python bench.py --steps 50 --warmup 5 $BENCH_ARGS) > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
  tail -4 $O/bench.log | cut -c1-600
fi
if has bench2; then
  # N = 2 control flow of bench.py on a 1-GPU box: two ranks share GPU 0, gloo (device tensors staged through the host)
  (FFGPU_BENCH_BACKEND=gloo FFGPU_BENCH_DEVICE=0 FFGPU_BENCH_N=2000000 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
     --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 5 --warmup 2 --no-extras) > $O/bench2.log 2>&1
  echo "bench2 rc=$?" >> $O/bench2.log
  tail -3 $O/bench2.log | cut -c1-1500
fi
export TMPDIR=/tmp
if has prof; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline) > $O/rocprof_$T.log 2>&1
  echo "rocprof rc=$?" >> $O/rocprof_$T.log
fi
if has aes && [ -d $R/_refstage/mpyc ]; then
  # the reference's own demo (staged copy, unmodified), MPYC_GPU=1 -> mpyc_amd.install() via sitecustomize
  export PYTHONPATH=$R/mpyc_amd/autoinstall:$R:$R/_refstage MPYC_GPU=1 MPYC_AMD_TRACE_INSTALL=1
  for M in "" "-M3"; do
    tag=aes_m${M:-1}; tag=${tag/-M/}
    (cd $R/_refstage/demos && time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_$tag -- python np_aes.py -1 $M) > $O/rocprof_${T}_$tag.log 2>&1
    echo "np_aes $M rc=$?" >> $O/rocprof_${T}_$tag.log
    grep -E "69c4e0d8|install|rc=" $O/rocprof_${T}_$tag.log | head -8
  done
  find $O/prof_${T}_aes* -name '*kernel_stats.csv' | head
fi
