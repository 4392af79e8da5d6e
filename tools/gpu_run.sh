#!/bin/bash
# The ONE script for passes on the GPU box (replaces the per-round gpu_r0*.sh).  STAGES (any subset, in this order):
#   newtests  pytest -m gpu on $NEWTESTS (files / node ids)          tests   the whole -m gpu suite
#   bench     bench.py --steps 20 --warmup 5 (the driver's line)      prof    rocprofv3 --kernel-trace --stats of the bench
#   benchfull the same with --full (every API-level leg, no budget)
#   pmc       FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py    valu    SQ_INSTS_VALU pass of tools/valu_probe.py
#   counters  SQ counter passes (waves, busy / wave cycles, VALU / LDS instructions and waits) of the GF(2^n) products and PRSS
#   probe     bash -c "$PROBE" (timeout $PROBE_TIMEOUT, default 600)
# TAG names the outputs (default r06): gpurun_out/{pytest_gpu,bench,rocprof_$TAG,...}.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r06}
STAGES=${STAGES:-"tests bench prof"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
export TMPDIR=/tmp
if has newtests; then
  (time timeout ${NEWTESTS_TIMEOUT:-1500} python -m pytest -m gpu -x -q --durations=8 ${NEWTESTS}) > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
  tail -${NEWTESTS_TAIL:-25} $O/pytest_new.log
fi
if has tests; then
  (time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -25 $O/pytest_gpu.log
fi
if has bench; then
  SECONDS=0
  (timeout 1500 python bench.py --steps 20 --warmup 5) > $O/bench.log 2> $O/bench.err; echo "bench rc=$? wall=${SECONDS}s" | tee -a $O/bench.err
  tail -n 1 $O/bench.log | wc -c
  tail -n 1 $O/bench.log | cut -c1-2500
fi
if has benchfull; then
  SECONDS=0
  (timeout 2400 python bench.py --steps 20 --warmup 5 --full) > $O/bench_full.log 2> $O/bench_full.err; echo "bench --full rc=$? wall=${SECONDS}s" | tee -a $O/bench_full.err
  cp bench_detail.json $O/bench_detail_full.json 2>/dev/null
  tail -n 1 $O/bench_full.log | cut -c1-600
fi
if has prof; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-leg) > $O/rocprof_$T.log 2>&1
  echo "rocprof rc=$?"
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${T}_$C -o $C -- python $R/tools/pmc_probe.py) > $O/pmc_${T}_$C.log 2>&1
    echo "pmc $C rc=$?"
  done
fi
if has valu; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/valu_$T -o valu -- python $R/tools/valu_probe.py) > $O/valu_$T.log 2>&1
  F=$(find $O/valu_$T -name '*counter_collection.csv' | head -1)
  python tools/valu_summary.py $F $O/valu_order.json $O/${T}_valu.json $O/${T}_valu.md | tail -20
fi
if has counters; then
  # SQ counters (three passes of four) of the GF(2^n) products (tools/gf2w_probe.py) and PRSS production mode (tools/prss_chacha_time.py)
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    # (one rocprofv3 run per program: processes of one run would overwrite each other's output files)
    (cd /tmp && GF2W_MUL_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/ctr_$T/p${i}_gf -o pmc -- python $R/tools/gf2w_probe.py) > $O/ctr_${T}_p${i}_gf.log 2>&1
    (cd /tmp && PRSS_N=10000000 PRSS_ONLY61=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/ctr_$T/p${i}_prss -o pmc -- python $R/tools/prss_chacha_time.py) > $O/ctr_${T}_p${i}_prss.log 2>&1
    echo "counters pass $i rc=$?"
  done
  python - <<PY | tee $O/${T}_sq_counters.txt
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/ctr_$T/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if any(k in name for k in ('k_gf2w64_mul_bitsliced', 'k_ew2<GF2W128, 2', 'k_ew2<GF2W64, 2', 'k_prss_chacha', 'k_gf2w_recombine_tab')):
            acc[name + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
fi
if has probe; then
  (timeout ${PROBE_TIMEOUT:-600} bash -c "$PROBE") > $O/probe_$T.log 2>&1; echo "probe rc=$?" >> $O/probe_$T.log
  tail -${PROBE_TAIL:-60} $O/probe_$T.log
fi
