#!/bin/bash
# Round-3 evidence pass on the GPU box: full GPU test suite, the bench line, rocprofv3 kernel trace + stats of the bench,
# PMC traffic passes (separate runs, kernel-trace only), SQ counters of the wide binary-field kernels and of the one-kernel
# S-box layer, kernel trace of the API-level program.   STAGES="tests bench prof pmc gf2w sbox api prss"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r03}
STAGES=${STAGES:-"tests bench prof pmc gf2w sbox api prss"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
export TMPDIR=/tmp
if has tests; then
  (time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
if has bench; then
  SECONDS=0
  (timeout 1500 python bench.py --steps 50 --warmup 5) > $O/bench.log 2> $O/bench.err; echo "bench rc=$? wall=${SECONDS}s" | tee -a $O/bench.err
  tail -c 400 $O/bench.log
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
if has gf2w; then bash tools/gpu_gf2w_pmc.sh ${T}_after > $O/gf2w_${T}_after.log 2>&1; tail -3 $O/gf2w_${T}_after.log; fi
if has sbox; then
  SO=$O/sbox_$T; mkdir -p $SO
  python tools/sbox_layer_time.py > $SO/plain.log 2>&1; cat $SO/plain.log
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    (cd /tmp && SBOX_N=1000000 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $SO/pmc$i -o pmc -- python $R/tools/sbox_layer_time.py) > $SO/pmc$i.log 2>&1
    echo "sbox pmc pass $i rc=$?"
  done
  python - <<PY | tee $SO/pmc_summary.txt
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$SO/pmc*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if 'k_gf8_sbox_layer' in name:
            acc[name + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
fi
if has api && [ -d $R/_refstage/mpyc ]; then
  AO=$O/api_$T; mkdir -p $AO
  export PYTHONPATH=$R/tests:$R:$R/_refstage
  for cfg in "1 10000000 20 1" "1 10000000 10 8" "1 100000000 5 1"; do
    set -- $cfg
    (cd /tmp && API_MODE=gpu API_N=$2 API_REPS=$3 API_WARMUP=3 API_CHAIN=$4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $AO/m$1_n$2_c$4 -o api -- python $R/tests/api_program.py --no-log) > $AO/m$1_n$2_c$4.log 2>&1
    grep API_RESULT $AO/m$1_n$2_c$4.log | cut -c1-600
  done
  (cd /tmp && API_MODE=gpu API_N=10000000 API_REPS=3 API_WARMUP=1 API_CPROFILE=$AO/m3_1e7.prof timeout 600 python $R/tests/api_program.py --no-log -M3) > $AO/m3_1e7.log 2>&1
  grep API_RESULT $AO/m3_1e7.log | cut -c1-600
  python - <<PY > $AO/m3_1e7_cprofile.txt 2>&1
import pstats
pstats.Stats('$AO/m3_1e7.prof').sort_stats('tottime').print_stats(28)
PY
fi
if has prss; then python tools/prss_time.py > $O/prss_$T.log 2>&1; cat $O/prss_$T.log; fi
