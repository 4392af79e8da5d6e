#!/bin/bash
# SQ counters of the round-4 kernels: batched inverse (k_inv_fast vs k_inv_batch), bit-sliced GF(2^64) product, S-box layer
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ctr_r04; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (INV_VARIANTS=1,0 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/inv$i -o pmc -- python $R/tools/inv_ab.py) > $O/inv$i.log 2>&1
  (GF2W_N=10000000 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/gf$i -o pmc -- python $R/tools/gf2w_probe.py) > $O/gf$i.log 2>&1
  (SBOX_N=100000000 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sb$i -o pmc -- python $R/tools/sbox_layer_time.py) > $O/sb$i.log 2>&1
  echo "pass $i done"
done
python - <<PY | tee $O/summary.txt
import csv, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        if any(k in name for k in ('k_inv_fast', 'k_inv_batch', 'k_gf2w64_mul_bitsliced', 'k_ew2<GF2W128, 2', 'k_gf8_sbox_layer<3, 1, 1>')):
            acc[name + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {sum(v)/len(v):14.4g} per launch ({len(v)} launches)')
PY
