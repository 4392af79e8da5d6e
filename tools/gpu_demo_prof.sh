#!/bin/bash
# rocprofv3 kernel stats of one reference demo under install():  gpu_demo_prof.sh TAG <demo and args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=$1; shift
export PYTHONPATH=$R/mpyc_amd/autoinstall:$R:$R/_refstage MPYC_GPU=1
cd $R/_refstage/demos
(time timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python "$@" --no-log) > $O/demo_$TAG.log 2>&1
grep -v "^[EWI]2026" $O/demo_$TAG.log | tail -6
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/prof_$TAG/${TAG}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'GPU time total {tot/1e6:.1f} ms in {sum(int(r["Calls"]) for r in rows)} launches')
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}%  x{r['Calls']:>6}  avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Name'][:120]}")
PY
