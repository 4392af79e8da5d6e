#!/bin/bash
# rocprofv3 kernel stats of tools/r02_probe.py (args: WHAT TAG [env assignments...])
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
WHAT=${1:-sbox,skinny}; TAG=${2:-probe}; shift; shift
export TMPDIR=/tmp "$@"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/tools/r02_probe.py $WHAT) > $O/probe_$TAG.log 2>&1
echo "rc=$?"; tail -2 $O/probe_$TAG.log | cut -c1-300
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/prof_$TAG/${TAG}_kernel_stats.csv')))
for r in rows:
    if any(s in r['Name'] for s in ('ffgpu::', 'k_gf8', 'k_sbox')):
        print(f"{float(r['AverageNs'])/1e3:9.2f} us x{r['Calls']:>4}  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}  {r['Name'][:150]}")
PY
