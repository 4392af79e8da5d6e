#!/bin/bash
# Round-end GPU pass in ONE box: parity tests, bench, kernel trace of the bench, PMC traffic passes, the reference's
# np_aes demo under install() with a kernel trace, np_lpsolver -i5 (136-bit root-of-unity field) on 1 and 3 parties
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${TAG:-r02}
STAGES="${FINAL_STAGES:-tests bench prof aes}" TAG=$T bash tools/gpu_r02.sh
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${T}_$C -o $C -- python $R/tools/pmc_probe.py) > $O/pmc_${T}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
[ -n "$NO_DEMOS" ] || DEMOS="np_lpsolver.py -i5
np_lpsolver.py -i5 -M3" timeout 400 bash tools/run_np_demos.sh _refstage gpu 2>&1 | tee $O/lp5.log | tail -3
