#!/bin/bash
# The reference's own vectorised demos (demos/np-run-all.sh), unmodified, with and without mpyc_amd.install():
# same printed results expected (offsets / seeds fixed so that the runs are comparable;
# np_cnnmnist is left out: its weight files are not part of the checkout).  usage: run_np_demos.sh <mpyc checkout> [cpuctx]   (cpuctx: build container, no GPU)
REF=${1:-_refstage}; MODE=${2:-gpu}
R=$(cd "$(dirname "$0")/.." && pwd)
REF=$(cd "$REF" && pwd)
SITE=$R/mpyc_amd/autoinstall; EXTRA=""
if [ "$MODE" = cpuctx ]; then SITE=$R/tests/devsite; EXTRA="MPYC_AMD_CPUCTX=1"; fi
cd $REF/demos
fail=0
while read -r line; do
  [ -z "$line" ] && continue
  loose=0
  if [ "${line:0:1}" = "~" ]; then loose=1; line=${line:1}; fi      # ~: output depends on fresh randomness, compare exit codes only
  t0=$(date +%s.%N)
  env PYTHONPATH=$REF python $line --no-log 2>&1 | grep -v 'amdgpu.ids' > /tmp/demo_ref.txt; rc0=${PIPESTATUS[0]}
  t1=$(date +%s.%N)
  env $EXTRA MPYC_GPU=1 PYTHONPATH=$SITE:$R/tests:$R:$REF python $line --no-log 2>&1 | grep -v 'amdgpu.ids' > /tmp/demo_gpu.txt; rc1=${PIPESTATUS[0]}
  t2=$(date +%s.%N)
  if [ $rc0 -eq 0 ] && [ $rc1 -eq 0 ] && cmp -s /tmp/demo_ref.txt /tmp/demo_gpu.txt; then st=SAME;
  elif [ $rc0 -eq 0 ] && [ $rc1 -eq 0 ] && [ $loose -eq 1 ]; then st="OK(random)";
  else st="DIFF(rc $rc0/$rc1)"; fail=1; fi
  printf "%-34s %-14s reference %6.2f s   install() %6.2f s   %d output lines\n" "$line" "$st" $(python -c "print($t1-$t0)") $(python -c "print($t2-$t1)") $(wc -l < /tmp/demo_ref.txt)
  if [ "${st:0:4}" = DIFF ]; then diff /tmp/demo_ref.txt /tmp/demo_gpu.txt | head -12; fi
done <<L
${DEMOS:-~pseudoinverse.py
np_id3gini.py
np_lpsolver.py
np_lpsolver.py -i5
~np_lpsolverfxp.py
np_bnnmnist.py -d0 -o 1234
np_aes.py -1
np_onewayhashchains.py -k2 --no-random-seed
sha3.py}
L
exit $fail
