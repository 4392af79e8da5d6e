#!/bin/bash
# secure fixed-point multiplication (np_multiply + np_trunc) through the reference runtime: the reference vs install(), m = 1 and -M3
cd /tmp; export PYTHONPATH=$GRAFT_REPO_ROOT/_refstage
for cfg in "ref 100000 " "gpu 100000 " "gpu 1000000 " "gpu 10000000 " "ref 100000 -M3" "gpu 100000 -M3" "gpu 1000000 -M3"; do set -- $cfg; API_MODE=$1 API_N=$2 timeout 600 python $GRAFT_REPO_ROOT/tools/fxp_api_probe.py --no-log $3 2>&1 | grep -E "RESULT|Error|BAD count" | cut -c1-160 | tail -2; done
# the rare 2^48 outliers are the reference's own: with the mask length of SCALAR fixed-point numbers (l + f) they vanish
for E in "FXP_REF_L=1" "FXP_REF_L="; do echo "== np_trunc replayed with $E"; env $E FXP_CHECK=1 API_MODE=gpu API_N=6000000 timeout 900 python $GRAFT_REPO_ROOT/tools/fxp_api_probe.py --no-log 2>&1 | grep -E "RESULT|BAD count" | cut -c1-100 | tail -2; done
