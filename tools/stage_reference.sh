#!/bin/bash
# Build container only: copy the parts of the reference checkout that the drop-in acceptance tests run
# (tests/test_mpyc_dropin.py, -m gpu) into the UNTRACKED scratch directory _refstage/ (git-ignored, never
# committed; gpurun ships it to the GPU box, where /root/reference does not exist).
set -e
cd "$(dirname "$0")/.."
rm -rf _refstage && mkdir -p _refstage/demos
cp -r /root/reference/mpyc _refstage/mpyc
cp -r /root/reference/tests _refstage/tests
cp /root/reference/demos/np_*.py /root/reference/demos/pseudoinverse.py /root/reference/demos/sha3.py _refstage/demos/
cp -r /root/reference/demos/data _refstage/demos/data 2>/dev/null || true
find _refstage -name __pycache__ -prune -exec rm -rf {} +
du -sh _refstage
