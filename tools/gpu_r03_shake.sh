#!/bin/bash
# SHAKE128 for PRSS: host backends, the PRSS call end to end, and the device probe (tools/shake_dev.hip).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "== host cores: $(nproc)"
python tools/shake_host_time.py
FFGPU_SHAKE_OWN=1 python tools/shake_host_time.py
echo "== PRSS call (libcrypto backend)"; python tools/prss_time.py
echo "== PRSS call (own Keccak)"; FFGPU_SHAKE_OWN=1 python tools/prss_time.py
echo "== device probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/shake_dev tools/shake_dev.hip 2>/dev/null && timeout 300 /tmp/shake_dev > /tmp/shake_dev.out; cat /tmp/shake_dev.out
python tools/shake_dev_check.py > /tmp/shake_want.txt
grep CHECK /tmp/shake_dev.out | diff - /tmp/shake_want.txt && echo "device SHAKE128 == hashlib"
} > gpurun_out/shake.log 2>&1
tail -40 gpurun_out/shake.log
