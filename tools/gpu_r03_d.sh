#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_pm192.py tests/test_gpu_sweep.py -m gpu -x -q -k "pow or inv or sqrt or recip or legendre or is_sqr or second or sweep" > gpurun_out/r03d_tests.log 2>&1; tail -4 gpurun_out/r03d_tests.log
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = 10_000_000
for name, p in (('p61', 2**61 - 1), ('p64', 2**64 - 189), ('p128', 2**128 - 173), ('p31', 2**31 - 1)):
    ctx = FieldContext(p, device=0)
    eb = ctx.elem_bytes
    sets = []
    for _ in range(3):
        if eb == 16:
            x = torch.randint(1, 2**62, (2, n, 2), dtype=torch.int64, device='cuda:0', generator=gen)
        elif eb == 8:
            x = bench.uniform_field(gen, 2 * n, p, 'cuda:0').reshape(2, n)
        else:
            x = torch.randint(1, 2**31 - 1, (2, n), dtype=torch.int32, device='cuda:0', generator=gen)
        sets.append([DevArray(ctx, x[0], n), DevArray(ctx, x[1], n)])
    ms = bench.time_launches(lambda s: ctx.inv(s[0], out=s[1], check_zero=False), sets, 3)
    chk = ctx.mul(sets[0][0], sets[0][1]).t
    print(name, 'inv %.1f us  %.0f GB/s frac %.3f' % (ms * 1e3, 2 * eb * n / ms / 1e6, 2 * eb * n / ms / 1e6 / 8000), 'ok' if bool((chk.reshape(n, -1)[:, 0] == 1).all()) else 'WRONG')
    ms = bench.time_launches(lambda s: ctx.pow(s[0], (p + 1) // 4, out=s[1]), sets, 2)
    print(name, 'pow (p+1)/4 %.1f us' % (ms * 1e3))
    ms = bench.time_launches(lambda s: ctx.pow(s[0], p - 2, out=s[1]), sets, 2)
    print(name, 'pow p-2 %.1f us' % (ms * 1e3))
PY
