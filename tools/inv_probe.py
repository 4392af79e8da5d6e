"""Batched inverse, square root ((p+1)/4) and Legendre ((p-1)/2) exponentiations at n = 10^7: time per launch and fraction of
the 8 TB/s HBM peak (one read + one write per element)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(3)
n = 10_000_000
for p in ((2**61 - 1,) if os.environ.get('PROBE_P61_ONLY') else (2**61 - 1, 2**64 - 189, 2**40 - 87)):
    ctx = FieldContext(p, device=0)
    eb = ctx.elem_bytes
    sets = [(DevArray(ctx, bench.uniform_field(gen, n, p, 'cuda:0'), n), ctx.empty(n)) for _ in range(3)]
    a0, o0 = sets[0]
    ctx.inv(a0, out=o0, check_zero=False)
    one = ctx.mul(a0, o0).t
    nz = int((a0.t != 0).sum())
    assert int((one == 1).sum()) == nz, (p, int((one == 1).sum()), nz)
    res = []
    ms = bench.time_launches(lambda s: ctx.inv(s[0], out=s[1], check_zero=False), sets, 5)
    res.append(('inv', ms))
    if p % 4 == 3:
        res.append(('sqrt (p+1)/4', bench.time_launches(lambda s: ctx.pow(s[0], (p + 1) // 4, out=s[1]), sets, 3)))
    res.append(('legendre (p-1)/2', bench.time_launches(lambda s: ctx.pow(s[0], (p - 1) // 2, out=s[1]), sets, 3)))
    res.append(('pow 65537', bench.time_launches(lambda s: ctx.pow(s[0], 65537, out=s[1]), sets, 3)))
    for name, ms in res:
        print('p=%d bits %-18s %8.1f us  frac of 8 TB/s %.3f' % (p.bit_length(), name, ms * 1e3, 2 * eb * n / ms / 1e6 / 8000))
