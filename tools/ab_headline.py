"""A/B of the headline kernels over GF(2^61-1) between two builds of libffgpu on the SAME box: FFGPU_LIB_AB=<path> loads
that library instead of mpyc_amd/libffgpu.so (used once, after the round-3 product rewrite: is the fused kernel slower?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpyc_amd import _ffi
alt = os.environ.get('FFGPU_LIB_AB')
if alt:
    _ffi.LIB_PATH = alt
import torch, bench
from mpyc_amd.engine import FieldContext
P = 2**61 - 1
ctx = FieldContext(P, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n, t, m = 10_000_000, 1, 3
sets = [bench.StepData(ctx, n, t, m, gen) for _ in range(4)]
res = {}
for rep in range(3):
    res.setdefault('mul', []).append(bench.time_launches(lambda s: ctx.mul(s.a, s.b, out=s.c), sets, 20))
    res.setdefault('split', []).append(bench.time_launches(lambda s: ctx.split(s.c, s.coef, t, m, out=s.shares), sets, 20))
    res.setdefault('mul_split', []).append(bench.time_launches(lambda s: ctx.split(s.a, s.coef, t, m, out=s.shares, mul_by=s.b), sets, 20))
print(os.path.basename(_ffi.LIB_PATH), {k: ['%.2f us' % (v * 1e3) for v in vs] for k, vs in res.items()})
