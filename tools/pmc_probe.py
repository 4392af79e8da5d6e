#!/usr/bin/env python3
"""Launch each hot kernel a few times on rotating buffers; run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and a second pass with --pmc WRITE_SIZE)
to get HBM traffic per launch.  k_copy16 (exactly 80 MB read + 80 MB written) calibrates the counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mpyc_amd.engine import FieldContext
from mpyc_amd import finfields as gff, gfpx, thresha as gth

torch.cuda.set_device(0)
n = 10_000_000
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for P, t, m in ((bench.P61, 1, 3), (bench.P64, 3, 7)):
    ctx = FieldContext(P, device=0)
    sets = [bench.StepData(ctx, n, t, m, gen) for _ in range(4)]
    k = 2 * t + 1
    lam = list(gth._recombination_vector(gff.GF(P), tuple(range(1, k + 1)), 0))
    for s in sets:
        s.rec = ctx.recombine_plan([s.shares.row(j) for j in range(k)], lam, s.y)
    for rep in range(3):
        for s in sets:
            ctx.copy(s.a.t, s.y.t)
            ctx.mul(s.a, s.b, out=s.c)
            ctx.split(s.c, s.coef, t, m, out=s.shares)
            ctx.split(s.a, s.coef, t, m, out=s.shares, mul_by=s.b)
            s.rec()
    torch.cuda.synchronize()
    assert torch.equal(sets[0].y.t, sets[0].c.t)
    del sets
    torch.cuda.empty_cache()
# 12-byte elements (PM96): does a dwordx3 stream move exactly its algorithmic bytes?
from mpyc_amd.engine import DevArray
ctx = FieldContext(2**96 - 17, device=0)
bufs = []
for _ in range(4):
    x = torch.randint(0, 2**31 - 1, (3, n, 3), dtype=torch.int32, device='cuda:0', generator=gen)
    bufs.append([DevArray(ctx, x[i], n) for i in range(3)])
for rep in range(3):
    for b in bufs:
        ctx.mul(b[0], b[1], out=b[2])
        ctx.add(b[0], b[1], out=b[2])
torch.cuda.synchronize()
del bufs
torch.cuda.empty_cache()
# configs[3]: two-limb prime 2^128-173, m = 7, t = 3 (mul, split, recombine k = 7)
P128 = 2**128 - 173
ctx = FieldContext(P128, device=0)
t, m, k = 3, 7, 7
lam = list(gth._recombination_vector(gff.GF(P128), tuple(range(1, k + 1)), 0))
sets = []
for _ in range(3):
    ab = bench.u128_rows(2 + t, n, 'cuda:0', gen)
    coef = ctx.empty_matrix(t, n)
    for j in range(t):
        coef.row(j).t.copy_(ab[2 + j])
    sh = ctx.empty_matrix(m, n)
    y, c = ctx.empty(n), ctx.empty(n)
    sets.append((DevArray(ctx, ab[0].contiguous(), n), DevArray(ctx, ab[1].contiguous(), n), coef, sh, y, c,
                 ctx.recombine_plan([sh.row(j) for j in range(k)], lam, y)))
    del ab
for rep in range(3):
    for a_, b_, coef, sh, y, c, rec in sets:
        ctx.mul(a_, b_, out=c)
        ctx.split(c, coef, t, m, out=sh)
        rec()
torch.cuda.synchronize()
assert torch.equal(sets[0][4].t, sets[0][5].t)
del sets
torch.cuda.empty_cache()
# 24-byte elements (136-bit prime), round 6: the wave moves 1536 contiguous bytes as dwordx4 accesses in which lanes 32..63
# repeat the second access of lanes 0..31 -- do the repeated lanes cost HBM traffic?  (mul, share generation m=3,t=1 with
# supplied coefficients, recombination k=3)
P136 = gff.find_prime_root(136)[0]
ctx = FieldContext(P136, device=0)
t, m, k = 1, 3, 3
lam = list(gth._recombination_vector(gff.GF(P136), tuple(range(1, k + 1)), 0))
sets = []
for _ in range(3):
    x = torch.randint(0, 2**62, (3, n, 3), dtype=torch.int64, device='cuda:0', generator=gen)
    rows = [ctx.reduce(DevArray(ctx, x[i], n)) for i in range(3)]
    coef = ctx.empty_matrix(t, n)
    coef.row(0).t.copy_(rows[2].t)
    sh = ctx.empty_matrix(m, n)
    y, c = ctx.empty(n), ctx.empty(n)
    sets.append((rows[0], rows[1], coef, sh, y, c, ctx.recombine_plan([sh.row(j) for j in range(k)], lam, y)))
    del x
for rep in range(3):
    for a_, b_, coef, sh, y, c, rec in sets:
        ctx.mul(a_, b_, out=c)
        ctx.split(c, coef, t, m, out=sh)
        rec()
torch.cuda.synchronize()
assert torch.equal(sets[0][4].t, sets[0][5].t)
print('pmc probe done')
