"""Table recombination over GF(2^128) / GF(2^64), n = 10^7: the runtime's own coefficient vectors (parties 1..4, 1..5, 1..7), k
distinct dense coefficients (k tables), grouped coefficients (rows that share a coefficient share a table); prints time, GB/s
and a digest of the result (round 6: how the mislabelled "dense" bench rows were found; the FFGPU_REC_VM switch this probe was
first written for -- some rows by direct multiplication beside the tables -- measured slower and is gone)."""
import os, sys, torch, random, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx as ggx, thresha as gth
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = 10_000_000
for name, mod, tail, eb in (('gf2_128', (1 << 128) | 0x87, (2,), 16), ('gf2_64', (1 << 64) | 0x1b, (), 8)):
    ctx = FieldContext(mod, binary=True, device=0)
    F = gff.GF(ggx.GFpX(2)(mod))
    x = torch.randint(-2**63, 2**63 - 1, (10, n) + tail, dtype=torch.int64, device='cuda:0', generator=gen)
    rows = [DevArray(ctx, x[i], n) for i in range(10)]
    out = rows[9]
    rg = random.Random(5)
    A, B = rg.randrange(2, F.order), rg.randrange(2, F.order)
    sets = [('k4 runtime xs=1..4', [int(v) for v in gth._recombination_vector(F, (1, 2, 3, 4), 0)]),
            ('k5 runtime xs=1..5', [int(v) for v in gth._recombination_vector(F, (1, 2, 3, 4, 5), 0)]),
            ('k7 runtime (ones)', [1] * 7),
            ('k3 distinct', [rg.randrange(2, F.order) for _ in range(3)]),
            ('k7 distinct', [rg.randrange(2, F.order) for _ in range(7)]),
            ('k9 distinct', [rg.randrange(2, F.order) for _ in range(9)]),
            ('k4 all equal', [A] * 4), ('k7 AAAABBB', [A] * 4 + [B] * 3)]
    for tag, lam in sets:
        plan = ctx.recombine_plan(rows[:len(lam)], lam, out)
        ms = bench.time_launches(lambda s: plan(), [0], 20)
        torch.cuda.synchronize()
        dg = hashlib.sha256(out.t.cpu().numpy().tobytes()).hexdigest()[:12]
        print(name, tag, '%.1f us  %.0f GB/s' % (ms * 1e3, (len(lam) + 1) * eb * n / ms / 1e6), dg, flush=True)
