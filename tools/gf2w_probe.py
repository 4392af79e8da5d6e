"""GF(2^128)/GF(2^64) kernels of configs[4] for rocprofv3 runs: mul, recombine k=4/k=7 at n = 10^7 (few launches each)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx as ggx, thresha as gth
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = int(os.environ.get('GF2W_N', 10_000_000))
for name, mod, tail, eb in (('gf2_128', (1 << 128) | 0x87, (2,), 16), ('gf2_64', (1 << 64) | 0x1b, (), 8)):
    ctx = FieldContext(mod, binary=True, device=0)
    F = gff.GF(ggx.GFpX(2)(mod))
    x = torch.randint(-2**63, 2**63 - 1, (9, n) + tail, dtype=torch.int64, device='cuda:0', generator=gen)
    rows = [DevArray(ctx, x[i], n) for i in range(9)]
    out = rows[8]
    ms = bench.time_launches(lambda s: ctx.mul(rows[0], rows[1], out=out), [0], 5)
    print(name, 'mul %.1f us %.0f GB/s' % (ms * 1e3, 3 * eb * n / ms / 1e6))
    if os.environ.get('GF2W_MUL_ONLY'):
        continue
    for k in (4, 5, 7):
        xs = [((3 + j) % 7) + 1 for j in range(k)] if os.environ.get('GF2W_ROT') else list(range(1, k + 1))
        lam = [int(v) for v in gth._recombination_vector(F, tuple(xs), 0)]
        plan = ctx.recombine_plan(rows[:k], lam, out)
        ms = bench.time_launches(lambda s: plan(), [0], 5)
        print(name, 'recombine k=%d xs=%s lam=%s: %.1f us %.0f GB/s' % (k, xs, [hex(v) for v in lam][:3], ms * 1e3, (k + 1) * eb * n / ms / 1e6))
    import random
    # dense coefficients: the same two coefficient sets on the same two row layouts -- rows that are slices of ONE tensor
    # (n * eb bytes apart) and rows of a share matrix (engine.empty_matrix: pitch padded to 256 B and skewed) -- to settle
    # which of the two in-repo measurements of this shape (this probe: 413 us, bench.py: 272 us for GF(2^128), k = 7) is what
    mtx = ctx.empty_matrix(8, n)
    for i in range(8):
        mtx.row(i).t.copy_(x[i])
    for k in (4, 7):
        rg3, rgb = random.Random(3), random.Random(1000 + k)
        A = rg3.randrange(2, F.order)
        for tag, lam in (('k DISTINCT coefficients, Random(3)', [rg3.randrange(2, F.order) for _ in range(k)]),
                         ('k DISTINCT coefficients, Random(1000+k) [bench.py]', [rgb.randrange(2, 1 << (8 * eb)) for _ in range(k)]),
                         ('ONE coefficient k times (grouped rows: what bench.py measured as "dense" until round 6)', [A] * k)):
            for layout, rr, oo in (('slices of one tensor', rows[:k], out), ('share-matrix rows', [mtx.row(j) for j in range(k)], mtx.row(7))):
                plan = ctx.recombine_plan(rr, lam, oo)
                ms = bench.time_launches(lambda s: plan(), [0], 5)
                print(name, 'recombine k=%d %s, %s: %.1f us %.0f GB/s' % (k, tag, layout, ms * 1e3, (k + 1) * eb * n / ms / 1e6))
