"""One line per (field family, kernel): achieved GB/s at a fixed size, to spot slow paths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx, thresha as gth
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
FAM = [('PM64 2^61-1', 2**61 - 1, False), ('PM64 2^64-189', 2**64 - 189, False), ('RC64 generic', 6616326157076047771, False),
       ('RC32 2^31-1', 2**31 - 1, False), ('PM128 2^128-173', 2**128 - 173, False), ('PM128 2^127-1', 2**127 - 1, False), ('PM96 2^96-17', 2**96 - 17, False),
       ('MONT128 generic', 258797994007609146293811961253269568351, False), ('PM192 2^136-c', gff.find_prime_root(136)[0], False), ('GF2P8 0x11b', 0x11b, True),
       ('GF2W64 2^64', (1 << 64) | 0x1b, True), ('GF2W128 2^128', (1 << 128) | 0x87, True)]
only = sys.argv[1:] 
for name, mod, binary in FAM:
    if only and not any(o in name for o in only):
        continue
    ctx = FieldContext(mod, binary, device=0)
    eb = ctx.elem_bytes
    n = 40_000_000 if eb == 1 else 10_000_000
    F = gff.GF(gfpx.BinaryPolynomial(mod)) if binary else gff.GF(mod)
    def rnd(rows):
        if eb == 24:
            x = torch.randint(0, 2**62, (rows, n, 3), dtype=torch.int64, device='cuda:0', generator=gen)
            x[..., 2] &= 0x7f                      # below 2^135: canonical
        elif eb == 16:
            x = torch.randint(0, 2**62, (rows, n, 2), dtype=torch.int64, device='cuda:0', generator=gen)
        elif eb == 12:
            x = torch.randint(0, 2**31 - 1, (rows, n, 3), dtype=torch.int32, device='cuda:0', generator=gen)
        elif eb == 8:
            x = torch.randint(0, 2**60, (rows, n), dtype=torch.int64, device='cuda:0', generator=gen)
        elif eb == 4:
            x = torch.randint(0, 2**31 - 1, (rows, n), dtype=torch.int32, device='cuda:0', generator=gen)
        else:
            x = torch.randint(0, 256, (rows, n), dtype=torch.uint8, device='cuda:0', generator=gen)
        return x
    res = []
    sets = []
    for _ in range(3):
        x = rnd(3)
        sets.append([DevArray(ctx, x[i], n) for i in range(3)])
    ms = bench.time_launches(lambda s: ctx.mul(s[0], s[1], out=s[2]), sets, 3)
    res.append(('mul', 3 * eb * n / ms / 1e6))
    ms = bench.time_launches(lambda s: ctx.add(s[0], s[1], out=s[2]), sets, 3)
    res.append(('add', 3 * eb * n / ms / 1e6))
    for (t, m) in ((1, 3), (3, 7)):
        if m >= F.order:
            continue
        coef = ctx.empty_matrix(t, n)
        cr = rnd(t)
        for j in range(t):
            coef.row(j).t.copy_(cr[j])
        if not binary:   # canonicalise random limbs
            for j in range(t):
                ctx.reduce(coef.row(j), out=coef.row(j))
        sh = [ctx.empty_matrix(m, n) for _ in range(2)]
        ms = bench.time_launches(lambda s: ctx.split(sets[0][0], coef, t, m, out=s), sh, 3)
        res.append((f'split m{m}t{t}', (1 + t + m) * eb * n / ms / 1e6))
        k = 2 * t + 1
        lam = list(gth._recombination_vector(F, tuple(range(1, k + 1)), 0))
        outs = [ctx.empty(n) for _ in range(2)]
        plans = [ctx.recombine_plan([sh[i].row(j) for j in range(k)], lam, outs[i]) for i in range(2)]
        ms = bench.time_launches(lambda pl: pl(), plans, 3)
        res.append((f'rec k{k}', (k + 1) * eb * n / ms / 1e6))
        ms = bench.time_launches(lambda s: ctx.split_rng(sets[0][0], t, m, key=bytes(32), nonce=1, rounds=20, out=s), sh, 3)
        res.append((f'split_rng m{m}t{t}', (1 + m) * eb * n / ms / 1e6))
        del sh, coef, outs, plans
    print(f'{name:18s} ' + '  '.join(f'{k}={v:6.0f}' for k, v in res))
    del sets
    torch.cuda.empty_cache()
