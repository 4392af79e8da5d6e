"""A/B of the batched-inverse kernels over 2^61 - 1 at n = 10^7 (FFGPU_INV_VARIANT, read per call by libffgpu): parity of
every variant (a * a^-1 == 1 on all non-zero inputs, zeros give zero, ragged lengths) and time per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(3)
p = 2**61 - 1
ctx = FieldContext(p, device=0)
# (the thirteen shapes of profiles/r04_alu.md were instantiations of k_inv_fast selected by this variable while the round-4
# kernel was chosen; the library keeps the winner -- CH 8, G 2, window 6, three waves, lean exponentiation -- and variant 0)
names = {1: 'k_inv_fast (round 4: CH8 G2 win6 waves3, lean exponentiation when the exponent allows it)',
         0: 'k_inv_batch (round 3: CH 8/10 by rounds, window table)'}
variants = [int(v) for v in os.environ.get('INV_VARIANTS', ','.join(str(k) for k in names)).split(',')]
for n in (10_000_000, 10_000_000 + 7, 4_000_003):
    sets = [(DevArray(ctx, bench.uniform_field(gen, n, p, 'cuda:0'), n), ctx.empty(n)) for _ in range(3)]
    a0, o0 = sets[0]
    a0.t[:5] = 0; a0.t[n // 2] = 0; a0.t[-1] = 0; a0.t[5] = 1; a0.t[6] = p - 1
    nz = a0.t != 0
    ref = None
    for v in variants:
        os.environ['FFGPU_INV_VARIANT'] = str(v)
        o0.t.zero_()
        ctx.inv(a0, out=o0, check_zero=False)
        one = ctx.mul(a0, o0).t
        ok = bool((one[nz] == 1).all()) and bool((o0.t[~nz] == 0).all()) and bool((o0.t < p).all())
        if ref is None:
            ref = o0.t.clone()
        same = torch.equal(ref, o0.t)
        try:
            ctx.inv(a0, out=o0, check_zero=True)
            zero_raised = False
        except ZeroDivisionError:
            zero_raised = True
        line = 'n=%d variant %2d %-50s parity %s same-as-variant-%d %s zero-flag %s' % (n, v, names.get(v, '?'), ok, variants[0], same, zero_raised)
        if n == 10_000_000:
            ms = min(bench.time_launches(lambda s: ctx.inv(s[0], out=s[1], check_zero=False), sets, 5) for _ in range(2))
            line += '  %7.1f us  frac of 8 TB/s %.3f' % (ms * 1e3, 16 * n / ms / 1e6 / 8000)
        print(line, flush=True)
        assert ok and same and zero_raised, line
    del sets
    torch.cuda.empty_cache()
