"""Batched inverse and inverse square root ((3p - 5) / 4) over the multi-limb default primes of 80 / 96 / 128 / 136 bits at
n = 10^7: time per launch (round 6: k_inv_digits, digit chains in k_pow)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd.finfields import find_prime_root
gen = torch.Generator(device='cuda:0'); gen.manual_seed(3)
n = 10_000_000
for bits in (80, 96, 128, 136):
    p = find_prime_root(bits)[0]
    ctx = FieldContext(p, device=0)
    eb = ctx.elem_bytes
    shape = {12: ((n, 3), torch.int32, 2**31 - 1), 16: ((n, 2), torch.int64, 2**62), 24: ((n, 3), torch.int64, 2**62)}[eb]
    sets = []
    for _ in range(3):
        a = DevArray(ctx, torch.randint(1, shape[2], shape[0], dtype=shape[1], device='cuda:0', generator=gen), n)
        a = ctx.reduce(a, out=a)
        sets.append((a, ctx.empty(n)))
    ms = bench.time_launches(lambda s: ctx.inv(s[0], out=s[1], check_zero=False), sets, 5)
    print('p=%d bits inv %8.1f us  frac of 8 TB/s %.3f' % (bits, ms * 1e3, 2 * eb * n / ms / 1e6 / 8000))
    ms = bench.time_launches(lambda s: ctx.pow(s[0], (3 * p - 5) // 4, out=s[1]), sets, 3)
    print('p=%d bits inv sqrt %8.1f us' % (bits, ms * 1e3))
