"""share generation with the device CSPRNG, configs[2] shape (GF(2^64-189), m=7, t=3) and the headline shape (GF(2^61-1), m=3, t=1):
a few launches for rocprofv3 --pmc runs (SQ_INSTS_VALU per secret)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = 10_000_000
for p, t, m in ((2**64 - 189, 3, 7), (2**61 - 1, 1, 3)):
    ctx = FieldContext(p, device=0)
    sets = [(DevArray(ctx, bench.uniform_field(gen, n, p, 'cuda:0'), n), ctx.empty_matrix(m, n)) for _ in range(3)]
    for rounds in (20, 8):
        ms = bench.time_launches(lambda s: ctx.split_rng(s[0], t, m, key=bytes(range(32)), nonce=9, rounds=rounds, out=s[1]), sets, 4)
        print('p=%d bits m=%d t=%d chacha%d: %.1f us  %.0f GB/s frac %.3f' % (p.bit_length(), m, t, rounds, ms * 1e3, (1 + m) * 8 * n / ms / 1e6, (1 + m) * 8 * n / ms / 1e6 / 8000))
