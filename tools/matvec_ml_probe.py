"""Matrix x vector and vector x matrix over the multi-limb default primes (80 / 128 bits) against the 61-bit prime: time per launch
and fraction of HBM for reading the 4096 x 4096 matrix once."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd.finfields import find_prime_root
gen = torch.Generator(device='cuda:0'); gen.manual_seed(4)
K = 4096
for bits in (61, 80, 128):
    p = 2**61 - 1 if bits == 61 else find_prime_root(bits)[0]
    ctx = FieldContext(p, device=0)
    eb = ctx.elem_bytes
    def rnd(n):
        if eb == 8:
            t = torch.randint(0, 2**60, (n,), dtype=torch.int64, device='cuda:0', generator=gen)
        elif eb == 12:
            t = torch.randint(0, 2**31 - 1, (n, 3), dtype=torch.int32, device='cuda:0', generator=gen)
        else:
            t = torch.randint(0, 2**62, (n, 2), dtype=torch.int64, device='cuda:0', generator=gen)
        a = DevArray(ctx, t, n)
        return ctx.reduce(a, out=a)
    mats = [rnd(K * K) for _ in range(3)]
    vec = rnd(K)
    for (M, N, tag) in ((K, 1, 'matrix x vector'), (1, K, 'vector x matrix'), (8, K, '8 rows x matrix')):
        if M == K:
            ms = bench.time_launches(lambda A: ctx.matmul(A, vec, K, K, 1), mats, 5)
        else:
            lhs = rnd(M * K)
            ms = bench.time_launches(lambda B: ctx.matmul(lhs, B, M, K, K), mats, 5)
        print('p=%3d bits %-18s %8.1f us  %.3f of 8 TB/s' % (bits, tag, ms * 1e3, K * K * eb / ms / 1e6 / 8000), flush=True)
    for S in (256, 384):                         # medium dense products: below the matrix-core threshold, the LDS-tiled k_matmul
        A_, B_ = rnd(S * S), rnd(S * S)
        ms = bench.time_launches(lambda z: ctx.matmul(A_, B_, S, S, S), [0], 10)
        print('p=%3d bits %dx%dx%d dense      %8.1f us  %.2f T field-MAC/s' % (bits, S, S, S, ms * 1e3, S**3 / ms / 1e9), flush=True)
