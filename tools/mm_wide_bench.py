"""Dense products over two-limb primes: matrix cores (default) vs the VALU kernel (FFGPU_MM_MFMA=0 in the environment)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
print('FFGPU_MM_MFMA =', os.environ.get('FFGPU_MM_MFMA', '1'))
for name, P in (('2^128-173', 2**128 - 173), ('2^96-17', 2**96 - 17), ('generic 128', 258797994007609146293811961253269568351)):
    ctx = FieldContext(P, device=0)
    for d in (1024, 2048) + ((4096,) if os.environ.get('FFGPU_MM_MFMA', '1') != '0' else ()):
        lb = ctx.limbs
        dt = torch.int32 if ctx.elem_bytes == 12 else torch.int64
        hi = 2**31 - 1 if ctx.elem_bytes == 12 else 2**62
        A = DevArray(ctx, torch.randint(0, hi, (d * d, lb), dtype=dt, device='cuda:0', generator=gen), d * d)
        B = DevArray(ctx, torch.randint(0, hi, (d * d, lb), dtype=dt, device='cuda:0', generator=gen), d * d)
        ctx.reduce(A, out=A); ctx.reduce(B, out=B)
        C = ctx.empty(d * d)
        ms = bench.time_launches(lambda s: ctx.matmul(A, B, d, d, d, out=C), [0], 2)
        print(f'{name:12s} {d}^3: {ms:9.3f} ms  {d**3/ms/1e9:7.3f} TMAC/s')
