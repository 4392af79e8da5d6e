import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for name, mod, shape_tail, eb in (('GF(2^128)', (1 << 128) | 0x87, (2,), 16), ('GF(2^64)', (1 << 64) | 0x1b, (), 8), ('GF(2^100)', None, (2,), 16)):
    if mod is None:
        from mpyc_amd.finfields import find_irreducible
        mod = int(find_irreducible(2, 100))
    ctx = FieldContext(mod, binary=True, device=0)
    n = 10_000_000
    bufs = []
    for _ in range(4):
        x = torch.randint(-2**63, 2**63 - 1, (3, n) + shape_tail, dtype=torch.int64, device='cuda:0', generator=gen)
        if name == 'GF(2^100)':
            x[..., 1] &= (1 << 36) - 1
        bufs.append(tuple(DevArray(ctx, x[i], n) for i in range(3)))
    ms = bench.time_launches(lambda s: ctx.mul(s[0], s[1], out=s[2]), bufs, 3)
    print(os.environ.get('FFGPU_GF2W_BITSERIAL', 'window'), name, 'mul %.1f us  %.1f G mul/s  %.0f GB/s' % (ms * 1e3, n / ms / 1e6, 3 * eb * n / ms / 1e6))
    del bufs; torch.cuda.empty_cache()
