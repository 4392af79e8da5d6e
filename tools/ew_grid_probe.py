"""VALU-bound element-wise products under a capped grid (FFGPU_BLOCKS_PER_CU, read once per device): one pack per thread
(uncapped, the streaming default) against a grid-stride loop over a few workgroups per CU.  n = 10^7."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = 10_000_000
print('FFGPU_BLOCKS_PER_CU =', os.environ.get('FFGPU_BLOCKS_PER_CU', '(unset)'))
for name, mod, binary in (('gf2_128', (1 << 128) | 0x87, True), ('gf2_64_multiplier', (1 << 64) | 0x1b, True), ('mont128_generic', 258797994007609146293811961253269568351, False),
                          ('p136', None, False), ('p128', 2**128 - 173, False), ('p61', 2**61 - 1, False)):
    if mod is None:
        from mpyc_amd.finfields import find_prime_root
        mod = find_prime_root(136)[0]
    if name == 'gf2_64_multiplier':
        os.environ['FFGPU_GF2W_BITSLICED'] = '0'
    ctx = FieldContext(mod, binary=binary, device=0)
    os.environ.pop('FFGPU_GF2W_BITSLICED', None)
    lb = ctx.limbs
    sets = []
    for _ in range(3):
        x = torch.randint(0, 2**62, (3, n, lb) if lb else (3, n), dtype=torch.int64, device='cuda:0', generator=gen) if ctx.elem_bytes != 12 else None
        rows = [DevArray(ctx, x[i], n) for i in range(3)]
        if not binary:
            for r in rows[:2]:
                ctx.reduce(r, out=r)
        sets.append(rows)
    ms = bench.time_launches(lambda s: ctx.mul(s[0], s[1], out=s[2]), sets, 5)
    print(f'{name:20s} mul {ms*1e3:8.1f} us  {3*ctx.elem_bytes*n/ms/1e6:7.0f} GB/s = {3*ctx.elem_bytes*n/ms/1e6/8000:.3f}')
    del sets
    torch.cuda.empty_cache()
