import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
rows8_ = [sum([1, 0, 0, 0, 1, 1, 1, 1][(c - j) % 8] << c for c in range(8)) for j in range(8)]
ctx8 = FieldContext(0x11b, binary=True, device=0)
rows8, b8 = rows8_, 0x63
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for n8 in (10_000_000, 1_000_000_000):
    bufs = []
    for _ in range(3 if n8 > 10**8 else 8):
        x = torch.randint(0, 256, (3, n8), dtype=torch.uint8, device='cuda:0', generator=gen)
        bufs.append((DevArray(ctx8, x[0], n8), DevArray(ctx8, x[1], n8), DevArray(ctx8, x[2], n8)))
    ms = bench.time_launches(lambda s: ctx8.mul(s[0], s[1], out=s[2]), bufs, 5)
    ms2 = bench.time_launches(lambda s: ctx8.sbox(s[0], rows8, b8, out=s[2]), bufs, 5)
    print(os.environ.get('FFGPU_TABLE_BPC'), n8, 'mul %.1f us %.0f GB/s | sbox %.1f us %.0f GB/s' % (ms*1e3, 3*n8/ms/1e6, ms2*1e3, 2*n8/ms2/1e6))
    del bufs; torch.cuda.empty_cache()
