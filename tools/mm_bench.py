import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for dim in (2048, 4096):
    A = DevArray(ctx, bench.uniform_field(gen, dim*dim, bench.P61, 'cuda:0'), dim*dim)
    B = DevArray(ctx, bench.uniform_field(gen, dim*dim, bench.P61, 'cuda:0'), dim*dim)
    C = ctx.empty(dim*dim)
    ms = bench.time_launches(lambda s: ctx.matmul(A, B, dim, dim, dim, out=C), [0], 2)
    print(os.environ.get('FFGPU_MM_TILE'), dim, '%.2f ms  %.2f TMAC/s' % (ms, dim**3/ms/1e9))
