import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
out = []
for (M, K, N) in ((1, 4096, 4096), (1, 16384, 4096), (4, 4096, 4096)):
    Ws = [DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N) for _ in range(3)]
    v = DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K)
    C = ctx.empty(M * N)
    ms = bench.time_launches(lambda w: ctx.matmul(v, w, M, K, N, out=C), Ws, 10)
    out.append(f'{M}x{K}x{N}: {ms*1e3:.1f} us {8*(K*N)/ms/1e6:.0f} GB/s')
    del Ws
print(' | '.join(out))
