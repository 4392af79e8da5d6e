"""Matrix-vector and skinny products over GF(2^61-1): the np_bnnmnist shape (demos/np_bnnmnist.py:10-15, a 4096^2 matvec)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for (M, K, N) in ((4096, 4096, 1), (4096, 4096, 8), (4096, 4096, 64), (8192, 8192, 1), (1, 4096, 4096), (8, 4096, 4096), (64, 4096, 4096), (256, 4096, 4096), (1024, 1024, 1024), (2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192)):
    As = [DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K) for _ in range(3)]
    B = DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N)
    C = ctx.empty(M * N)
    ms = bench.time_launches(lambda a: ctx.matmul(a, B, M, K, N, out=C), As, 3)
    byts = 8 * (M * K + K * N + M * N)
    print(f'{M}x{K} @ {K}x{N}: {ms*1e3:8.1f} us   {byts/ms/1e6:7.0f} GB/s   {M*K*N/ms/1e9:6.3f} TMAC/s')
