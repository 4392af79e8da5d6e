"""Where does the matrix-core product start to pay?  Same shapes with FFGPU_MM_MFMA=1 (run twice: set the env outside)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
print('FFGPU_MM_MFMA =', os.environ.get('FFGPU_MM_MFMA', '1'))
for (M, K, N) in ((64, 64, 64), (128, 128, 128), (200, 200, 200), (256, 256, 256), (384, 384, 384), (512, 512, 512), (128, 1024, 128),
                  (64, 4096, 64), (1024, 64, 1024), (96, 512, 96)):
    A = DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K)
    B = DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N)
    C = ctx.empty(M * N)
    ms = bench.time_launches(lambda s: ctx.matmul(A, B, M, K, N, out=C), [0], 20)
    print(f'{M}x{K}x{N}: {ms*1e3:8.1f} us  {M*K*N/ms/1e9:6.3f} TMAC/s')
