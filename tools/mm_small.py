import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
ctx.set_timing(True)
for (M, K, N) in ((64, 64, 64), (256, 256, 256), (16, 16, 16)):
    A = DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K)
    B = DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N)
    C = ctx.empty(M * N)
    for _ in range(3):
        ctx.matmul(A, B, M, K, N, out=C)
    gpu = ctx.last_kernel_ms() * 1e3
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        ctx.matmul(A, B, M, K, N, out=C)
    torch.cuda.synchronize(); host = (time.perf_counter() - t) / 200 * 1e6
    print(M, K, N, 'gpu %.1f us' % gpu, 'wall per call %.1f us' % host)
x = DevArray(ctx, bench.uniform_field(gen, 4096, bench.P61, 'cuda:0'), 4096)
for _ in range(3): ctx.mul(x, x, out=x)
print('mul 4096: gpu %.1f us' % (ctx.last_kernel_ms() * 1e3))
