import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
M = K = 4096
As = [DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K) for _ in range(3)]
for N in (1, 2, 3, 4, 8):
    B = DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N)
    C = ctx.empty(M * N)
    ms = bench.time_launches(lambda a: ctx.matmul(a, B, M, K, N, out=C), As, 3)
    print(N, '%.1f us' % (ms * 1e3))
