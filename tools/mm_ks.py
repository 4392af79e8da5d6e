"""64 x 4096 x 4096 (and two neighbours) over GF(2^61-1) for different split-K targets (FFGPU_MM_KS = workgroups aimed at)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for (M, K, N) in ((64, 4096, 4096), (128, 4096, 4096), (64, 8192, 2048), (256, 4096, 1024)):
    A = DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K)
    Bs = [DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N) for _ in range(3)]
    C = ctx.empty(M * N)
    ref = None
    for target in (256, 384, 512, 768, 1024, 1536):
        os.environ['FFGPU_MM_KS'] = str(target)
        ms = min(bench.time_launches(lambda b: ctx.matmul(A, b, M, K, N, out=C), Bs, 5) for _ in range(2))
        ctx.matmul(A, Bs[0], M, K, N, out=C)
        if ref is None:
            ref = C.t.clone()
        print('%dx%dx%d target %4d workgroups: %.1f us  same %s' % (M, K, N, target, ms * 1e3, torch.equal(ref, C.t)), flush=True)
