"""Skinny products (M <= 8 rows against a big matrix): time per call over shapes, over 2^61 - 1, on rotating weights so
that nothing is served from the caches.  `python tools/vecmat_probe.py` on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray

def run(mod, shapes, label):
    ctx = FieldContext(mod, device=0)
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
    eb = ctx.elem_bytes
    for (M, K, N) in shapes:
        nb = max(2, min(8, int(1.5e9 // (K * N * eb))))
        Bs = [DevArray(ctx, bench.uniform_field(gen, K * N, mod, 'cuda:0'), K * N) for _ in range(nb)]
        A = DevArray(ctx, bench.uniform_field(gen, M * K, mod, 'cuda:0'), M * K)
        C = ctx.empty(M * N)
        for i in range(5):
            ctx.matmul(A, Bs[i % nb], M, K, N, out=C)
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for i in range(reps):
                ctx.matmul(A, Bs[i % nb], M, K, N, out=C)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        bytes_ = (K * N + M * K + M * N) * eb
        print(f'{label} {M}x{K}x{N}: {best:.1f} us  frac {bytes_ / best / 1e6 / 8:.3f}', end='; ' if os.environ.get('VECMAT_MORE', '1') != '1' else '\n', flush=True)

shapes = os.environ.get('VECMAT_SHAPES')
run(bench.P61, [tuple(int(v) for v in t.split('x')) for t in shapes.split(',')] if shapes else [(m, 4096, 4096) for m in (1, 2, 3, 4, 5, 6, 8)] + ([(1, 1024, 8192), (1, 8192, 2048), (8, 8192, 2048), (1, 16384, 16384)] if os.environ.get('VECMAT_MORE', '1') == '1' else []), 'p61')
