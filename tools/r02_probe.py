"""rocprofv3 target: the all-party S-box layer (n = 1e6 and 1e8) and the skinny products, a few launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx, protocols
what = sys.argv[1] if len(sys.argv) > 1 else 'sbox,skinny'
if 'sbox' in what:
    ctx = FieldContext(0x11b, True, device=0)
    F = gff.GF(gfpx.GFpX(2)(0x11b))
    r_ = [1, 0, 0, 0, 1, 1, 1, 1]
    rows8 = [sum(r_[(c - j) % 8] << c for c in range(8)) for j in range(8)]
    A = [[(rows8[r] >> c) & 1 for c in range(8)] for r in range(8)]
    B = [(0x63 >> r) & 1 for r in range(8)]
    for n in (1_000_000, 100_000_000):
        x = DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0'), n)
        xs = protocols.as_matrix(ctx, protocols.share(ctx, x, 1, 3))
        rb = DevArray(ctx, torch.randint(0, 2, (8 * n,), dtype=torch.uint8, device='cuda:0'), 8 * n)
        rbits = protocols.as_matrix(ctx, protocols.share(ctx, rb, 1, 3))
        for _ in range(3):
            out = protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B)
        torch.cuda.synchronize()
        assert torch.equal(protocols.open_(ctx, F, [out.row(i) for i in range(3)], 1).t, ctx.sbox(x, rows8, 0x63).t)
        if n <= 10**6 and 'graph' in what:
            from mpyc_amd.engine import CapturedLaunches
            st = ctx.rng_state()
            cg = CapturedLaunches(lambda: protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B, rng=st))
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(20):
                cg.replay()
            torch.cuda.synchronize()
            print('graph replay us', (time.perf_counter() - t0) / 20 * 1e6)
            assert torch.equal(protocols.open_(ctx, F, [cg.result.row(i) for i in range(3)], 1).t, ctx.sbox(x, rows8, 0x63).t)
        del x, xs, rb, rbits, out
        torch.cuda.empty_cache()
if 'skinny' in what:
    ctx = FieldContext(bench.P61, device=0)
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
    for (M, K, N) in ((4096, 4096, 1), (1, 4096, 4096), (16384, 4096, 1), (1, 16384, 4096)):
        As = [DevArray(ctx, bench.uniform_field(gen, M * K if N == 1 else K * N, bench.P61, 'cuda:0'), max(M, N) * K) for _ in range(3)]
        v = DevArray(ctx, bench.uniform_field(gen, K, bench.P61, 'cuda:0'), K)
        C = ctx.empty(M * N)
        for _ in range(5):
            for a in As:
                if N == 1:
                    ctx.matmul(a, v, M, K, N, out=C)
                else:
                    ctx.matmul(v, a, M, K, N, out=C)
        torch.cuda.synchronize()
print('ok')
