"""secure S-box layer, all m=3 parties (t=1): one-kernel layer vs the 13-launch composition, eager and as HIP graphs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray, CapturedLaunches
from mpyc_amd import finfields as gff, gfpx as ggx, protocols
ctx = FieldContext(0x11b, binary=True, device=0)
F = gff.GF(ggx.GFpX(2)(0x11b))
r_ = [1, 0, 0, 0, 1, 1, 1, 1]
rows8 = [sum(r_[(c_ - j_) % 8] << c_ for c_ in range(8)) for j_ in range(8)]
A = [[(rows8[r] >> c) & 1 for c in range(8)] for r in range(8)]
B = [(0x63 >> r) & 1 for r in range(8)]
gen = torch.Generator(device='cuda:0'); gen.manual_seed(3)
import os as _os
for n in ([int(_os.environ["SBOX_N"])] if _os.environ.get("SBOX_N") else [10**6, 10**7, 10**8]):
    xpub = DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=gen), n)
    xs = protocols.as_matrix(ctx, protocols.share(ctx, xpub, 1, 3))
    rb = DevArray(ctx, torch.randint(0, 2, (8 * n,), dtype=torch.uint8, device='cuda:0', generator=gen), 8 * n)
    rbits = protocols.as_matrix(ctx, protocols.share(ctx, rb, 1, 3))
    want = ctx.sbox(xpub, rows8, 0x63).t
    for fused in (True, False):
        for rounds in ((20, 8) if fused else (20,)):
            st = ctx.rng_state(rounds=rounds)
            res = protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B, rng=st, fused=fused)
            ok = torch.equal(protocols.open_(ctx, F, [res.row(i) for i in range(3)], 1).t, want)
            ms = bench.time_launches(lambda s: protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B, rng=st, fused=fused), [0], 5 if n < 10**8 else 2)
            line = 'n=%d fused=%s chacha%d eager %.1f us (%.3g secure bytes/s) %s' % (n, fused, rounds, ms * 1e3, n / ms * 1e3, 'ok' if ok else 'WRONG')
            if n <= 10**7:
                cg = CapturedLaunches(lambda: protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B, rng=st, fused=fused))
                msg = bench.time_launches(lambda s: cg.replay(), [0], 20)
                line += '  graph %.1f us' % (msg * 1e3)
            print(line)
    del xs, rbits, xpub, rb
    torch.cuda.empty_cache()
