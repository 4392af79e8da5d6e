"""Is the 10^6-byte S-box layer slow because of its launch shape or because a ~1 ms measurement never sees the clocks the
10^8-byte run sees?  The same captured layer replayed 20 / 200 / 2000 / 20000 times back to back (events on the stream),
then the 10^8 run, then the 20-replay measurement again right after it (GPU still warm)."""
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


def make(n):
    xpub = DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=gen), n)
    xs = protocols.as_matrix(ctx, protocols.share(ctx, xpub, 1, 3))
    rb = DevArray(ctx, torch.randint(0, 2, (8 * n,), dtype=torch.uint8, device='cuda:0', generator=gen), 8 * n)
    rbits = protocols.as_matrix(ctx, protocols.share(ctx, rb, 1, 3))
    return xs, rbits


xs, rbits = make(10**6)
st = ctx.rng_state(rounds=20)
cg = CapturedLaunches(lambda: protocols.sbox_layer_all(ctx, F, xs, rbits, 1, A, B, rng=st, fused=True))
for reps in (20, 200, 2000, 20000, 20):
    ms = bench.time_launches(lambda s: cg.replay(), [0], reps)
    print('n=1e6 graph replay x %5d: %.2f us per layer' % (reps, ms * 1e3), flush=True)
xl, rl = make(10**8)
ms = bench.time_launches(lambda s: protocols.sbox_layer_all(ctx, F, xl, rl, 1, A, B, rng=st, fused=True), [0], 5)
print('n=1e8 eager: %.1f us per layer = %.2f us per 1e6 bytes' % (ms * 1e3, ms * 10), flush=True)
for reps in (20, 2000):
    ms = bench.time_launches(lambda s: cg.replay(), [0], reps)
    print('n=1e6 graph replay x %5d right after: %.2f us per layer' % (reps, ms * 1e3), flush=True)
for n in (2 * 10**6, 4 * 10**6):
    xm, rm = make(n)
    ms = bench.time_launches(lambda s: protocols.sbox_layer_all(ctx, F, xm, rm, 1, A, B, rng=st, fused=True), [0], 200)
    print('n=%d eager x 200: %.1f us per layer = %.2f us per 1e6 bytes' % (n, ms * 1e3, ms * 1e9 / n), flush=True)
