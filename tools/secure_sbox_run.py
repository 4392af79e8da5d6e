"""Run the secure AES S-box layer (mpyc_amd/protocols.py) a few times: target for rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx, protocols
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = FieldContext(0x11b, True, device=0)
F = gff.GF(gfpx.GFpX(2)(0x11b))
r_ = [1, 0, 0, 0, 1, 1, 1, 1]
rows8 = [sum(r_[(c - j) % 8] << c for c in range(8)) for j in range(8)]
A = [[(rows8[r] >> c) & 1 for c in range(8)] for r in range(8)]
B = [(0x63 >> r) & 1 for r in range(8)]
x = DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0'), n)
xs = protocols.share(ctx, x, 1, 3)
rb = DevArray(ctx, torch.randint(0, 2, (8 * n,), dtype=torch.uint8, device='cuda:0'), 8 * n)
rbits = protocols.share(ctx, rb, 1, 3)
for _ in range(reps):
    out = protocols.sbox_layer(ctx, F, xs, rbits, 1, A, B)
torch.cuda.synchronize()
assert torch.equal(protocols.open_(ctx, F, out, 1).t, ctx.sbox(x, rows8, 0x63).t)
print('ok')
