"""A/B of the matrix-core product variants (FFGPU_MM_MFMA = 1: LDS-staged, 3: LDS-staged + software pipeline):
timing over GF(2^61-1) and bit-exact comparison of the two results (fresh contexts are not needed: the knob is
read once per process, so this script is run once per value and prints a checksum)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
out = []
for (M, K, N) in ((2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (64, 4096, 4096), (300, 1031, 257), (1000, 9000, 500)):
    A = DevArray(ctx, bench.uniform_field(gen, M * K, bench.P61, 'cuda:0'), M * K)
    B = DevArray(ctx, bench.uniform_field(gen, K * N, bench.P61, 'cuda:0'), K * N)
    C = ctx.empty(M * N)
    ms = bench.time_launches(lambda s: ctx.matmul(A, B, M, K, N, out=C), [0], 3)
    h = hashlib.sha256(C.t.cpu().numpy().tobytes()).hexdigest()[:12]
    out.append(f'{M}x{K}x{N}: {ms:.3f} ms {M*K*N/ms/1e9:.2f} TMAC/s sha {h}')
print(os.environ.get('FFGPU_MM_MFMA', 'default'), ' | '.join(out))
