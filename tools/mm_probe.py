"""One 4096^3 product over GF(2^61-1), a few times: target for rocprofv3 --pmc (MFMA utilisation of k_limb_gemm_glds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
ctx = FieldContext(bench.P61, device=0)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
d = 4096
A = DevArray(ctx, bench.uniform_field(gen, d * d, bench.P61, 'cuda:0'), d * d)
B = DevArray(ctx, bench.uniform_field(gen, d * d, bench.P61, 'cuda:0'), d * d)
C = ctx.empty(d * d)
for _ in range(3):
    ctx.matmul(A, B, d, d, d, out=C)
torch.cuda.synchronize()
print('done')
