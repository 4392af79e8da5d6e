"""A/B of library builds: time the GF(2^128) product (and any other row named) with the .so given as argv[1].
usage: lib_variant_probe.py path/to/libffgpu_variant.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpyc_amd._ffi as _ffi
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
from mpyc_amd.engine import FieldContext, DevArray
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
n = 10_000_000
ctx = FieldContext((1 << 128) | 0x87, binary=True, device=0)
sets = []
for _ in range(3):
    x = torch.randint(-2**63, 2**63 - 1, (3, n, 2), dtype=torch.int64, device='cuda:0', generator=gen)
    sets.append([DevArray(ctx, x[i], n) for i in range(3)])
best = []
for rep in range(3):
    ms = bench.time_launches(lambda s: ctx.mul(s[0], s[1], out=s[2]), sets, 10)
    best.append(ms)
print(os.path.basename(_ffi.LIB_PATH), 'gf2_128 mul us:', ' '.join(f'{m*1e3:.1f}' for m in best), f'-> {48*n/min(best)/1e6/8000:.3f} of HBM')
if os.environ.get('VARIANT_SPLIT'):
    P64 = 2**64 - 189
    c64 = FieldContext(P64, device=0)
    s64 = [bench.StepData(c64, n, 3, 7, gen) for _ in range(3)]
    key = bytes(range(32))
    for rounds in (20, 8):
        best = [bench.time_launches(lambda s: c64.split_rng(s.a, 3, 7, key=key, nonce=7, rounds=rounds, out=s.shares), s64, 10) for _ in range(3)]
        print(os.path.basename(_ffi.LIB_PATH), f'split_rng_p64_m7t3 chacha{rounds} us:', ' '.join(f'{m*1e3:.1f}' for m in best))
if os.environ.get('VARIANT_PRSS'):
    import itertools
    from mpyc_amd import finfields as gff, thresha as gth
    F61 = gff.GF(2**61 - 1)
    gth.prss_prf, gth.prss_allow_chacha8 = 'chacha', True
    for (mm, ii), rr in itertools.product(((7, 2), (3, 0)), (20, 8)):
        keys = {S: bytes([sum(S) % 256]) * 16 for S in itertools.combinations(range(mm), mm - (mm - 1) // 2) if ii in S}
        prfs = {S: gth.PRF(k_, F61.order) for S, k_ in keys.items()}
        gth.prss_rounds = rr
        best = [bench.time_launches(lambda s: gth.np_pseudorandom_share(F61, mm, ii, prfs, b'uci', n), [0], 5) for _ in range(3)]
        print(os.path.basename(_ffi.LIB_PATH), f'prss m={mm} keys={len(keys)} chacha{rr} ms:', ' '.join(f'{m:.3f}' for m in best))
if os.environ.get('VARIANT_SMALLGF'):
    from mpyc_amd.gfpx import BinaryPolynomial
    for deg in (9, 16, 24, 32):
        mod = int(BinaryPolynomial.next_irreducible(1 << deg))
        cb = FieldContext(mod, binary=True, device=0)
        ss = []
        for _ in range(3):
            if cb.elem_bytes == 4:          # (round 6: 4-byte storage for 9 <= n <= 32)
                x = torch.randint(-2**31, 2**31 - 1, (3, n), dtype=torch.int32, device='cuda:0', generator=gen)
            else:
                x = torch.randint(0, 1 << deg, (3, n), dtype=torch.int64, device='cuda:0', generator=gen)
            rows_ = [DevArray(cb, x[i], n) for i in range(3)]
            for r_ in rows_[:2]:
                cb.reduce(r_, out=r_)
            ss.append(rows_)
        best = [bench.time_launches(lambda s: cb.mul(s[0], s[1], out=s[2]), ss, 10) for _ in range(3)]
        print(os.path.basename(_ffi.LIB_PATH), f'gf2_{deg} (modulus {hex(mod)}) mul us:', ' '.join(f'{m*1e3:.1f}' for m in best), f'-> {3*cb.elem_bytes*n/min(best)/1e6/8000:.3f} of HBM ({cb.elem_bytes}-byte storage)')
