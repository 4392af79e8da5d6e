#!/usr/bin/env python3
"""The bench rows whose limiter is the VALU (in-kernel ChaCha, GF(2^n) products, exponentiations), launched in a FIXED ORDER,
three launches each, at n = 10^7.  Run under
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d DIR -o valu -- python tools/valu_probe.py
and feed the counter CSV to tools/valu_summary.py: library dispatches are consumed in order, three per row -> VALU
instructions per element for every row (profiles/r05_valu.json, read by bench.py for `valu_frac`)."""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, gfpx as ggx, thresha as gth
from mpyc_amd.finfields import find_prime_root

torch.cuda.set_device(0)
n = 10_000_000
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
P61, P64 = bench.P61, bench.P64
rows = []          # (row name, callable)
keep = []


def lagrange(modulus, xs):
    return list(gth._recombination_vector(gff.GF(modulus), tuple(xs), 0))


# ---- 2^61 - 1, m = 3, t = 1
ctx = FieldContext(P61, device=0)
t, m, k = 1, 3, 3
s = bench.StepData(ctx, n, t, m, gen)
lam = lagrange(P61, range(1, k + 1))
key = bytes(range(32))
for rounds in (20, 8):
    rows.append((f'split_rng_p61_m3t1_chacha{rounds}', lambda rounds=rounds: ctx.split_rng(s.c, t, m, key=key, nonce=3, rounds=rounds, out=s.shares)))
out_chain = ctx.empty_matrix(m, n)
for rounds in (20, 8):
    st = ctx.rng_state(rounds=rounds)
    keep.append(st)
    rows.append((f'chain_gate_p61_m3t1_chacha{rounds}', lambda st=st: ctx.gate([s.shares.row(j) for j in range(k)], lam, None, None, t, m, state=st, out=out_chain)))
rows.append(('inv_p61', lambda: ctx.inv(s.a, out=s.c, check_zero=False)))
rows.append(('sqrt_p61', lambda: ctx.pow(s.a, (P61 + 1) // 4, out=s.c)))
# ---- PRSS shares in production mode (ChaCha streams expanded in k_prss_chacha), one party of m = 7, t = 3 / m = 3, t = 1
import itertools
F61 = gff.GF(P61)
for (mm_, ii_), rr_ in itertools.product(((7, 2), (3, 0)), (20, 8)):
    keys_ = {S: bytes([sum(S) % 256]) * 16 for S in itertools.combinations(range(mm_), mm_ - (mm_ - 1) // 2) if ii_ in S}
    prfs_ = {S: gth.PRF(kk_, F61.order) for S, kk_ in keys_.items()}

    def prss_row(mm_=mm_, ii_=ii_, rr_=rr_, prfs_=prfs_):
        gth.prss_prf, gth.prss_rounds, gth.prss_allow_chacha8 = 'chacha', rr_, True
        gth.np_pseudorandom_share(F61, mm_, ii_, prfs_, b'uci', n)
        gth.prss_prf, gth.prss_rounds = 'shake', 20
    rows.append((f'prss_share_p61_m{mm_}t{(mm_ - 1) // 2}_chacha{rr_}', prss_row))
# ---- 2^64 - 189, m = 7, t = 3
ctx64 = FieldContext(P64, device=0)
t2, m2 = 3, 7
s64 = bench.StepData(ctx64, n, t2, m2, gen)
for rounds in (20, 12, 8):
    rows.append((f'split_rng_p64_m7t3_chacha{rounds}', lambda rounds=rounds: ctx64.split_rng(s64.a, t2, m2, key=key, nonce=7, rounds=rounds, out=s64.shares)))
# ---- the 80-bit prime of the default SecFxp(): the inverse square roots of np_random_bits (product chain in 27-bit digits)
P80 = find_prime_root(80)[0]
ctx80 = FieldContext(P80, device=0)
a80 = DevArray(ctx80, torch.randint(0, 2**31 - 1, (n, 3), dtype=torch.int32, device='cuda:0', generator=gen), n)
a80 = ctx80.reduce(a80, out=a80)
c80 = ctx80.empty(n)
rows.append(('inv_sqrt_p80', lambda: ctx80.pow(a80, (3 * P80 - 5) >> 2, out=c80)))
rows.append(('inv_p80', lambda: ctx80.inv(a80, out=c80, check_zero=False)))
# ---- 136-bit prime (three limbs)
P136 = find_prime_root(136)[0]
ctx136 = FieldContext(P136, device=0)
a3 = DevArray(ctx136, torch.randint(0, 2**62, (n, 3), dtype=torch.int64, device='cuda:0', generator=gen), n)
a3 = ctx136.reduce(a3, out=a3)
sh3 = ctx136.empty_matrix(m, n)
rows.append(('split_rng_p136_m3t1_chacha20', lambda: ctx136.split_rng(a3, t, m, key=key, nonce=5, out=sh3)))
# ---- GF(2^64), GF(2^128)
for label, modulus, tail, ebg in (('gf2_64', (1 << 64) | 0x1b, (), 8), ('gf2_128', (1 << 128) | 0x87, (2,), 16)):
    cb = FieldContext(modulus, binary=True, device=0)
    x = torch.randint(-2**63, 2**63 - 1, (3, n) + tail, dtype=torch.int64, device='cuda:0', generator=gen)
    bufs = tuple(DevArray(cb, x[i], n) for i in range(3))
    rows.append((f'mul_{label}', lambda cb=cb, bufs=bufs: cb.mul(bufs[0], bufs[1], out=bufs[2])))
    shb = cb.empty_matrix(m2, n)
    for j in range(m2):
        shb.row(j).t.copy_(bufs[j % 2].t)
    for kk in (4, 7):
        rg_ = random.Random(1000 + kk)
        lamd = [rg_.randrange(2, 1 << (8 * ebg)) for _ in range(kk)]        # k distinct coefficients (as bench.py)
        plan = cb.recombine_plan([shb.row(j) for j in range(kk)], lamd, bufs[2])
        rows.append((f'recombine_{label}_k{kk}_dense', plan))
    keep.append((cb, bufs, shb))
# ---- the one-kernel secure S-box layer (three parties, t = 1): per secure BYTE, 10^6 (fresh keystream per step) and 10^8 (continued)
from mpyc_amd import protocols
ctx8 = FieldContext(0x11b, binary=True, device=0)
F8 = gff.GF(ggx.GFpX(2)(0x11b))
r_ = [1, 0, 0, 0, 1, 1, 1, 1]
rows8 = [sum(r_[(c_ - j_) % 8] << c_ for c_ in range(8)) for j_ in range(8)]
A8 = [[(rows8[r] >> c) & 1 for c in range(8)] for r in range(8)]
B8 = [(0x63 >> r) & 1 for r in range(8)]
units = {}
for tag, n8 in (('1e6', 10**6), ('1e8', 10**8)):
    xs8 = ctx8.empty_matrix(3, n8); xs8.t[:, :n8].copy_(torch.randint(0, 256, (3, n8), dtype=torch.uint8, device='cuda:0', generator=gen))
    rb8 = ctx8.empty_matrix(3, 8 * n8); rb8.t[:, :8 * n8].copy_(torch.randint(0, 2, (3, 8 * n8), dtype=torch.uint8, device='cuda:0', generator=gen))
    st8 = ctx8.rng_state()
    keep.append((xs8, rb8, st8))
    rows.append((f'secure_sbox_layer_m3t1_{tag}', lambda xs8=xs8, rb8=rb8, st8=st8: protocols.sbox_layer_all(ctx8, F8, xs8, rb8, 1, A8, B8, rng=st8, fused=True)))
    units[f'secure_sbox_layer_m3t1_{tag}'] = n8
torch.cuda.synchronize()
# ---- the measured sequence: a marker copy (k_copy16) opens every row
order = []
for name, fn in rows:
    ctx.copy(s.a.t, s.y.t)
    for _ in range(3):
        fn()
    order.append(name)
torch.cuda.synchronize()
out = os.environ.get('VALU_ORDER', os.path.join(ROOT, 'gpurun_out', 'valu_order.json'))
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({'n': n, 'launches_per_row': 3, 'rows': order, 'units': units}, open(out, 'w'))
print('valu probe done:', len(order), 'rows')
