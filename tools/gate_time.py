"""Isolate the chain-gate kernel cost: host key vs device state, plain split_rng vs gate, several fields."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpyc_amd.engine import FieldContext, DevArray
from mpyc_amd import finfields as gff, thresha as gth
n = 10_000_000
gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
for P in (2**61 - 1, 2**64 - 189):
    ctx = FieldContext(P, device=0)
    t, m, k = 1, 3, 3
    lam = [int(v) for v in gth._recombination_vector(gff.GF(P), (1, 2, 3), 0)]
    sets = [bench.StepData(ctx, n, t, m, gen) for _ in range(3)]
    outs = [ctx.empty_matrix(m, n) for _ in range(3)]
    key = bytes(32)
    st = ctx.rng_state(key=key)
    pairs = list(zip(sets, outs))
    r = {}
    r['split_rng key'] = bench.time_launches(lambda p: ctx.split_rng(p[0].a, t, m, key=key, nonce=1, out=p[1]), pairs, 5)
    r['split_rng state'] = bench.time_launches(lambda p: ctx.split_rng(p[0].a, t, m, state=st, out=p[1]), pairs, 5)
    r['mul_split_rng key'] = bench.time_launches(lambda p: ctx.split_rng(p[0].a, t, m, key=key, nonce=1, out=p[1], mul_by=p[0].b), pairs, 5)
    r['gate 1row x 1row key'] = bench.time_launches(lambda p: ctx.gate([p[0].a], [1], [p[0].b], [1], t, m, key=key, nonce=1, out=p[1]), pairs, 5)
    r['gate 3rows square key'] = bench.time_launches(lambda p: ctx.gate([p[0].shares.row(j) for j in range(k)], lam, None, None, t, m, key=key, nonce=1, out=p[1]), pairs, 5)
    r['gate 3rows square state'] = bench.time_launches(lambda p: ctx.gate([p[0].shares.row(j) for j in range(k)], lam, None, None, t, m, state=st, out=p[1]), pairs, 5)
    print(hex(P), '  '.join(f'{a}={b*1e3:.0f}us' for a, b in r.items()))
    del sets, outs, pairs
    torch.cuda.empty_cache()
