"""PRSS share generation on the device: production mode (ChaCha streams expanded in k_prss_chacha) vs parity mode (host SHAKE128
+ k_prss), one party of m = 3, t = 1 (2 subset keys) and of m = 7, t = 3 (20 subset keys), GF(2^61 - 1) and GF(2^128 - 173)."""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpyc_amd import finfields as gff, thresha as gth

N = int(os.environ.get('PRSS_N', '10000000'))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for mod in ((2**61 - 1,) if os.environ.get('PRSS_ONLY61') else (2**61 - 1, 2**128 - 173)):
    F = gff.GF(mod)
    for m, t, i in ((3, 1, 0), (7, 3, 2)):
        keys = {S: bytes([sum(S) % 256]) * 16 for S in itertools.combinations(range(m), m - t) if i in S}
        prfs = {S: gth.PRF(k, F.order) for S, k in keys.items()}
        ctx = gff._context(F)
        ctx.set_timing(True, accumulate=True)
        for mode, rounds, n in (('chacha', 20, N), ('chacha', 12, N), ('chacha', 8, N), ('shake', 0, min(N, 2_000_000))):
            gth.prss_prf, gth.prss_allow_chacha8 = mode, True
            if rounds:
                gth.prss_rounds = rounds
            ctx.busy_ms()
            dt = timed(lambda: gth.np_pseudorandom_share(F, m, i, prfs, b'uci', n))
            busy, calls = ctx.busy_ms()
            dt0 = timed(lambda: gth.np_pseudorandom_share_0(F, m, i, prfs, b'uci', n)) if mode == 'chacha' else float('nan')
            print(f'bits={mod.bit_length()} m={m} t={t} keys={len(keys)} mode={mode}{rounds or ""} n={n}: share {dt*1e3:.3f} ms '
                  f'({n/dt:.3e} shares/s; kernel {busy/4/max(1,calls//4)*1:.3f} ms/launch over {calls} launches), zero-share {dt0*1e3:.3f} ms', flush=True)
