"""BASELINE.json configs[0] at its stated size: thresha.random_split + recombine, LIST path, m = 3, t = 1, GF(2^61 - 1),
10^4 secrets (mpyc/thresha.py:23-44, 88-116) -- the mirror under install() beside the
reference's own functions in ONE process, on the same replayed draws.

Under install() `mpyc.thresha.random_split / recombine` are routed to the device from `mpyc_amd.list_path_min` secrets on
(default 256: n = 10^4 takes the device path); the reference's functions stay reachable as `<routed fn>.reference`.  The
reference draws with secrets.randbelow (thresha.py:37), the mirror through its `randbelow` hook: both are fed the same
random.Random(seed) stream, restarted before every call.

Checks (every mismatch ends the program with a non-zero status):
  split   reference == mirror == oracle/pyoracle.random_split, element by element, for raw-int secrets and for
          field-element secrets (thresha.py:33-36); every share is a canonical int
  open    recombine from every set of k = t+1 = 2 points and from all k = 3 points, at x_r = 0 and for x_rs = [0, 5]:
            raw-int shares:        the reference returns the UN-REDUCED sums (thresha.py:109) = oracle.recombine_unreduced
                                   exactly; the mirror returns them reduced (documented convention):
                                   mirror == reference % p, and both == the secrets after the caller's reduction
                                   (runtime.py:588,682)
            field-element shares:  both return field elements (thresha.py:110-113), compared by value and type
Prints `LIST_PATH_RESULT {...}` with SHA-256 digests of the share matrix and the opened secrets and with timings
(secrets/s of split + recombine(k=2), reference vs mirror; bench.py's `list_path_p61_1e4_m3t1` row).

    LP_MODE=gpu|cpuctx  LP_N=10000  LP_M=3  LP_T=1  LP_PRIME=2305843009213693951  LP_SEED=5  LP_REPS=3
"""
import hashlib
import itertools
import json
import os
import random
import sys
import time

MODE = os.environ.get('LP_MODE', 'gpu')
HERE = os.path.dirname(os.path.abspath(__file__))
for p_ in (HERE, os.path.dirname(HERE)):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import mpyc_amd                                    # noqa: E402
mpyc_amd.install()
if MODE == 'cpuctx':
    from cpuctx import use_cpu_contexts
    use_cpu_contexts()

import secrets                                     # noqa: E402
from mpyc import finfields, thresha                # noqa: E402
import mpyc_amd.thresha as gth                     # noqa: E402
from oracle import pyoracle as po                  # noqa: E402

P = int(os.environ.get('LP_PRIME', str(2**61 - 1)))
N = int(os.environ.get('LP_N', '10000'))
M = int(os.environ.get('LP_M', '3'))
T = int(os.environ.get('LP_T', '1'))
SEED = int(os.environ.get('LP_SEED', '5'))
REPS = int(os.environ.get('LP_REPS', '3'))
ORIG_RANDBELOW = secrets.randbelow


def fail(msg):
    print('LIST_PATH_FAIL ' + msg, flush=True)
    sys.exit(1)


def replay(seed):
    """both randomness hooks on one fresh stream; returns the list that records every draw"""
    rnd = random.Random(seed)
    taken = []

    def draw(order):
        v = rnd.randrange(order)
        taken.append(v)
        return v
    secrets.randbelow = draw
    gth.randbelow = draw
    return taken


def digest(rows):
    h = hashlib.sha256()
    for row in rows:
        for v in row:
            h.update(int(v).to_bytes(16, 'little'))
    return h.hexdigest()


def main():
    F = finfields.GF(P)
    if not issubclass(F.array, mpyc_amd.finfields.FieldArray):
        fail('install() did not put the device array type under this field')
    split_dev, rec_dev = thresha.random_split, thresha.recombine
    split_ref, rec_ref = split_dev.reference, rec_dev.reference
    if N < mpyc_amd.list_path_min:
        fail(f'n = {N} is below list_path_min = {mpyc_amd.list_path_min}: the device path would not be taken')
    OF = po.Field(P)
    rng = random.Random(SEED)
    s_int = [rng.randrange(P) for _ in range(N)]
    for j, v in enumerate((0, 1, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2)):       # edge set of SURVEY 8(d)
        s_int[j] = v % P
    s_fld = [F(v) for v in s_int]
    out = {'mode': MODE, 'n': N, 'm': M, 't': T, 'prime_bits': P.bit_length(), 'list_path_min': mpyc_amd.list_path_min}

    # ---- share generation ----------------------------------------------------------------------------------------------
    shares = None
    for kind, s in (('int', s_int), ('field', s_fld)):
        d_ref = replay(SEED + 1)
        a = split_ref(F, s, T, M)
        d_ref = list(d_ref)
        d_dev = replay(SEED + 1)
        b = split_dev(F, s, T, M)
        if list(d_dev) != d_ref or len(d_ref) != T * N:
            fail(f'split[{kind}]: the mirror drew {len(d_dev)} values, the reference {len(d_ref)} (or in another order)')
        o = po.random_split(OF, s_int, T, M, d_ref)
        if len(a) != M or len(b) != M:
            fail(f'split[{kind}]: {len(a)} / {len(b)} rows')
        for i in range(M):
            if not (list(a[i]) == list(b[i]) == o[i]):
                bad = next(h for h in range(N) if not (a[i][h] == b[i][h] == o[i][h]))
                fail(f'split[{kind}] row {i} differs at {bad}: reference {a[i][bad]} mirror {b[i][bad]} oracle {o[i][bad]}')
            if any(type(v) is not int or not 0 <= v < P for v in b[i]):
                fail(f'split[{kind}] row {i}: mirror returned a non-canonical or non-int share')
        shares = a
    out['split_digest'] = digest(shares)

    # ---- recombination ---------------------------------------------------------------------------------------------------
    subsets = [c for k in (T + 1, M) for c in itertools.combinations(range(M), k)]
    subsets.append(tuple(reversed(range(M))))                       # another x order: the vector follows xs (appendix A.3)
    opened = None
    for sub in subsets:
        pts_int = [(i + 1, shares[i]) for i in sub]
        pts_fld = [(i + 1, [F(v) for v in shares[i]]) for i in sub]
        for x_rs in (0, [0, 5]):
            r_ref, r_dev = rec_ref(F, pts_int, x_rs), rec_dev(F, pts_int, x_rs)
            rows_ref = r_ref if isinstance(x_rs, list) else [r_ref]
            rows_dev = r_dev if isinstance(x_rs, list) else [r_dev]
            xr = x_rs if isinstance(x_rs, list) else [x_rs]
            if len(rows_ref) != len(rows_dev) or len(rows_dev) != len(xr):
                fail(f'recombine{sub} x_rs={x_rs}: shapes differ')
            for w, (ra, rb) in enumerate(zip(rows_ref, rows_dev)):
                ou = po.recombine_unreduced(OF, pts_int, xr[w])
                if list(ra) != ou:
                    fail(f'recombine{sub} at {xr[w]}: oracle.recombine_unreduced differs from the reference')
                if any(type(v) is not int or not 0 <= v < P for v in rb):
                    fail(f'recombine{sub} at {xr[w]}: the mirror returned a non-canonical value')
                if [u % P for u in ra] != list(rb):
                    bad = next(h for h in range(N) if ra[h] % P != rb[h])
                    fail(f'recombine{sub} at {xr[w]} differs at {bad}: reference {ra[bad]} (mod p {ra[bad] % P}) mirror {rb[bad]}')
            if x_rs == 0:
                if list(r_dev) != s_int:
                    fail(f'recombine{sub}: the secrets did not come back')
                opened = r_dev
            f_ref, f_dev = rec_ref(F, pts_fld, x_rs), rec_dev(F, pts_fld, x_rs)
            rows_ref = f_ref if isinstance(x_rs, list) else [f_ref]
            rows_dev = f_dev if isinstance(x_rs, list) else [f_dev]
            for ra, rb in zip(rows_ref, rows_dev):
                if any(type(v) is not F for v in rb[:8]) or any(type(v) is not F for v in ra[:8]):
                    fail(f'recombine{sub}: field-element shares must give field elements (thresha.py:110-113)')
                if [int(v.value) for v in ra] != [int(v.value) for v in rb]:
                    fail(f'recombine{sub} x_rs={x_rs} on field elements differs')
    out['opened_digest'] = digest([opened])
    out['subsets_checked'] = len(subsets)

    # ---- timings: split + recombine from t+1 points, live randomness on both sides -------------------------------------
    secrets.randbelow = ORIG_RANDBELOW                         # reference: the OS CSPRNG again (thresha.py:37)
    gth.randbelow = None                                       # mirror: device CSPRNG (production mode)
    sync = None
    if MODE == 'gpu':
        import torch
        sync = torch.cuda.synchronize

    def timed(split, rec):
        best = None
        for _ in range(REPS):
            t0 = time.perf_counter()
            sh = split(F, s_int, T, M)
            y = rec(F, [(i + 1, sh[i]) for i in range(T + 1)])
            if sync is not None:
                sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            if [v % P for v in y] != s_int:
                fail('timed round trip did not return the secrets')
        return best
    t_ref, t_dev = timed(split_ref, rec_ref), timed(split_dev, rec_dev)
    out.update(reference_s=t_ref, mirror_s=t_dev, reference_secrets_per_s=N / t_ref, mirror_secrets_per_s=N / t_dev)
    print('LIST_PATH_RESULT ' + json.dumps(out), flush=True)


main()
