"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (mpyc_amd/).

Times the REFERENCE ITSELF (lschoe/mpyc, pure Python) on the host cores for bench.py's `cpu_baseline` leg
(SURVEY.md section 8d "CPU baseline, same run"): the functions the GPU path replaces, called
exactly as the runtime calls them for `a * b` on a SecFld array in the default 3-party setting --

    c      = a * b                                  FiniteFieldArray.__mul__          finfields.py:1105-1112
    shares = thresha.np_random_split(F, c, t, m)    live secrets.randbelow draws      thresha.py:47-64
    y      = thresha.np_recombine(F, points)        2t+1 points                       thresha.py:119-132

-- on 1 core, and on `procs` cores as independent processes over equal slices (the reference is single-threaded;
SURVEY 8d prescribes N processes on n/N slices).  `mpyc` must be importable (PYTHONPATH holds the reference
checkout: /root/reference in the build container, the staged copy _refstage/ on the GPU box); only
mpyc.finfields and mpyc.thresha are imported (mpyc.runtime parses sys.argv and starts a runtime).
"""
import os
import sys
import time


def available(extra_paths=()):
    for p in extra_paths:
        if p and os.path.isdir(os.path.join(p, 'mpyc')) and p not in sys.path:
            sys.path.append(p)
    try:
        import mpyc.finfields  # noqa: F401
        import mpyc.thresha    # noqa: F401
        return True
    except Exception:            # noqa: BLE001 -- any import problem means "no reference here"
        return False


def one_pass(args):
    """One pass of the gate over n fresh random elements; returns (seconds per stage..., n).  Runs in a worker."""
    modulus, n, t, m, seed = args
    import numpy as np
    from mpyc import finfields, thresha
    F = finfields.GF(modulus)
    rng = np.random.default_rng(seed)
    a = F.array(rng.integers(0, modulus, size=n, dtype=np.int64).astype(object), check=False)
    b = F.array(rng.integers(0, modulus, size=n, dtype=np.int64).astype(object), check=False)
    t0 = time.perf_counter()
    c = a * b
    t1 = time.perf_counter()
    shares = thresha.np_random_split(F, c.value, t, m)          # runtime.py:643-662: x = c.value; random_split(field, x, t, m)
    t2 = time.perf_counter()
    points = [(j + 1, shares[j]) for j in range(2 * t + 1)]
    y = thresha.np_recombine(F, points)
    t3 = time.perf_counter()
    assert (y.value[:8] == c.value[:8]).all()
    return (t1 - t0, t2 - t1, t3 - t2, n)


def measure(modulus, t, m, n_one, n_each, procs, seed=20260925):
    """-> dict with field-ops/s (3 per element: one per stage) of the reference on 1 core (n_one elements) and on
    `procs` processes (n_each elements each, wall clock of the slowest)."""
    import multiprocessing as mp
    one_pass((modulus, 1000, t, m, seed))                       # warm caches (recombination vector, imports)
    s_mul, s_split, s_rec, n = one_pass((modulus, n_one, t, m, seed))
    one = {'n': n, 'mul_s': s_mul, 'split_s': s_split, 'recombine_s': s_rec,
           'field_ops_per_s': 3 * n / (s_mul + s_split + s_rec),
           'mulmod_per_s': n / s_mul, 'split_secrets_per_s': n / s_split, 'recombine_secrets_per_s': n / s_rec}
    out = {'one_core': one, 'procs': procs}
    if procs > 1:
        ctx = mp.get_context('fork')

        def run(nproc, n_per):
            with ctx.Pool(nproc) as pool:
                pool.map(one_pass, [(modulus, 1000, t, m, seed + 1 + i) for i in range(nproc)])       # start + warm every worker
                t0 = time.perf_counter()
                res = pool.map(one_pass, [(modulus, n_per, t, m, seed + 1000 + i) for i in range(nproc)], chunksize=1)
                wall = time.perf_counter() - t0
            total = sum(r[3] for r in res)
            return {'procs': nproc, 'n_total': total, 'n_each': n_per, 'wall_s': wall, 'field_ops_per_s': 3 * total / wall}
        # the reference does not scale to every logical CPU (its coefficient draws go through the kernel CSPRNG): pick the
        # process count that gives it the HIGHEST throughput on a short probe, then measure at that count
        cands = sorted({c for c in (procs, procs // 2, procs // 4, procs // 8) if c >= 2})
        probe = [run(c, max(20_000, n_each // 4)) for c in cands] if len(cands) > 1 else []
        best = max(probe, key=lambda r: r['field_ops_per_s'])['procs'] if probe else procs
        out['all_cores'] = run(best, n_each)
        out['procs'] = best
        out['probe'] = [{'procs': r['procs'], 'field_ops_per_s': round(r['field_ops_per_s'], 1)} for r in probe]
    return out


if __name__ == '__main__':
    # python oracle/refbaseline.py <reference checkout> [n_one n_each procs]  -> one JSON line (bench.py runs it as a bounded
    # sub-process: a worker pool that loses a worker would otherwise hang the run)
    import json
    ok = available(sys.argv[1:2])
    if len(sys.argv) >= 5:
        n_one, n_each, procs = (int(v) for v in sys.argv[2:5])
    else:
        n_one, n_each, procs = 200_000, 50_000, os.cpu_count() or 1
    print('REFBASELINE ' + json.dumps(measure(2**61 - 1, 1, 3, n_one, n_each, procs) if ok else {'error': 'mpyc not importable'}))
