"""Randomised cross-check of HostView's device-resident integer algebra (mpyc_amd/finfields.py) against NumPy on object
arrays of Python integers, on the CPU context: random expression chains over the operators and NumPy functions the
runtime's protocols apply to `.value` (+, -, *, <<, **2, %, &, sum, reshape, transpose, slicing, where, vstack, cumsum,
comparisons).  After every step: the residues a view stands for (`F.array(view)`) are the true integers mod p, and the
integers it yields when materialised (`_real()`, the exact fallback) ARE the true integers -- whatever mixture of device
steps and fallbacks produced them."""
import random

import numpy as np
import pytest

PRIMES = [2**61 - 1, 2**64 - 59, 2**80 - 65, 2**127 - 1, 2**31 - 1, 2**128 - 173]


def _chain(F, seed, steps=14):
    from mpyc_amd.finfields import HostView
    p = F.modulus
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    shape = rng.choice([(24,), (6, 4), (4, 6), (2, 3, 4)])
    n = int(np.prod(shape))

    def fresh(sh=None):
        sh = sh or shape
        k = int(np.prod(sh))
        t = np.array([rng.randrange(p) for _ in range(k)], dtype=object).reshape(sh)
        return HostView(F.array(t), lazy=True), t

    def check(v, t, what):
        assert tuple(np.shape(v)) == tuple(np.shape(t)), (what, np.shape(v), np.shape(t))
        if isinstance(v, HostView):
            res = np.asarray(F.array(v).value, dtype=object)
            assert (res == (t % p)).all(), ('residues', what, seed)
            real = v._real() if v._is_lazy else np.asarray(v)
            assert (np.asarray(real, dtype=object) == t).all(), ('integers', what, seed)
        else:
            assert (np.asarray(v, dtype=object) == t).all(), ('fallback', what, seed)

    v, t = fresh()
    for step in range(steps):
        op = rng.choice(['add_view', 'sub_view', 'mul_view', 'add_int', 'rsub_int', 'mul_int', 'shl', 'shl_vec', 'sq', 'modp', 'and',
                         'sum', 'reshape', 'T', 'slice', 'where', 'vstack', 'cumsum', 'neq', 'mod2', 'eq_int', 'rshift_outer'])
        if not isinstance(v, HostView):
            v, t = fresh(np.shape(t) if np.ndim(t) else shape)           # a fallback ended the chain: start a new one
        big = max(int(abs(x)).bit_length() for x in np.asarray(t, dtype=object).reshape(-1)) if np.size(t) else 0
        if big > 900:
            v, t = fresh(np.shape(t))
        if op in ('add_view', 'sub_view', 'mul_view'):
            w, u = fresh(np.shape(t))
            v, t = {'add_view': (v + w, t + u), 'sub_view': (v - w, t - u), 'mul_view': (v * w, t * u)}[op]
        elif op == 'add_int':
            c = rng.randrange(1 << 70)
            v, t = v + c, t + c
        elif op == 'rsub_int':
            c = rng.randrange(1 << 40)
            v, t = c - v, c - t
        elif op == 'mul_int':
            c = rng.choice([2, 3, (p + 1) >> 1, rng.randrange(1 << 33)])
            v, t = v * c, t * c
        elif op == 'shl':
            k = rng.randrange(0, 40)
            v, t = v << k, t << k
        elif op == 'shl_vec' and np.ndim(t) >= 1:
            ks = nrng.integers(0, 20, size=np.shape(t)[-1])
            v, t = v << ks, t << ks
        elif op == 'sq':
            v, t = v ** 2, t ** 2
        elif op == 'modp':
            v, t = v % p, t % p
        elif op == 'and':
            m = (1 << rng.randrange(1, 20)) - 1
            v, t = v & m, t & m
        elif op == 'mod2':
            l = rng.randrange(1, 30)
            v, t = v % (1 << l), t % (1 << l)
        elif op == 'sum' and np.ndim(t) >= 2:
            ax = rng.randrange(np.ndim(t))
            v, t = np.sum(v, axis=ax), np.sum(t, axis=ax)
        elif op == 'reshape':
            sh = rng.choice([s for s in [(24,), (6, 4), (4, 6), (2, 12), (2, 3, 4), (3, 8)] if int(np.prod(s)) == np.size(t)] or [np.shape(t)])
            v, t = v.reshape(sh), t.reshape(sh)
        elif op == 'T' and np.ndim(t) == 2:
            v, t = v.T, t.T
        elif op == 'slice' and np.size(t) >= 4:
            sl = (slice(rng.randrange(0, 2), None, rng.choice([1, 2])),)
            v, t = v[sl], t[sl]
        elif op == 'where':
            mask = nrng.integers(0, 2, size=np.shape(t)).astype(bool)
            w, u = fresh(np.shape(t))
            v, t = np.where(mask, v, w), np.where(mask, t, u)
        elif op == 'vstack' and np.ndim(t) == 2:
            w, u = fresh(np.shape(t))
            v, t = np.vstack((v, w)), np.vstack((t, u))
        elif op == 'cumsum' and np.ndim(t) >= 1:
            ax = rng.randrange(np.ndim(t))
            v, t = np.cumsum(v, axis=ax), np.cumsum(t, axis=ax)
        elif op == 'eq_int':
            c = rng.choice([-1, p, p + 1, 0, int(np.asarray(t, dtype=object).reshape(-1)[0])])
            assert (np.asarray(v == c) == (t == c)).all() and (np.asarray(v != c) == (t != c)).all(), ('eq_int', c, seed)
            continue
        elif op == 'rshift_outer' and np.ndim(t) == 1:
            u = t % p
            w = v % p
            sh = np.arange(rng.randrange(1, 6))
            ov, ot = np.right_shift.outer(w, sh), np.right_shift.outer(u, sh)
            kind = rng.choice(['and1', 'and3', 'T_and1', 'add', 'index', 'eq'])
            got, want = {'and1': lambda: (ov & 1, ot & 1), 'and3': lambda: (ov & 3, ot & 3), 'T_and1': lambda: (ov.T & 1, ot.T & 1),
                         'add': lambda: (ov + 5, ot + 5), 'index': lambda: (ov[1:], ot[1:]), 'eq': lambda: (ov == 0, ot == 0)}[kind]()
            got = got._real() if isinstance(got, HostView) and got._is_lazy else np.asarray(got)
            assert np.shape(got) == np.shape(want) and (np.asarray(got, dtype=object) == want).all(), ('rshift_outer', kind, seed)
            continue
        elif op == 'neq':
            got, want = (v != 0), (t != 0)
            assert (np.asarray(got) == want).all(), ('neq', seed)
            assert np.count_nonzero(v) == np.count_nonzero(t)
            continue
        else:
            continue
        check(v, t, (step, op))
        if np.size(t) == 0 or np.size(t) > 4000:
            v, t = fresh()


@pytest.mark.parametrize('p', PRIMES)
def test_random_expression_chains_host_logic(p, monkeypatch):
    from cpuctx import use_cpu_contexts
    import mpyc_amd.finfields as gff
    use_cpu_contexts(monkeypatch)
    monkeypatch.setattr(gff, '_ctx_cache', {})
    gff._pGF.cache_clear()
    try:
        F = gff.GF(p)
        for seed in range(40):
            _chain(F, 1000 * (p % 97) + seed)
    finally:
        gff._pGF.cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize('p', PRIMES)
def test_random_expression_chains_on_gpu(p):
    """The same random chains with the views backed by device arrays: every device-resident branch of HostView (the
    code behind np_trunc / np_to_bits / np_random_bits / np_sgn / np_lsb of the runtime) against NumPy on Python ints."""
    import mpyc_amd.finfields as gff
    F = gff.GF(p)
    for seed in range(25):
        _chain(F, 7000 + seed)
