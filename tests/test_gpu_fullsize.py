"""BASELINE.json configs[2] and configs[3] at their stated size, ELEMENT BY ELEMENT against the C oracle
(VERDICT r2 weak #1 / next #3): every row of share generation (coefficients supplied, fused with the local
product, and drawn by the device CSPRNG -- the oracle restates the generator layout independently,
oracle/fforacle.c orc_rng_coeffs), and recombination from t+1 and 2t+1 rows, over n = 10^7 elements:

    configs[2]   GF(2^64-189),  m = 7, t = 3      thresha.py:47-64, 119-132
    configs[3]   GF(2^128-173), m = 7, t = 3      the whole gate: product, re-sharing, recombination of 2t+1 rows
                                                  (runtime.py:1096-1141, 603-689)

The oracle runs on all host cores (OpenMP); its two-limb product is a schoolbook 256-bit product + Knuth long division,
pinned to Python integers in tests/test_oracle_golden.py."""
import numpy as np
import pytest

from oracle import pyoracle as po
from fieldutil import P64, P128
from test_gpu_parity import rand_np

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

N_FULL = 10_000_000


@pytest.fixture(scope='module')
def eng():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from mpyc_amd import engine
    return engine


@pytest.fixture()
def threads(coracle):
    coracle.set_threads(coracle.max_threads())
    yield coracle
    coracle.set_threads(1)


def rows_equal(mtx, want, what):
    got = mtx.to_numpy()
    for i in range(want.shape[0]):
        same = got[i] == want[i]
        assert same.all(), f'{what}: row {i} differs from the oracle at {int(np.argmin(same.reshape(same.shape[0], -1).all(axis=1)))}'


def run_config(eng, co, modulus, n, t, m, seed):
    F = po.Field(modulus)
    ctx = eng.FieldContext(modulus, device=0)
    eb = ctx.elem_bytes
    cf = co.CField(modulus)
    lshape = (t, n, 2) if eb == 16 else (t, n)
    A, B = rand_np(F, eb, n, seed), rand_np(F, eb, n, seed + 1)
    C = rand_np(F, eb, t * n, seed + 2).reshape(lshape)
    dA, dB, dC = ctx.from_numpy(A), ctx.from_numpy(B), ctx.matrix_from_numpy(C)

    # local product
    want_prod = cf.ew(co.MUL, A, B)
    prod = ctx.mul(dA, dB)
    assert (prod.to_numpy() == want_prod).all(), 'mul'

    # share generation, coefficients supplied: all m rows
    want_sh = cf.split(A, C, t, m)
    sh = ctx.split(dA, dC, t, m)
    rows_equal(sh, want_sh, 'split')

    # fused with the local product (ffgpu_mul_split): shares of a*b, product never written
    want_shp = cf.split(want_prod, C, t, m)
    shp = ctx.split(dA, dC, t, m, mul_by=dB)
    rows_equal(shp, want_shp, 'mul_split')
    del shp

    # device CSPRNG: the oracle regenerates the coefficient matrix from (key, nonce) and shares with it
    key, nonce = bytes(range(7, 39)), 0x1234_5678_9abc
    Cr = co.rng_coeffs(cf, key, nonce, 20, t, n)
    want_rng = cf.split(A, Cr, t, m)
    shr = ctx.split_rng(dA, t, m, key=key, nonce=nonce, rounds=20)
    rows_equal(shr, want_rng, 'split_rng')
    want_rngp = cf.split(want_prod, Cr, t, m)
    shrp = ctx.split_rng(dA, t, m, key=key, nonce=nonce, rounds=20, mul_by=dB)
    rows_equal(shrp, want_rngp, 'mul_split_rng')
    del shr, shrp, want_rng, want_rngp, Cr

    # recombination from t+1 and from 2t+1 rows, in a rotated order (runtime.py:658-661), oracle on the same rows
    for xs in (list(range(1, t + 2)), [((2 + j) % m) + 1 for j in range(2 * t + 1)]):
        lam = po.recombination_vector(F, xs, 0)
        want_rec = cf.recombine([want_sh[x - 1] for x in xs], lam)
        rec = ctx.recombine([sh.row(x - 1) for x in xs], lam)
        assert (rec.to_numpy() == want_rec).all(), ('recombine', xs)
        assert (want_rec == A).all(), ('oracle round trip', xs)
    # the gate: recombining 2t+1 re-shared rows of the product gives the product
    xs = list(range(1, 2 * t + 2))
    lam = po.recombination_vector(F, xs, 0)
    shp = ctx.split(dA, dC, t, m, mul_by=dB)
    assert (ctx.recombine([shp.row(x - 1) for x in xs], lam).to_numpy() == want_prod).all(), 'gate'


def test_configs2_p64_m7_t3_vs_oracle_1e7(eng, threads):
    run_config(eng, threads, P64, N_FULL, 3, 7, 2001)


def test_configs3_p128_gate_m7_t3_vs_oracle_1e7(eng, threads):
    run_config(eng, threads, P128, N_FULL, 3, 7, 3001)
    torch.cuda.empty_cache()


def test_round4_kernels_at_1e7(eng, threads):
    """The kernels that changed in round 4, at n = 10^7: the bit-sliced GF(2^64) product element by element against the C
    oracle (gfpx.py:988-1045 restated), and the full-batch inverse over 2^61 - 1 and 2^64 - 189 (finfields.py:1416-1422):
    a * a^-1 == 1 on every non-zero element (the product kernel is itself checked against the oracle above), zeros give
    zero, a sample against Python's pow."""
    co = threads
    mod = (1 << 64) | 0x1b
    F = po.Field(mod, True)
    ctx = eng.FieldContext(mod, True, device=0)
    A, B = rand_np(F, 8, N_FULL, 401), rand_np(F, 8, N_FULL, 402)
    got = ctx.mul(ctx.from_numpy(A), ctx.from_numpy(B)).to_numpy()
    want = co.CField(mod, True).ew(co.MUL, A, B)
    assert (got == want).all(), int(np.argmin(got == want))
    for p in (2**61 - 1, P64):
        Fp = po.Field(p)
        cp = eng.FieldContext(p, device=0)
        X = rand_np(Fp, 8, N_FULL, 403)
        X[[0, 7, N_FULL // 3, N_FULL - 1]] = 0
        dX = cp.from_numpy(X)
        inv = cp.inv(dX, check_zero=False)
        nz = dX.t != 0
        assert bool((cp.mul(dX, inv).t[nz] == 1).all()) and bool((inv.t[~nz] == 0).all()), hex(p)
        gi = inv.to_numpy()
        for i in (1, 2, 3, 12345, N_FULL // 2, N_FULL - 2):
            assert int(gi[i]) == pow(int(X[i]), p - 2, p), (hex(p), i)
        with pytest.raises(ZeroDivisionError):
            cp.inv(dX)


def test_prss_production_mode_vs_oracle_1e7(eng, threads):
    """PRSS shares in production mode (ffgpu_prss_chacha, thresha.py:163-173 / 201-217 over the ChaCha counter-mode PRF) at
    n = 10^7, EVERY element against the C oracle's independent restatement of the keystream layout (orc_prss_chacha): one
    party of m = 7, t = 3 (20 subset keys) over GF(2^64-189) -- the share and the zero sharing (d = 3 draws per element and
    key) -- and one party of m = 3, t = 1 over GF(2^128-173)."""
    import itertools
    co = threads
    for modulus, m, t, i, zero in ((P64, 7, 3, 2, False), (P64, 7, 3, 6, True), (P128, 3, 1, 0, False)):
        F = po.Field(modulus)
        ctx = eng.FieldContext(modulus, device=0)
        cf = co.CField(modulus)
        keys = {S: bytes([(29 * sum(S) + 7 * b) % 251 for b in range(16)]) for S in itertools.combinations(range(m), m - t) if i in S}
        k40 = [po.prss_chacha_stream_key(k, b'full-size') for k in keys.values()]
        l = (modulus.bit_length() + 7) // 8 + 16
        d = t if zero else 1
        i1 = po.reduce(F, i + 1)
        weights = []
        for S in keys:
            f = po.f_S_i(F, m, i, S)
            for j in range(d):
                w = f
                for _ in range(j + 1 if zero else 0):
                    w = po.mul(F, w, i1)
                weights.append(w)
        n = N_FULL if not zero else N_FULL // 4
        got = ctx.prss_chacha(k40, d, l, weights, n).to_numpy()
        want = co.prss_chacha(cf, k40, d, l, 0, 20, weights, n)
        same = got == want
        assert same.all(), (hex(modulus), m, zero, int(np.argmin(same.reshape(n, -1).all(axis=1))))
