"""Three-limb prime fields (129..192 bits: PM192 for p = 2^k - c -- the default fields of SecInt(97..160) --, MONT192
for every other odd prime, e.g. the root-of-unity field of demos/np_lpsolver.py's largest dataset) through the C ABI
and the mirror API, against Python integers (oracle/pyoracle.py; the C oracle stops at 128 bits) and the reference's
vectors (tests/golden/wide.json).  Bit-exact."""
import random

import numpy as np
import pytest

from oracle import pyoracle as po
from fieldutil import pack, unpack, lshape

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def primes():
    """2^k - c primes (PM192: MPyC's defaults) and primes of no special shape (MONT192): the root-of-unity prime of
    SecInt(104, n=118) -- the field of demos/np_lpsolver.py -i5 --, the first prime above 2^128, one close to 2^192"""
    from mpyc_amd.finfields import find_prime_root, next_prime
    return [find_prime_root(l)[0] for l in (129, 136, 160, 192)] + [find_prime_root(136, n=118)[0], next_prime(2**128),
                                                                    next_prime(2**192 - 2**40)]


@pytest.fixture(scope='module')
def eng():
    assert torch.cuda.is_available()
    from mpyc_amd import engine
    return engine


def dev(ctx, vals):
    return ctx.from_numpy(pack(vals, 24))


def host(a):
    return unpack(a.to_numpy(), 24)


def test_pm192_elementwise_sharing_and_linear_algebra(eng):
    rng = random.Random(24)
    for p in primes():
        F = po.Field(p, False)
        ctx = eng.FieldContext(p, device=0)
        assert ctx.elem_bytes == 24 and ctx.scalar_limbs == 3 and ctx.limbs == 3
        for n in (1, 2, 63, 4099):                       # vector path, ragged tails
            edge = [0, 1, p - 1, p - 2, 2**64, 2**128 - 1, 2**128, (p - 1) // 2][:n]
            a = edge + [rng.randrange(p) for _ in range(n - len(edge))]
            b = list(reversed(edge)) + [rng.randrange(p) for _ in range(n - len(edge))]
            c = [rng.randrange(p) for _ in range(n)]
            A, B, C = dev(ctx, a), dev(ctx, b), dev(ctx, c)
            assert host(ctx.add(A, B)) == [(x + y) % p for x, y in zip(a, b)]
            assert host(ctx.sub(A, B)) == [(x - y) % p for x, y in zip(a, b)]
            assert host(ctx.mul(A, B)) == [(x * y) % p for x, y in zip(a, b)]
            assert host(ctx.neg(A)) == [(-x) % p for x in a]
            assert host(ctx.muladd(A, B, C)) == [(x * y + z) % p for x, y, z in zip(a, b, c)]
            s = rng.randrange(p)
            assert host(ctx.mul_scalar(A, s)) == [(x * s) % p for x in a]
            assert host(ctx.add_scalar(A, s)) == [(x + s) % p for x in a]
            assert host(ctx.rsub_scalar(A, s)) == [(s - x) % p for x in a]
            raw = [rng.randrange(2**192) for _ in range(n)]
            assert host(ctx.reduce(dev(ctx, raw))) == [x % p for x in raw]
            e = rng.randrange(p)
            assert host(ctx.pow(A, e)) == [pow(x, e, p) for x in a]
            assert host(ctx.pow(A, 0)) == [1] * n
            nz = [x or 5 for x in a]
            inv = host(ctx.inv(dev(ctx, nz)))
            assert [(x * y) % p for x, y in zip(nz, inv)] == [1] * n
            assert host(ctx.dot(A, B)) == [sum(x * y for x, y in zip(a, b)) % p]
            assert host(ctx.sum(A)) == [sum(a) % p]
            # sharing: supplied coefficients (reference convention), fused product, every t, Lagrange from any subset
            for t, m in ((1, 3), (2, 5), (3, 7), (4, 9), (5, 11)):
                coef = [[rng.randrange(p) for _ in range(n)] for _ in range(t)]
                Cm = ctx.matrix_from_numpy(pack([v for row in coef for v in row], 24).reshape(lshape(24, t, n)))
                draws = [v for row in coef for v in row]
                want = po.np_random_split(F, a, t, m, draws)
                sh = ctx.split(A, Cm, t, m)
                got = [unpack(sh.to_numpy()[i], 24) for i in range(m)]
                assert got == want, (hex(p), n, t)
                fused = ctx.split(A, Cm, t, m, mul_by=B)
                prod = [(x * y) % p for x, y in zip(a, b)]
                assert [unpack(fused.to_numpy()[i], 24) for i in range(m)] == po.np_random_split(F, prod, t, m, draws)
                xs = rng.sample(range(1, m + 1), t + 1)
                lam = po.recombination_vector(F, xs, 0)
                assert host(ctx.recombine([sh.row(x - 1) for x in xs], lam)) == a
                if t == 3:                               # several targets at once (w > 1)
                    lam2 = lam + po.recombination_vector(F, xs, 7)
                    out = ctx.recombine([sh.row(x - 1) for x in xs], lam2, w=2)
                    assert unpack(out.to_numpy()[0], 24) == a and unpack(out.to_numpy()[1], 24) == want[6]
            # device CSPRNG: split_rng(key) == split(rng_coeffs(key)); samples canonical; gate kernel
            key = bytes(range(32))
            for t, m in ((1, 3), (3, 7)):
                cm = ctx.rng_coeffs(key, 9, t, n)
                flat = [unpack(cm.to_numpy()[j], 24) for j in range(t)]
                assert all(0 <= v < p for row in flat for v in row)
                s1 = ctx.split_rng(A, t, m, key=key, nonce=9)
                s2 = ctx.split(A, cm, t, m)
                assert (s1.to_numpy() == s2.to_numpy()).all()
                k = 2 * t + 1
                lamk = po.recombination_vector(F, list(range(1, k + 1)), 0)
                g = ctx.gate([s1.row(j) for j in range(k)], lamk, None, None, t, m, key=key, nonce=11)
                sq = [(x * x) % p for x in a]
                assert host(ctx.recombine([g.row(j) for j in range(k)], lamk)) == sq
        # dense / skinny products and Gaussian elimination (VALU kernels; the matrix cores stop at 128 bits)
        for (M, K, N) in ((5, 7, 3), (70, 40, 1), (1, 300, 130), (33, 200, 35)):
            Am = [rng.randrange(p) for _ in range(M * K)]
            Bm = [rng.randrange(p) for _ in range(K * N)]
            Am[0] = Bm[0] = p - 1
            got = host(ctx.matmul(dev(ctx, Am), dev(ctx, Bm), M, K, N))
            want = [sum(Am[i * K + kk] * Bm[kk * N + j] for kk in range(K)) % p for i in range(M) for j in range(N)]
            assert got == want, (hex(p), M, K, N)


@pytest.mark.parametrize('root_n', [1, 118])
def test_pm192_mirror_api_wire_and_full_size_round_trip(root_n):
    from mpyc_amd import finfields, thresha
    rng = random.Random(7)
    p = finfields.find_prime_root(136, n=root_n)[0]        # n = 1: 2^136 - c; n = 118: 1 + 2n(3 + 2j), no special shape
    F = finfields.GF(p)
    assert F.byte_length == 17
    a = [rng.randrange(p) for _ in range(200)]
    b = [rng.randrange(p) for _ in range(200)]
    A, B = F.array(a), F.array(b)
    ints = lambda x: [int(v) for v in x.value.reshape(-1)]
    assert ints(A * B) == [(x * y) % p for x, y in zip(a, b)]
    assert ints(A - B) == [(x - y) % p for x, y in zip(a, b)]
    assert ints(F.array([-1, p + 5, 2**200])) == [p - 1, 5, 2**200 % p]
    assert ints((A * A.reciprocal())[:5]) == [1] * 5 if all(a[:5]) else True
    M = F.array([[rng.randrange(p) for _ in range(6)] for _ in range(6)])
    Minv = np.linalg.inv(M)
    assert ints(M @ Minv) == [int(i == j) for i in range(6) for j in range(6)]
    w = A.to_wire()
    assert w == b''.join(v.to_bytes(17, 'little') for v in a) and ints(F.array.from_wire(w)) == a
    # second-tier operations and array plumbing of the mirror over a three-limb field
    sq = A * A
    r = sq.sqrt()
    assert ints(r * r) == ints(sq) and bool(sq.is_sqr().all())
    assert ints(A ** 5) == [pow(x, 5, p) for x in a] and ints(A ** -1 * A) == [1] * 200
    assert ints(A << 3) == [(x << 3) % p for x in a] and ints((A << 3) >> 3) == a
    assert ints(np.cumsum(A[:17])) == [sum(a[:i + 1]) % p for i in range(17)]
    A2 = A.reshape(10, 20)
    assert ints(np.sum(A2, axis=0)) == [sum(a[i * 20 + j] for i in range(10)) % p for j in range(20)]
    assert ints(A2 @ B.reshape(20, 10)) == [sum(a[i * 20 + k] * b[k * 10 + j] for k in range(20)) % p
                                           for i in range(10) for j in range(10)]
    C = np.concatenate((A[:3], B[-2:]))
    C[1] = F(7)
    assert ints(C) == [a[0], 7, a[2], b[-2], b[-1]] and bool((A == A.copy()).all()) and not bool((A == B).all())
    assert ints(-A + A) == [0] * 200 and ints(A / A) == [1] * 200 if all(a) else True
    sh = thresha.np_random_split(F, A * B, 2, 5)                         # fused product + device CSPRNG
    y = thresha.np_recombine(F, [(x, sh[x - 1]) for x in (5, 1, 3)])
    assert ints(y) == [(x * y_) % p for x, y_ in zip(a, b)]
    # 10^6 secrets: any t+1 rows give back the secrets, t rows do not determine them (degree check: 2t+1 rows agree)
    n = 1_000_000
    S = F.array(torch.randint(0, 2**62, (n,), dtype=torch.int64).numpy())
    sh = thresha.np_random_split(F, S, 3, 7)
    r1 = thresha.np_recombine(F, [(x, sh[x - 1]) for x in (1, 2, 3, 4)])
    r2 = thresha.np_recombine(F, [(x, sh[x - 1]) for x in (7, 5, 2, 6)])
    r3 = thresha.np_recombine(F, [(x, sh[x - 1]) for x in range(1, 8)])
    assert bool((r1 == S).all()) and bool((r2 == S).all()) and bool((r3 == S).all())


def test_pm192_golden_vectors_from_the_reference(eng, golden_wide):
    """element-wise results, both sharing conventions, recombination from several point sets and at several targets:
    the reference's own outputs for the 129 / 136 / 160 / 192-bit default primes (tests/golden/wide.json)"""
    import test_gpu_parity as tp
    tp.test_golden_elementwise(eng, golden_wide)
    tp.test_golden_sharing(eng, golden_wide)


def test_wave_contiguous_accesses_of_24_byte_elements(eng):
    """Round 6: the streaming kernels move 24-byte elements wave by wave (kernels.hpp ldgw / stgw: 1536 contiguous bytes
    per wave as dwordx4, sorted out through LDS) for whole waves of 16-byte aligned rows, and fall back to the scalar
    tail for the ragged rest and for rows that are only 8-byte aligned.  Element-wise operations, fused multiply-add,
    share generation (supplied coefficients, fused product) and the Beaver combination around every wave boundary, on
    views at odd element offsets, in place, for a 2^k - c prime and a prime of no special shape."""
    from mpyc_amd.finfields import find_prime_root, next_prime
    rng = random.Random(2406)
    for p in (find_prime_root(136)[0], next_prime(2**192 - 2**40)):
        F = po.Field(p, False)
        ctx = eng.FieldContext(p, device=0)
        big = 3 * 256 * 64 + 130                                  # several workgroups, several waves each, a ragged end
        pool_a = [rng.randrange(p) for _ in range(big + 8)]
        pool_b = [rng.randrange(p) for _ in range(big + 8)]
        pool_a[:4] = [0, 1, p - 1, 2**128]
        DA, DB = dev(ctx, pool_a), dev(ctx, pool_b)
        for n in (63, 64, 65, 127, 128, 129, 191, 192, 256 + 64, 1000, big):
            for off in (0, 1, 2):                                 # element offsets 1 (8-byte aligned only) and 2 (16-byte aligned)
                a, b = pool_a[off:off + n], pool_b[off:off + n]
                A = eng.DevArray(ctx, DA.t[off:off + n], n)
                B = eng.DevArray(ctx, DB.t[off:off + n], n)
                assert host(ctx.mul(A, B)) == [(x * y) % p for x, y in zip(a, b)], (hex(p), n, off)
                assert host(ctx.add(A, B)) == [(x + y) % p for x, y in zip(a, b)], (hex(p), n, off)
                assert host(ctx.sub(A, B)) == [(x - y) % p for x, y in zip(a, b)], (hex(p), n, off)
                s = rng.randrange(p)
                assert host(ctx.mul_scalar(A, s)) == [(x * s) % p for x in a]
                assert host(ctx.neg(A)) == [(-x) % p for x in a]
                assert host(ctx.muladd(A, B, A)) == [(x * y + x) % p for x, y in zip(a, b)]
            # in place, and an output view at an odd offset of a larger buffer
            a, b = pool_a[:n], pool_b[:n]
            A, B = dev(ctx, a), dev(ctx, b)
            ctx.mul(A, B, out=A)
            assert host(A) == [(x * y) % p for x, y in zip(a, b)]
            buf = ctx.empty(n + 3)
            O = eng.DevArray(ctx, buf.t[1:1 + n], n)
            ctx.add(dev(ctx, a), B, out=O)
            assert host(O) == [(x + y) % p for x, y in zip(a, b)]
            # Beaver combination z + d*y + e*x (+ d*e): five operand rows in flight
            z, d, e = ([rng.randrange(p) for _ in range(n)] for _ in range(3))
            got = ctx.beaver_combine(dev(ctx, z), A, B, dev(ctx, d), dev(ctx, e), True)
            a2 = host(A)
            assert host(got) == [(zz + dd * y + ee * x + dd * ee) % p for zz, x, y, dd, ee in zip(z, a2, b, d, e)]
            # share generation with supplied coefficients (+ fused product) and recombination from 2t+1 rows
            for t, m in ((1, 3), (3, 7)):
                coef = [[rng.randrange(p) for _ in range(n)] for _ in range(t)]
                Cm = ctx.matrix_from_numpy(pack([v for row in coef for v in row], 24).reshape(lshape(24, t, n)))
                draws = [v for row in coef for v in row]
                A = dev(ctx, a)
                sh = ctx.split(A, Cm, t, m)
                assert [unpack(sh.to_numpy()[i], 24) for i in range(m)] == po.np_random_split(F, a, t, m, draws), (hex(p), n, t)
                prod = [(x * y) % p for x, y in zip(a, b)]
                fused = ctx.split(A, Cm, t, m, mul_by=B)
                assert [unpack(fused.to_numpy()[i], 24) for i in range(m)] == po.np_random_split(F, prod, t, m, draws)
                xs = list(range(1, 2 * t + 2))
                lam = po.recombination_vector(F, xs, 0)
                assert host(ctx.recombine([fused.row(x - 1) for x in xs], lam)) == prod
