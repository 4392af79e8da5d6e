"""Randomised differential sweep on the GPU: many (field, n, t, m, alignment) combinations, every
kernel family against the pinned C oracle.  Fields are drawn from all reduction strategies, including
bit lengths no other test uses (33..128-bit pseudo-Mersenne and generic primes, GF(2^n) for odd n)."""
import random

import pytest

from oracle import pyoracle as po
from fieldutil import unpack, lshape
from test_gpu_parity import rand_np

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def field_pool():
    from mpyc_amd.finfields import is_prime, prev_prime
    from mpyc_amd.gfpx import BinaryPolynomial
    rng = random.Random(20260925)
    primes = []
    for k in (33, 34, 40, 47, 56, 61, 63, 64, 65, 66, 80, 89, 96, 100, 113, 127, 128):   # pseudo-Mersenne: largest below 2^k
        primes.append(prev_prime(1 << k))
    for k in (40, 63, 64, 80, 96, 128):                  # pseudo-Mersenne with the LARGEST admissible c (tightest carry chains)
        cb = min((k - 1) // 2, 31) if k <= 64 else 31
        c = (1 << cb) - 1
        while not is_prime((1 << k) - c):
            c -= 2
        primes.append((1 << k) - c)
    for k in (17, 24, 31, 32, 35, 48, 62, 64, 70, 90, 110, 128):                       # generic: random k-bit primes
        x = rng.getrandbits(k) | (1 << (k - 1)) | 1
        while not is_prime(x):
            x += 2
        if x.bit_length() == k:
            primes.append(x)
    binaries = [int(BinaryPolynomial.next_irreducible(1 << d)) for d in (1, 2, 3, 5, 7, 8, 9, 13, 31, 32, 33, 47, 63, 64, 65, 77, 127, 128)]
    binaries.append(int(BinaryPolynomial.next_irreducible((1 << 60) | (1 << 50))))      # dense r: window kernel
    return [(p, False) for p in primes] + [(b, True) for b in binaries]


@pytest.fixture(scope='module')
def eng():
    assert torch.cuda.is_available()
    from mpyc_amd import engine
    return engine


def test_random_sweep(eng, coracle):
    rng = random.Random(7)
    pool = field_pool()
    assert len(pool) > 40
    key = bytes(rng.randrange(256) for _ in range(32))
    for it, (modulus, binary) in enumerate(pool):
        F = po.Field(modulus, binary)
        ctx = eng.FieldContext(modulus, binary, device=0)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        slow = eb >= 12
        for rep in range(5):
            n = rng.choice([1, 2, 3, 15, 16, 17, 255, 256, 257, rng.randrange(300, 3000)]) if not slow else \
                rng.choice([1, 2, 17, rng.randrange(100, 700)])
            off = rng.choice([0, 0, 1]) if eb < 16 else 0        # off = 1: pointers not 16-byte aligned
            A, B = rand_np(F, eb, n + off, 1000 + it), rand_np(F, eb, n + off, 2000 + it)
            dA0, dB0 = ctx.from_numpy(A), ctx.from_numpy(B)
            dA = eng.DevArray(ctx, dA0.t[off:], n)
            dB = eng.DevArray(ctx, dB0.t[off:], n)
            A, B = A[off:], B[off:]
            tag = (hex(modulus), n, off)
            prod = cf.ew(coracle.MUL, A, B)
            assert (ctx.mul(dA, dB).to_numpy() == prod).all(), ('mul',) + tag
            assert (ctx.add(dA, dB).to_numpy() == cf.ew(coracle.ADD, A, B)).all(), ('add',) + tag
            assert (ctx.sub(dA, dB).to_numpy() == cf.ew(coracle.SUB, A, B)).all(), ('sub',) + tag
            sc = rng.randrange(F.order)
            assert unpack(ctx.mul_scalar(dA, sc).to_numpy(), eb) == [po.mul(F, x, sc) for x in unpack(A, eb)], ('muls',) + tag
            m = rng.choice([1, 2, 3, 4, 5, 7, 9, 12])
            if m >= F.order:
                m = max(1, F.order - 1)
            t = rng.randrange(0, m)
            Cn = rand_np(F, eb, max(t, 1) * n, 3000 + it).reshape(lshape(eb, max(t, 1), n))
            dC = ctx.matrix_from_numpy(Cn)
            want = cf.split(A, Cn, t, m)
            sh = ctx.split(dA, dC, t, m)
            assert (sh.to_numpy() == want).all(), ('split', t, m) + tag
            assert (ctx.split(dA, dC, t, m, mul_by=dB).to_numpy() == cf.split(prod, Cn, t, m)).all(), ('mul_split', t, m) + tag
            xs = rng.sample(range(1, m + 1), t + 1)
            lam = po.recombination_vector(F, xs, 0)
            assert (ctx.recombine([sh.row(x - 1) for x in xs], lam).to_numpy() == A).all(), ('rec', t, m, xs) + tag
            if t:
                rounds = rng.choice([8, 12, 20])
                nonce = rng.getrandbits(64)
                Cr = ctx.rng_coeffs(key, nonce, t, n, rounds)
                assert (Cr.to_numpy() == coracle.rng_coeffs(cf, key, nonce, rounds, t, n)).all(), ('rng', t) + tag
                assert (ctx.split_rng(dA, t, m, key=key, nonce=nonce, rounds=rounds).to_numpy() ==
                        ctx.split(dA, Cr, t, m).to_numpy()).all(), ('split_rng', t, m) + tag
            e = rng.choice([0, 1, 2, 5, 65537, F.order - 2 if F.order > 2 else 1])
            vals = unpack(A, eb)
            got = unpack(ctx.pow(dA, e).to_numpy(), eb)
            chk = range(0, n, max(1, n // 40))
            for i in chk:
                r, b_, ee = 1, vals[i], e
                while ee:
                    if ee & 1:
                        r = po.mul(F, r, b_)
                    b_ = po.mul(F, b_, b_)
                    ee >>= 1
                assert got[i] == r, ('pow', e, i) + tag
            inv = unpack(ctx.inv(dA, check_zero=False).to_numpy(), eb)
            for i in chk:
                assert (inv[i] == 0) if vals[i] == 0 else (po.mul(F, inv[i], vals[i]) == 1), ('inv', i) + tag


def test_digit_arithmetic_every_bit_length(eng):
    """Round 6: recombination (dot products in 28-bit digits, fields.hpp LazyDot) and exponentiation (product chains in
    ceil(k / NL)-bit digits, DigitChain; exponents 3 e' + 1 as (e', r^3 a)) over the 2^k - c primes of EVERY bit length
    65..192 -- every digit width and count the kernels can take -- against Python integers."""
    from mpyc_amd.finfields import find_prime_root
    from fieldutil import pack
    rng = random.Random(6192)
    for bits in range(65, 193):
        p = find_prime_root(bits)[0]
        ctx = eng.FieldContext(p, device=0)
        eb = ctx.elem_bytes
        n = 70
        rows = [[0, 1, p - 1, p - 2, (p - 1) // 2, 2**64 % p] + [rng.randrange(p) for _ in range(n - 6)] for _ in range(9)]
        dev = [ctx.from_numpy(pack(r, eb)) for r in rows]
        for k in (1, 2, 5, 9):
            lam = [p - 1 if j % 4 == 0 else rng.randrange(p) for j in range(k)]
            got = unpack(ctx.recombine(dev[:k], lam).to_numpy(), eb)
            assert got == [sum(lam[j] * rows[j][i] for j in range(k)) % p for i in range(n)], (bits, k)
        a = rows[0]
        exps = [(p + 1) // 4, (3 * p - 5) // 4, (p - 1) // 2, p - 2, rng.randrange(p), 3 * (2**40 - 1) + 1]
        for e in exps:
            got = unpack(ctx.pow(dev[0], e).to_numpy(), eb)
            assert got == [pow(x, e, p) for x in a], (bits, hex(e))
        # batched inverse in digits (k_inv_digits: arrays of at least 4096 elements): 24-32 elements per thread share one
        # exponentiation; zeros are flagged and map to 0, ragged sizes, every element checked by a * a^-1 = 1
        for nn in (4096, 4096 + 37 + bits):
            vals = [rng.randrange(1, p) for _ in range(nn)]
            for z in (0, 31, 32, nn // 2, nn - 1):
                vals[z] = 0
            vals[1], vals[2], vals[3] = 1, p - 1, 2
            d = ctx.from_numpy(pack(vals, eb))
            with pytest.raises(ZeroDivisionError):
                ctx.inv(d)
            inv = unpack(ctx.inv(d, check_zero=False).to_numpy(), eb)
            assert all((x * y) % p == (1 if x else 0) and y < p for x, y in zip(vals, inv)), (bits, nn)
            vals2 = [v or 7 for v in vals]
            inv2 = unpack(ctx.inv(ctx.from_numpy(pack(vals2, eb))).to_numpy(), eb)
            assert all((x * y) % p == 1 for x, y in zip(vals2, inv2)), (bits, nn)
