"""GPU parity tests: the HIP path, called through the C ABI (mpyc_amd.engine -> libffgpu.so),
against (1) the golden vectors produced by the real reference, (2) the pinned oracle on seeded
inputs, (3) size-independent properties at BASELINE.json's full sizes (10^7).
Bit-exact everywhere: this is integer / GF(2) arithmetic."""
import random

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle.coracle import elem_bytes
from fieldutil import P61, P64, P128, edge_values, field_of, pack, unhex, unpack, lshape

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def eng():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from mpyc_amd import engine
    return engine


_ctx_cache = {}


def ctx_for(eng, modulus, binary):
    key = (modulus, binary)
    if key not in _ctx_cache:
        _ctx_cache[key] = eng.FieldContext(modulus, binary, device=0)
    return _ctx_cache[key]


def dev(ctx, vals):
    return ctx.from_numpy(pack(vals, ctx.elem_bytes))


def host(arr):
    return unpack(arr.to_numpy(), arr.ctx.elem_bytes)


def devmat(ctx, rows_of_ints):
    eb = ctx.elem_bytes
    r, n = len(rows_of_ints), len(rows_of_ints[0])
    flat = pack([v for row in rows_of_ints for v in row], eb)
    return ctx.matrix_from_numpy(flat.reshape(lshape(eb, r, n)))


def hostmat(mtx):
    a = mtx.to_numpy()
    return [unpack(a[i], mtx.ctx.elem_bytes) for i in range(mtx.rows)]


# ---------------------------------------------------------------------------
# 1. golden vectors from the reference
# ---------------------------------------------------------------------------
def test_golden_elementwise(eng, golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        ctx = ctx_for(eng, F.modulus, F.binary)
        assert ctx.elem_bytes == elem_bytes(F.modulus, F.binary)
        a, b = unhex(case['a']), unhex(case['b'])
        A, B = dev(ctx, a), dev(ctx, b)
        assert host(ctx.add(A, B)) == unhex(case['add']), name
        assert host(ctx.sub(A, B)) == unhex(case['sub']), name
        assert host(ctx.mul(A, B)) == unhex(case['mul']), name
        assert host(ctx.neg(A)) == unhex(case['neg']), name
        sc = int(case['scalar'], 16)
        assert host(ctx.add_scalar(A, sc)) == unhex(case['add_scalar']), name
        assert host(ctx.mul_scalar(A, sc)) == unhex(case['mul_scalar']), name
        assert host(ctx.rsub_scalar(A, sc)) == unhex(case['rsub_scalar']), name
        if case['raw_width'] == 8 * ctx.elem_bytes:
            assert host(ctx.reduce(dev(ctx, unhex(case['raw'])))) == unhex(case['raw_reduced']), name


def test_golden_sharing(eng, golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        ctx = ctx_for(eng, F.modulus, F.binary)
        a = unhex(case['a'])
        s = a[:13] + a[-5:]
        n = len(s)
        S = dev(ctx, s)
        for sc in case['sharing']:
            t, m, draws = sc['t'], sc['m'], unhex(sc['draws'])
            C = devmat(ctx, [draws[j * n:(j + 1) * n] for j in range(t)]) if t else None
            sh = ctx.split(S, C, t, m)
            got = hostmat(sh)
            assert got == [unhex(r) for r in sc['np_shares']], (name, t, m)
            # list-path convention = np kernel fed with permuted draws (thresha.py:37-43)
            if t:
                perm = po.list_to_np_draws(draws, t, n)
                Cl = devmat(ctx, [perm[j * n:(j + 1) * n] for j in range(t)])
                assert hostmat(ctx.split(S, Cl, t, m)) == [unhex(r) for r in sc['list_shares']], (name, t, m)
            for rec in sc['recombine']:
                xs = rec['xs']
                out = ctx.recombine([sh.row(x - 1) for x in xs], unhex(rec['vector']))
                assert host(out) == unhex(rec['np_out']), (name, t, m, xs)
            mu = sc['multi']
            lam = [v for vv in mu['vectors'] for v in unhex(vv)]
            out = ctx.recombine([sh.row(x - 1) for x in mu['xs']], lam, w=len(mu['x_rs']))
            assert hostmat(out) == [unhex(r) for r in mu['out']], (name, t, m)


def test_golden_sbox_and_kats(eng, golden_sbox):
    ctx = ctx_for(eng, 0x11b, True)
    x = dev(ctx, list(range(256)))
    out = ctx.sbox(x, golden_sbox['rows8'], golden_sbox['b'])
    assert host(out) == golden_sbox['table']
    # tests/test_finfields.py:29-30,94-99
    a = dev(ctx, [16, 32, 57, 3, 48])
    b = dev(ctx, [16, 16, 67, 3, 16])
    assert host(ctx.mul(a, b)) == [27, 54, 137, 5, 45]
    # x^254 * x == 1 for x != 0 (tests/test_runtime.py:818-843), through the mul kernel
    v = dev(ctx, list(range(1, 256)))
    d = v
    c = ctx.mul(d, d); c = ctx.mul(c, c); c = ctx.mul(c, c); c = ctx.mul(c, d); c = ctx.mul(c, c)
    c, d = ctx.mul(c, c), ctx.mul(c, d)
    c, d = ctx.mul(c, c), ctx.mul(c, d)
    c = ctx.mul(c, d); c = ctx.mul(c, c)
    assert host(c) == golden_sbox['pow254'][1:]
    assert host(ctx.mul(c, v)) == [1] * 255
    # big random S-box batch vs the table
    rng = np.random.default_rng(1)
    big = rng.integers(0, 256, size=1_000_003, dtype=np.uint8)
    got = ctx.sbox(ctx.from_numpy(big), golden_sbox['rows8'], golden_sbox['b']).to_numpy()
    assert (got == np.array(golden_sbox['table'], dtype=np.uint8)[big]).all()


# ---------------------------------------------------------------------------
# 2. seeded inputs vs the pinned C oracle (vector path + scalar tails)
# ---------------------------------------------------------------------------
FIELDS = [(P61, False), (P64, False), (P128, False), (2**127 - 1, False), (2**96 - 17, False),
          (6616326157076047771, False), (258797994007609146293811961253269568351, False), (2**31 - 1, False),
          (65537, False), (0x11b, True), (0b10011, True), ((1 << 64) | 0x1b, True), ((1 << 128) | 0x87, True),
          # GF(2^n), 9 <= n <= 32: four-byte storage (GF2W32, round 6) -- sparse moduli (fold reduction) and one whose low part is long
          # (long division of the 64-bit product)
          (0x1002b, True), (0x10000008d, True), (0x203, True), (0x1c0003, True)]


def rand_np(F, eb, n, seed):
    """Uniform canonical elements, edge block first."""
    rng = np.random.default_rng(seed)
    if F.binary:
        nb = F.n
    else:
        nb = F.modulus.bit_length()
    raw = rng.integers(0, 2**64, size=(n, 2), dtype=np.uint64)
    if eb <= 8:
        if F.binary:
            v = raw[:, 0] & np.uint64((1 << nb) - 1)
        else:
            v = (raw[:, 0] >> np.uint64(64 - nb)) if nb < 64 else raw[:, 0]
            p = np.uint64(F.modulus)
            v = np.where(v >= p, v - p, v)
        out = v.astype({1: np.uint8, 4: np.uint32, 8: np.uint64}[eb])
    else:
        hi_bits = nb - 64
        hi = raw[:, 1] >> np.uint64(64 - hi_bits) if hi_bits < 64 else raw[:, 1]
        out = np.stack([raw[:, 0], hi], axis=1)
        # canonicalise with python for the (rare) >= p case
        if not F.binary:
            p = F.modulus
            big = (out[:, 1] > np.uint64(p >> 64)) | ((out[:, 1] == np.uint64(p >> 64)) & (out[:, 0] >= np.uint64(p & (2**64 - 1))))
            for i in np.nonzero(big)[0]:
                v = ((int(out[i, 1]) << 64) | int(out[i, 0])) - p
                out[i, 0], out[i, 1] = v & (2**64 - 1), v >> 64
        if eb == 12:                                   # three 32-bit limbs: lo32, mid32, hi32 (hi < 2^32)
            out = np.stack([out[:, 0] & np.uint64(0xffffffff), out[:, 0] >> np.uint64(32), out[:, 1]],
                           axis=1).astype(np.uint32)
    ev = pack(edge_values(F), eb)
    k = min(len(ev), n)
    out[:k] = ev[:k]
    return out


@pytest.mark.parametrize('modulus,binary', FIELDS)
def test_vs_oracle_elementwise(eng, coracle, modulus, binary):
    F = po.Field(modulus, binary)
    ctx = ctx_for(eng, modulus, binary)
    eb = ctx.elem_bytes
    cf = coracle.CField(modulus, binary)
    n = 20011 if eb >= 12 and (binary or ctx.reduction == 'montgomery') else 100003   # odd: exercises tails
    A, B, Cc = rand_np(F, eb, n, 11), rand_np(F, eb, n, 12), rand_np(F, eb, n, 13)
    dA, dB, dC = ctx.from_numpy(A), ctx.from_numpy(B), ctx.from_numpy(Cc)
    assert (ctx.add(dA, dB).to_numpy() == cf.ew(coracle.ADD, A, B)).all()
    assert (ctx.sub(dA, dB).to_numpy() == cf.ew(coracle.SUB, A, B)).all()
    prod = cf.ew(coracle.MUL, A, B)
    assert (ctx.mul(dA, dB).to_numpy() == prod).all()
    assert (ctx.neg(dA).to_numpy() == cf.ew(coracle.NEG, A)).all()
    assert (ctx.muladd(dA, dB, dC).to_numpy() == cf.ew(coracle.ADD, prod, Cc)).all()
    # in place (finfields.py:1114-1124 __imul__)
    tmp = dA.clone()
    ctx.mul(tmp, dB, out=tmp)
    assert (tmp.to_numpy() == prod).all()
    # raw reduction of arbitrary limb patterns
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 2**64, size=A.shape, dtype=np.uint64).astype(A.dtype) if eb != 1 else \
        rng.integers(0, 256, size=A.shape, dtype=np.uint8)
    assert (ctx.reduce(ctx.from_numpy(raw)).to_numpy() == cf.ew(coracle.REDUCE, raw)).all()


@pytest.mark.parametrize('modulus,binary', FIELDS)
def test_vs_oracle_sharing(eng, coracle, modulus, binary):
    F = po.Field(modulus, binary)
    ctx = ctx_for(eng, modulus, binary)
    eb = ctx.elem_bytes
    cf = coracle.CField(modulus, binary)
    n = 3001 if eb >= 12 else 30011          # the C oracle's two-limb mulmod is shift-and-add
    S, B = rand_np(F, eb, n, 21), rand_np(F, eb, n, 22)
    dS, dB = ctx.from_numpy(S), ctx.from_numpy(B)
    for (t, m) in [(0, 1), (1, 3), (2, 5), (3, 7), (4, 9), (6, 13)]:
        if m >= F.order:
            continue
        Cn = rand_np(F, eb, max(t, 1) * n, 30 + t).reshape(lshape(eb, max(t, 1), n))
        dC = ctx.matrix_from_numpy(Cn)
        want = cf.split(S, Cn, t, m)
        sh = ctx.split(dS, dC, t, m)
        assert (sh.to_numpy() == want).all(), (t, m)
        # fused local product + split == split(mul)
        prod = cf.ew(coracle.MUL, S, B)
        fused = ctx.split(dS, dC, t, m, mul_by=dB)
        assert (fused.to_numpy() == cf.split(prod, Cn, t, m)).all(), (t, m)
        # recombine from t+1 and (if available) 2t+1 rows, rotated x order
        for k in {t + 1, min(2 * t + 1, m)}:
            xs = [((3 + j) % m) + 1 for j in range(k)]
            lam = po.recombination_vector(F, xs, 0)
            rec = ctx.recombine([sh.row(x - 1) for x in xs], lam)
            assert (rec.to_numpy() == cf.recombine([want[x - 1] for x in xs], lam)).all(), (t, m, k)
            assert (rec.to_numpy() == S).all(), (t, m, k)          # round trip: it IS the secret


@pytest.mark.parametrize('modulus,binary', FIELDS)
def test_pow_and_inverse_kernels(eng, modulus, binary):
    """Second-tier ops (finfields.py:1159-1187,1278-1281,1408-1422): in-register pow with a public
    exponent and the batched inverse, against Python integers / the GF(2^n) oracle."""
    F = po.Field(modulus, binary)
    ctx = ctx_for(eng, modulus, binary)
    eb = ctx.elem_bytes
    q = F.order
    n = 3000 + 7
    A = rand_np(F, eb, n, 71)
    vals = unpack(A, eb)
    dA = ctx.from_numpy(A)

    def fpow(x, e):
        if not binary:
            return pow(x, e, modulus)
        r, b = 1, x
        while e:
            if e & 1:
                r = po.mul(F, r, b)
            b = po.mul(F, b, b)
            e >>= 1
        return r

    # exponent shapes of ff_pow (kernels.hpp): leading runs of ones (raised by doubling) of lengths around the
    # threshold and across limb boundaries, runs followed by sparse and by dense tails (plain / sliding windows), powers of 2
    shapes = [2**11 - 1, 2**12 - 1, 2**13 - 1, 2**40 - 1, (2**17 - 1) << 20 | 0xABCDE, (2**12 - 1) << 3, (2**29 - 1) << 40 | 1,
              2**59, (q - 1) // 2, q // 3 | 1, 0x9E3779B97F4A7C15F39CC0605CEDC834 % q]
    if q.bit_length() > 64:
        shapes += [2**64 - 1, 2**65 - 1, (2**70 - 1) << 5 | 9, 2**64, 2**63]
    # exponents 3 e' + 1 with a long run of ones in e' (round 6: ffgpu_pow raises to e' and finishes with r^3 a when that
    # chain is shorter): the inverse square root exponent (3q - 5) / 4 of a prime q = 3 mod 4 (finfields.py:1424-1437) and
    # constructed ones, beside neighbours that must NOT take that form
    for e1 in ((q - 3) // 4, 2**20 - 1, 2**45 - 5, (2**33 - 1) << 7 | 0x55, q // 5):
        if e1 > 8:
            shapes += [3 * e1 + 1, 3 * e1 + 2, 3 * e1]
    for e in [0, 1, 2, 3, 254, 65537, q - 2, q - 1, (q + 1) // 4 if q > 4 else 1] + [e for e in shapes if 0 < e < q]:
        if e < 0:
            continue
        got = unpack(ctx.pow(dA, e).to_numpy(), eb)
        sample = range(0, n, 97) if (eb >= 12 or binary) and e > 1000 else range(n)
        assert all(got[i] == fpow(vals[i], e) for i in sample), (hex(modulus), e)
    # inverse: zeros are flagged (ZeroDivisionError) and map to 0 when unchecked
    with pytest.raises(ZeroDivisionError):
        ctx.inv(dA)                                   # edge block contains 0
    inv = unpack(ctx.inv(dA, check_zero=False).to_numpy(), eb)
    for i in range(n):
        if vals[i] == 0:
            assert inv[i] == 0
        else:
            assert po.mul(F, inv[i], vals[i]) == 1, (hex(modulus), i)
    nz = [v if v else 1 for v in vals]
    dN = ctx.from_numpy(pack(nz, eb))
    assert unpack(ctx.mul(ctx.inv(dN), dN).to_numpy(), eb) == [1] * n
    if eb < 16:                                       # unaligned -> scalar path
        va = eng.DevArray(ctx, dN.t[1:], n - 1)
        assert unpack(ctx.mul(ctx.inv(va), va).to_numpy(), eb) == [1] * (n - 1)


@pytest.mark.gpu
@pytest.mark.parametrize('modulus', [P61, P64, 2**40 - 87, 2**63 - 25, 8123557937065977257, 2**33 - 9])
def test_batched_inverse_full_batch_kernel(eng, modulus, monkeypatch):
    """k_inv_fast (kernels.hpp; one-word fields, n >= 32768): full blocks without bounds checks + the extra blocks for
    the leftover packs and the odd last element; the lean exponentiation (run of ones + short tail: every 2^k - c prime)
    and the window-table one (8123557937065977257: a prime whose q - 2 has 38 set bits); zeros anywhere give zero and
    raise; against Python's pow on a sample and a * a^-1 == 1 on every element; pieces of the array below the kernel's
    threshold (k_inv_batch) must give the same elements."""
    import torch
    ctx = ctx_for(eng, modulus, False)
    if ctx.elem_bytes != 8:
        pytest.skip('one-word fields only')
    r = random.Random(modulus % 1009)
    for n in (32768 * 2, 200_003, 1_000_000, 8192 * 32 + 5119 * 2 + 1):
        vals = [r.randrange(modulus) for _ in range(n)]
        for i in (0, 1, 2, 777, n // 2, n - 2, n - 1):
            vals[i] = 0
        vals[3], vals[4] = 1, modulus - 1
        dA = ctx.from_numpy(pack(vals, 8))
        with pytest.raises(ZeroDivisionError):
            ctx.inv(dA)
        inv = ctx.inv(dA, check_zero=False)
        got = inv.to_numpy().view(np.uint64)
        for i in list(range(0, 64)) + list(range(n - 64, n)) + [r.randrange(n) for _ in range(300)]:
            assert int(got[i]) == (pow(vals[i], modulus - 2, modulus) if vals[i] else 0), (hex(modulus), n, i)
        nz = dA.t != 0
        one = ctx.mul(dA, inv).t
        assert bool((one[nz] == 1).all()) and bool((inv.t[~nz] == 0).all())
        # the same array in pieces of fewer than 32768 elements takes the general kernel (k_inv_batch): same elements
        step = 30_000
        for lo in range(0, n, step * 7):
            part = ctx.inv(eng.DevArray(ctx, dA.t[lo:lo + step].clone(), min(step, n - lo)), check_zero=False)
            assert torch.equal(part.t, inv.t[lo:lo + step]), (hex(modulus), n, lo)
        vals2 = [v or 1 for v in vals]
        dN = ctx.from_numpy(pack(vals2, 8))
        ctx.inv(dN)                                      # no zeros: no exception


def test_binary_fields_dense_and_small(eng, coracle):
    """GF(2^n) multiplication paths: in-register carry-less product + fold (sparse moduli), 4-bit
    window LDS kernel (dense moduli), long division (n <= 32) -- all against the oracle; plus the
    gate identity recombine(split(a*b)) == a*b on a dense 128-bit modulus."""
    from mpyc_amd.gfpx import BinaryPolynomial
    mods = [int(BinaryPolynomial.next_irreducible((1 << 64) | (1 << 45))),
            int(BinaryPolynomial.next_irreducible((1 << 128) | (1 << 100))),
            int(BinaryPolynomial.next_irreducible(1 << 9)), int(BinaryPolynomial.next_irreducible(1 << 32)),
            int(BinaryPolynomial.next_irreducible(1 << 33)), int(BinaryPolynomial.next_irreducible(1 << 127))]
    for mod in mods:
        F = po.Field(mod, True)
        ctx = ctx_for(eng, mod, True)
        eb = ctx.elem_bytes
        cf = coracle.CField(mod, True)
        n = 20011
        A, B = rand_np(F, eb, n, 91), rand_np(F, eb, n, 92)
        dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
        want = cf.ew(coracle.MUL, A, B)
        assert (ctx.mul(dA, dB).to_numpy() == want).all(), hex(mod)
        t, m = 2, 5
        Cn = rand_np(F, eb, t * n, 93).reshape(lshape(eb, t, n))
        sh = ctx.split(dA, ctx.matrix_from_numpy(Cn), t, m, mul_by=dB)
        xs = [2, 4, 5, 1, 3]
        rec = ctx.recombine([sh.row(x - 1) for x in xs], po.recombination_vector(F, xs, 0))
        assert (rec.to_numpy() == want).all(), hex(mod)


def test_wide_binary_products_all_degrees(eng, coracle):
    """GF(2^n), 33 <= n <= 128, sparse moduli: element-wise products at every degree around the word and top-bit
    boundaries, extreme and random operands, in place, against the oracle."""
    from mpyc_amd.gfpx import BinaryPolynomial
    for deg in (33, 40, 60, 61, 62, 63, 64, 65, 96, 100, 124, 125, 126, 127, 128):
        mod = int(BinaryPolynomial.next_irreducible(1 << deg))
        assert mod.bit_length() - 1 == deg
        F = po.Field(mod, True)
        ctx = ctx_for(eng, mod, True)
        eb = ctx.elem_bytes
        cf = coracle.CField(mod, True)
        n = 16384 + 4099
        A, B = rand_np(F, eb, n, 700 + deg), rand_np(F, eb, n, 800 + deg)
        top = pack([F.order - 1, F.order >> 1, (F.order >> 1) | 1, 7 << (deg - 3), 5 << (deg - 3), 1 << (deg - 1)], eb)
        A[100:100 + len(top)] = top
        B[100:100 + len(top)] = top[::-1]
        A[200:200 + len(top)] = top
        B[200:200 + len(top)] = top
        dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
        want = cf.ew(coracle.MUL, A, B)
        assert (ctx.mul(dA, dB).to_numpy() == want).all(), (deg, hex(mod))
        ctx.mul(dA, dB, out=dA)                                    # in place
        assert (dA.to_numpy() == want).all(), deg


def test_small_binary_fields_every_degree(eng, coracle):
    """GF(2^n), 9 <= n <= 32 on four-byte storage (GF2W32, round 6): every degree, the first irreducible (sparse: fold
    reduction) and one with a long low part (long division): products at extreme and random operands, in place, the share
    generation / recombination round trip with t = 2, k = 3 and the all-ones coefficients of the runtime (plain XOR of the
    rows), inverse -- against the oracle."""
    from mpyc_amd.gfpx import BinaryPolynomial
    for deg in range(9, 33):
        for mod in (int(BinaryPolynomial.next_irreducible(1 << deg)),
                    int(BinaryPolynomial.next_irreducible((1 << deg) | (1 << (deg - 1)) | (1 << (deg - 2))))):
            F = po.Field(mod, True)
            ctx = ctx_for(eng, mod, True)
            eb = ctx.elem_bytes
            assert eb == 4 and F.n == deg
            cf = coracle.CField(mod, True)
            n = 8192 + 1027
            A, B = rand_np(F, eb, n, 300 + deg), rand_np(F, eb, n, 400 + deg)
            top = pack([F.order - 1, F.order >> 1, (F.order >> 1) | 1, 7 << (deg - 3), 5 << (deg - 3), 1 << (deg - 1), 1, 0], eb)
            A[100:100 + len(top)] = top
            B[100:100 + len(top)] = top[::-1]
            A[200:200 + len(top)] = top
            B[200:200 + len(top)] = top
            dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
            want = cf.ew(coracle.MUL, A, B)
            assert (ctx.mul(dA, dB).to_numpy() == want).all(), (deg, hex(mod))
            assert (ctx.add(dA, dB).to_numpy() == cf.ew(coracle.ADD, A, B)).all()
            tmp = dA.clone()
            ctx.mul(tmp, dB, out=tmp)
            assert (tmp.to_numpy() == want).all(), deg
            t, m = 2, 5
            C = np.stack([rand_np(F, eb, n, 500 + deg + j) for j in range(t)])
            sh = ctx.split(dA, ctx.matrix_from_numpy(C), t, m)
            assert (sh.to_numpy() == cf.split(A, C, t, m)).all(), (deg, 'split')
            xs = [2, 4, 5]
            lam = po.recombination_vector(F, xs, 0)
            assert (ctx.recombine([sh.row(x - 1) for x in xs], lam).to_numpy() == A).all(), (deg, 'recombine')
            ones = ctx.recombine([sh.row(0), sh.row(1), sh.row(2)], [1, 1, 1]).to_numpy()
            assert (ones == cf.recombine([sh.to_numpy()[j] for j in range(3)], [1, 1, 1])).all()
            inv = ctx.inv(ctx.from_numpy(np.where(A == 0, 1, A).astype(A.dtype)))
            assert (ctx.mul(inv, ctx.from_numpy(np.where(A == 0, 1, A).astype(A.dtype))).to_numpy() == 1).all(), (deg, 'inv')


@pytest.mark.parametrize('modulus', [(1 << 64) | 0x1b, (1 << 40) | 0x39, (1 << 63) | 0x3])
def test_batched_inverse_full_batch_kernel_binary(eng, modulus, monkeypatch):
    """Large arrays over GF(2^n), 33 <= n <= 64 (one-word elements, q - 2 = 2^n - 3): these keep the round-3 kernel
    (k_inv_batch; the full-batch kernel is for prime fields) -- inverse property, zeros, a sample against the oracle."""
    from mpyc_amd.gfpx import BinaryPolynomial
    if not BinaryPolynomial.is_irreducible(modulus):
        modulus = int(BinaryPolynomial.next_irreducible(modulus))
    F = po.Field(modulus, True)
    ctx = ctx_for(eng, modulus, True)
    assert ctx.elem_bytes == 8
    n = 300_007
    A = rand_np(F, 8, n, 91)
    A[[0, 5, n // 2, n - 1]] = 0
    A[1], A[2] = 1, F.order - 1
    dA = ctx.from_numpy(A)
    with pytest.raises(ZeroDivisionError):
        ctx.inv(dA)
    inv = ctx.inv(dA, check_zero=False)
    nz = dA.t != 0
    assert bool((ctx.mul(dA, inv).t[nz] == 1).all()) and bool((inv.t[~nz] == 0).all())
    got = inv.to_numpy()
    for i in [1, 2, 3, 4, n - 2] + list(range(100, 140)):
        if int(A[i]):
            assert po.mul(F, int(got[i]), int(A[i])) == 1, (hex(modulus), i)
    part = ctx.inv(eng.DevArray(ctx, dA.t[:30_000].clone(), 30_000), check_zero=False)       # below the threshold: k_inv_batch
    assert torch.equal(part.t, inv.t[:30_000])


def test_gf2_64_bitsliced_product(eng, coracle, monkeypatch):
    """GF(2^64) with the default modulus x^64 + x^4 + x^3 + x + 1, n >= 2^21: the bit-sliced kernel
    (k_gf2w64_mul_bitsliced, misc.hip) multiplies the whole 2048-element slabs, the element-wise kernel the rest.
    Every element against the C oracle (gfpx.py:988-1045 restated), extreme operands in the first and the last slab and
    across the seam, in place, and equal to the multiplier kernel (FFGPU_GF2W_BITSLICED=0)."""
    mod = (1 << 64) | 0x1b
    F = po.Field(mod, True)
    ctx = ctx_for(eng, mod, True)
    cf = coracle.CField(mod, True)
    for n in ((1 << 21), (1 << 21) + 2048 * 3 + 5, 5_000_011):
        A, B = rand_np(F, 8, n, 41 + n % 7), rand_np(F, 8, n, 43 + n % 7)
        top = pack([F.order - 1, F.order >> 1, (F.order >> 1) | 1, 7 << 61, 5 << 61, 1 << 63, 0, 1, 2, 0x1b], 8)
        seam = n // 2048 * 2048
        for at in (0, 2048 * 7 + 100, seam - len(top) - 3, max(0, min(seam - 2, n - len(top))), n - len(top)):
            A[at:at + len(top)] = top
            B[at:at + len(top)] = top[::-1] if at % 2 else top
        dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
        want = cf.ew(coracle.MUL, A, B)
        got = ctx.mul(dA, dB)
        assert (got.to_numpy() == want).all(), n
        monkeypatch.setenv('FFGPU_GF2W_BITSLICED', '0')          # (switches are read when a context is created)
        plain = eng.FieldContext(mod, binary=True)
        monkeypatch.delenv('FFGPU_GF2W_BITSLICED')
        assert torch.equal(plain.mul(eng.DevArray(plain, dA.t, n), eng.DevArray(plain, dB.t, n)).t, got.t)
        ctx.mul(dA, dB, out=dA)                                    # in place
        assert (dA.to_numpy() == want).all(), n


def test_wide_binary_recombination_tables(eng, coracle):
    """GF(2^n), 9 <= n <= 128: recombination of large arrays goes through per-workgroup nibble tables of
    the Lagrange coefficients (k_gf2w_recombine_tab); same bits as the plain kernel and the oracle."""
    from mpyc_amd.gfpx import BinaryPolynomial
    for mod in ((1 << 128) | 0x87, (1 << 64) | 0x1b, int(BinaryPolynomial.next_irreducible(1 << 100)),
                int(BinaryPolynomial.next_irreducible(1 << 33)), int(BinaryPolynomial.next_irreducible(1 << 13))):
        F = po.Field(mod, True)
        ctx = ctx_for(eng, mod, True)
        eb = ctx.elem_bytes
        cf = coracle.CField(mod, True)
        n = 70001
        for k in (1, 3, 7, 9):
            rows = [rand_np(F, eb, n, 300 + j) for j in range(k)]
            rng = random.Random(k)
            lam = [rng.randrange(F.order) for _ in range(k)]
            lam[0] = F.order - 1
            got = ctx.recombine([ctx.from_numpy(r) for r in rows], lam).to_numpy()
            assert (got == cf.recombine(rows, lam)).all(), (hex(mod), k)
            small = ctx.recombine([ctx.from_numpy(r[:1000]) for r in rows], lam).to_numpy()     # plain kernel
            assert (small == got[:1000]).all()
        rows = [rand_np(F, eb, n, 400 + j) for j in range(3)]
        lam = [random.Random(9).randrange(F.order) for _ in range(6)]
        out = ctx.recombine([ctx.from_numpy(r) for r in rows], lam, w=2)
        assert (out.to_numpy() == cf.recombine(rows, lam, w=2)).all()
        # coefficients 1 are XORed in without a table, coefficients 0 dropped: all-ones (the parties 1..7 / 1..3 at
        # x = 0: thresha._recombination_vector gives [1, ..., 1] over GF(2^n)), mixes, and the all-zero vector
        rows = [rand_np(F, eb, n, 450 + j) for j in range(7)]
        dev_rows = [ctx.from_numpy(r) for r in rows]
        dense = random.Random(10).randrange(2, F.order)
        for lam in (po.recombination_vector(F, [1, 2, 3, 4, 5, 6, 7], 0), [1, 0, dense, 1, 0, 1, dense ^ 1], [0] * 7,
                    [dense, dense ^ 5, dense, 1, dense ^ 5, dense, 0], [dense] * 7,        # repeated coefficients: grouped rows
                    po.recombination_vector(F, [1, 2, 3, 4, 5], 0) + [0, 0],
                    [0, 0, 0, 0, 0, 0, 1], [dense] + [0] * 6):
            got = ctx.recombine(dev_rows, lam).to_numpy()
            assert (got == cf.recombine(rows, lam)).all(), (hex(mod), lam)
        assert po.recombination_vector(F, [1, 2, 3, 4, 5, 6, 7], 0) == [1] * 7
        assert po.recombination_vector(F, [3, 1, 2], 0) == [1] * 3


@pytest.mark.parametrize('modulus,binary', FIELDS)
def test_dot_and_sum_reductions(eng, modulus, binary):
    """Two-stage reduction kernels: inner product and sum, vs Python integers / the GF(2^n) oracle."""
    F = po.Field(modulus, binary)
    ctx = ctx_for(eng, modulus, binary)
    eb = ctx.elem_bytes
    for n in (0, 1, 2, 17, 255, 4099, 200003 if eb < 16 else 30011):
        A, B = rand_np(F, eb, n, 501), rand_np(F, eb, n, 502)
        a, b = unpack(A, eb), unpack(B, eb)
        dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
        if binary:
            want_dot, want_sum = 0, 0
            for x, y in zip(a, b):
                want_dot ^= po.mul(F, x, y)
                want_sum ^= x
        else:
            want_dot = sum(x * y for x, y in zip(a, b)) % modulus
            want_sum = sum(a) % modulus
        assert unpack(ctx.dot(dA, dB).to_numpy(), eb) == [want_dot], (hex(modulus), n)
        assert unpack(ctx.sum(dA).to_numpy(), eb) == [want_sum], (hex(modulus), n)
        if eb < 16 and n > 2:
            va, vb = eng.DevArray(ctx, dA.t[1:], n - 1), eng.DevArray(ctx, dB.t[1:], n - 1)
            if binary:
                w = 0
                for x, y in zip(a[1:], b[1:]):
                    w ^= po.mul(F, x, y)
            else:
                w = sum(x * y for x, y in zip(a[1:], b[1:])) % modulus
            assert unpack(ctx.dot(va, vb).to_numpy(), eb) == [w]


def test_group_matvec_aes_affine_and_from_bits(eng, golden_sbox):
    """Small public matrix over the last axis: (1) the AES S-box affine layer applied to SHARES of the
    8 bits of x^254 followed by np_from_bits equals the S-box table when the 'shares' are the bits
    themselves (m = 1), and stays linear on random GF(2^8) shares; (2) np_from_bits over a prime field;
    (3) random matrices vs Python for P61 / P128 / GF(2^8)."""
    F8 = po.Field(0x11b, True)
    ctx = ctx_for(eng, 0x11b, True)
    rows8, b8 = golden_sbox['rows8'], golden_sbox['b']
    A = [[(rows8[r] >> c) & 1 for c in range(8)] for r in range(8)]
    Bv = [(b8 >> r) & 1 for r in range(8)]
    inv = golden_sbox['pow254']
    bits = [(inv[v] >> j) & 1 for v in range(256) for j in range(8)]            # np_to_bits of x^254, shape (256, 8)
    y = ctx.group_matvec(ctx.from_numpy(np.array(bits, dtype=np.uint8)), A, Bv)  # A @ bits + B
    w = ctx.group_matvec(y, [[1 << j for j in range(8)]])                       # np_from_bits
    assert unpack(w.to_numpy(), 1) == golden_sbox['table']
    rng = random.Random(31)
    x = [rng.randrange(256) for _ in range(8 * 5000)]
    got = unpack(ctx.group_matvec(ctx.from_numpy(np.array(x, dtype=np.uint8)), A, Bv).to_numpy(), 1)
    want = []
    for i in range(5000):
        for r in range(8):
            acc = Bv[r]
            for c in range(8):
                if A[r][c]:
                    acc ^= x[8 * i + c]
            want.append(acc)
    assert got == want
    for modulus, binary in [(P61, False), (P128, False), (0x11b, True), (2**31 - 1, False)]:
        F = po.Field(modulus, binary)
        c2 = ctx_for(eng, modulus, binary)
        eb = c2.elem_bytes
        r_, g_, ng = 3, 5, 1203
        M = [[rng.randrange(F.order) for _ in range(g_)] for _ in range(r_)]
        bias = [rng.randrange(F.order) for _ in range(r_)]
        X = rand_np(F, eb, g_ * ng, 611)
        xs = unpack(X, eb)
        got = unpack(c2.group_matvec(c2.from_numpy(X), M, bias).to_numpy(), eb)
        want = []
        for i in range(ng):
            for a in range(r_):
                acc = bias[a]
                for c in range(g_):
                    acc = po.add(F, acc, po.mul(F, M[a][c], xs[i * g_ + c]))
                want.append(acc)
        assert got == want, hex(modulus)
        if not binary:                                                        # np_from_bits: sum_j x_j 2^j
            l = 16
            bitsv = [rng.randrange(F.order) for _ in range(l * 100)]
            got = unpack(c2.group_matvec(c2.from_numpy(pack(bitsv, eb)), [[1 << j for j in range(l)]]).to_numpy(), eb)
            assert got == [sum(bitsv[i * l + j] << j for j in range(l)) % modulus for i in range(100)]


def test_gauss_golden_and_random(eng):
    """ffgpu_gauss vs the reference's np.linalg.det / inv / solve (tests/golden/linalg.json: matrices
    that need row swaps, singular, all-zero, batched det) and vs the oracle on larger random systems."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'linalg.json')))
    for name, fc in g.items():
        modulus, binary = int(fc['modulus'], 16), fc['binary']
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        ux = lambda m: [[int(v, 16) for v in row] for row in m]
        red = lambda v: int(v, 16) if binary else int(v, 16) % modulus
        for c in fc['cases']:
            n = c['n']
            A, B = ux(c['A']), ux(c['B'])
            d, sing = ctx.gauss(dev(ctx, [v for row in A for v in row]), n, n, 1, det=True)
            assert host(d) == [red(c['det'])], (name, c['kind'])
            assert int(sing[0]) == (1 if 'error' in c else 0)
            eye = [[int(i == j) for j in range(n)] for i in range(n)]
            aug = dev(ctx, [v for ra, rb, re in zip(A, B, eye) for v in ra + rb + re])
            _, sing = ctx.gauss(aug, n, 2 * n + 2, 1)
            if 'error' in c:
                assert int(sing[0]) == 1
                continue
            assert int(sing[0]) == 0
            rows = [host(aug)[i * (2 * n + 2):(i + 1) * (2 * n + 2)] for i in range(n)]
            assert [r[n:n + 2] for r in rows] == [[red(v) for v in row] for row in c['solve']], (name, c['kind'])
            assert [r[n + 2:] for r in rows] == [[red(v) for v in row] for row in c['inv']], (name, c['kind'])
        flat = [int(v, 16) for m in fc['stack'] for row in m for v in row]
        d, sing = ctx.gauss(dev(ctx, flat), 3, 3, 4, det=True)
        assert host(d) == [red(v) for row in fc['stack_det'] for v in row], name
        assert [int(v) for v in sing.cpu()] == [int(v == 0) for v in host(d)] and int(sing[3]) == 1
    rng = random.Random(99)
    for modulus, binary, n in [(P61, False, 40), (P128, False, 17), (0x11b, True, 33), (2**31 - 1, False, 64),
                               (6616326157076047771, False, 21)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        A = [[rng.randrange(F.order) for _ in range(n)] for _ in range(n)]
        for i in range(0, n, 5):
            A[i][i] = 0
        for i in range(3):
            A[i][0] = 0
        B = [[rng.randrange(F.order) for _ in range(3)] for _ in range(n)]
        aug = dev(ctx, [v for ra, rb in zip(A, B) for v in ra + rb])
        _, sing = ctx.gauss(aug, n, n + 3, 1)
        assert int(sing[0]) == 0
        want = po.gauss_solve(F, A, B)
        got = host(aug)
        assert [got[i * (n + 3) + n:(i + 1) * (n + 3)] for i in range(n)] == want, hex(modulus)
        d, _ = ctx.gauss(dev(ctx, [v for row in A for v in row]), n, n, 1, det=True)
        assert host(d) == [po.gauss_det(F, A)]


def test_beaver_combine(eng, coracle):
    """Beaver multiplication vs GRR resharing on the same inputs: both open to a*b.  (Parity UNPINNED:
    the reference has no Beaver triples; this checks the textbook identity only.)  Three simulated
    parties, t = 1, every party's shares on this one GPU."""
    for modulus in (P61, P128):
        F = po.Field(modulus)
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus)
        n, t, m = 5003, 1, 3
        A, B, X, Y = (rand_np(F, eb, n, s) for s in (601, 602, 603, 604))
        Z = cf.ew(coracle.MUL, X, Y)
        def share(V, seed):
            C = rand_np(F, eb, n, seed).reshape(lshape(eb, 1, n))
            return ctx.split(ctx.from_numpy(V), ctx.matrix_from_numpy(C), t, m)
        sa, sb, sx, sy, sz = (share(V, 700 + i) for i, V in enumerate((A, B, X, Y, Z)))
        lam2 = po.recombination_vector(F, [1, 2], 0)
        # open d = a - x and e = b - y (public)
        dsh = [ctx.sub(sa.row(i), sx.row(i)) for i in range(m)]
        esh = [ctx.sub(sb.row(i), sy.row(i)) for i in range(m)]
        d = ctx.recombine(dsh[:2], lam2)
        e = ctx.recombine(esh[:2], lam2)
        assert (d.to_numpy() == cf.ew(coracle.SUB, A, X)).all()
        # every party combines locally; party 0 adds the public d*e term ... as a SHARE of a public value the
        # constant polynomial d*e has the same share for every party, so every party adds it here
        csh = [ctx.beaver_combine(sz.row(i), sx.row(i), sy.row(i), d, e, True) for i in range(m)]
        c = ctx.recombine(csh[:2], lam2)
        want = cf.ew(coracle.MUL, A, B)
        assert (c.to_numpy() == want).all(), modulus
        c23 = ctx.recombine(csh[1:], po.recombination_vector(F, [2, 3], 0))
        assert (c23.to_numpy() == want).all()
        # the GRR route on the same shares: local products (degree 2t) recombined from 2t+1 parties
        prod = [ctx.mul(sa.row(i), sb.row(i)) for i in range(m)]
        assert (ctx.recombine(prod, po.recombination_vector(F, [1, 2, 3], 0)).to_numpy() == want).all()
        # add_de = False omits the d*e term
        nod = ctx.beaver_combine(sz.row(0), sx.row(0), sy.row(0), d, e, False)
        de = ctx.mul(d, e)
        assert (ctx.add(nod, de).to_numpy() == csh[0].to_numpy()).all()


def test_hip_graph_capture_of_a_gate(eng, coracle):
    """A gate (fused local product + share generation, then recombination) captured once into a HIP graph
    and replayed on fresh inputs gives the same bits as eager launches."""
    F = po.Field(P61)
    ctx = ctx_for(eng, P61, False)
    cf = coracle.CField(P61)
    n, t, m = 4096, 1, 3
    a, b = ctx.from_numpy(rand_np(F, 8, n, 1)), ctx.from_numpy(rand_np(F, 8, n, 2))
    coef = ctx.matrix_from_numpy(rand_np(F, 8, n, 3).reshape(1, n))
    shares, y = ctx.empty_matrix(m, n), ctx.empty(n)
    lam = po.recombination_vector(F, [1, 2, 3], 0)
    rec = ctx.recombine_plan([shares.row(j) for j in range(3)], lam, y)

    def gate():
        ctx.split(a, coef, t, m, out=shares, mul_by=b)
        rec()

    g = eng.CapturedLaunches(gate)
    for seed in (10, 11, 12):
        A, B = rand_np(F, 8, n, seed), rand_np(F, 8, n, seed + 100)
        a.t.copy_(ctx.from_numpy(A).t)
        b.t.copy_(ctx.from_numpy(B).t)
        y.t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (y.to_numpy() == cf.ew(coracle.MUL, A, B)).all(), seed


def test_matmul(eng, coracle):
    """Dense product over the field (finfields.py:1126-1135): golden matrices from the reference,
    then ragged / large-K shapes against the oracle (K > 192 exercises the accumulator flush)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'matmul.json')) as fh:
        gold = json.load(fh)
    for name, case in gold.items():
        mod = int(case['modulus'], 16)
        ctx = ctx_for(eng, mod, case['binary'])
        eb = ctx.elem_bytes
        for c in case['cases']:
            M, K, N = c['M'], c['K'], c['N']
            A = pack([int(v, 16) for r in c['A'] for v in r], eb)
            B = pack([int(v, 16) for r in c['B'] for v in r], eb)
            got = unpack(ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy(), eb)
            assert got == [int(v, 16) for r in c['C'] for v in r], (name, M, K, N)
    for modulus, binary in [(P61, False), (P64, False), (P128, False), (6616326157076047771, False), (2**31 - 1, False),
                            (258797994007609146293811961253269568351, False), (0x11b, True), ((1 << 64) | 0x1b, True),
                            ((1 << 128) | 0x87, True), (0x1002b, True), (0x10000008d, True)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        slow = eb >= 12
        for (M, K, N) in [(1, 1, 1), (64, 64, 64), (70, 130, 65), (3, 500, 5), (33, 17, 129)] if not slow else \
                [(1, 1, 1), (33, 40, 35), (3, 300, 5)]:
            A, B = rand_np(F, eb, M * K, 81), rand_np(F, eb, K * N, 82)
            if not binary:
                A[:K] = pack([F.order - 1] * K, eb)             # worst-case accumulation in row 0
                B[::N] = pack([F.order - 1] * K, eb)            # ... column 0
            got = ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy()
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A, B, M, K, N)
            coracle.set_threads(1)
            assert (got == want).all(), (hex(modulus), M, K, N)


def test_skinny_products(eng, coracle):
    """Matrix x few columns and few rows x matrix (the np_bnnmnist shape, demos/np_bnnmnist.py:10-15) take
    dedicated HBM-bound kernels: one output dimension <= 8, the other >= 64.  Against the oracle for every
    reduction strategy, with worst-case accumulation (all operands p-1), K beyond the 192-term accumulator
    flush, ragged sizes, and sub-matrix views with a leading dimension larger than the row length.  Few rows x matrix over
    one-word primes is k_vecmat_partial_col (column sums of 22 x 32-bit partial products, one instantiation per M, tiles of 128
    staged rows, groups of rows of B in flight); other fields k_vecmat_partial."""
    cases = [(200, 333, 1), (100, 257, 3), (64, 1000, 8), (77, 5, 2), (1, 500, 300), (3, 1000, 129), (8, 2100, 64), (2, 7, 1000),
             (64, 300, 200), (40, 1000, 90), (130, 2000, 70), (9, 129, 9),      # these four: tiled kernel with split-K
             (5000, 7, 1), (3000, 32, 3),                                        # short rows: one thread per row
             (4096, 1024, 1), (4101, 1030, 2), (8192, 2050, 1),                  # several rows per workgroup (k_matvec_rows_r)
             (1, 4096, 4096), (2, 300, 4098), (5, 1111, 640), (8, 4500, 300), (7, 129, 70)]   # K split over workgroups, several staged tiles
    for modulus, binary in [(P61, False), (P64, False), (2**96 - 17, False), (P128, False), (6616326157076047771, False),
                            (2**31 - 1, False), (258797994007609146293811961253269568351, False),
                            ((1 << 64) | 0x1b, True), ((1 << 128) | 0x87, True), (0x10000008d, True)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        for (M, K, N) in cases if eb < 12 else cases[:2] + cases[4:6] + cases[8:10] + cases[12:13] + cases[14:15] + cases[17:18]:
            A, B = rand_np(F, eb, M * K, 91), rand_np(F, eb, K * N, 92)
            if not binary:
                A[:K] = pack([F.order - 1] * K, eb)
                B[::N] = pack([F.order - 1] * K, eb)
            got = ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy()
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A, B, M, K, N)
            coracle.set_threads(1)
            assert (got == want).all(), (hex(modulus), M, K, N)
    # every operand p - 1, every M of the few-rows kernels (one instantiation each for one-word primes: column sums of
    # 22 x 32-bit partial products, flushed every 192 terms), packed and ragged column counts
    # (round 6: the multi-limb 2^k - c primes accumulate in 28-bit digits, flushed every 32 terms: the same at their extremes)
    for modulus in (P61, P64, 6616326157076047771, 2**40 - 87, 2**80 - 65, 2**96 - 17, P128, 2**136 - 113):
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        if eb > 8:
            rng_ = random.Random(eb)
            for (M, K, N) in ((3, 97, 70), (8, 1000, 129), (6, 33, 300)):           # random operands, ragged K around the flush
                a = [rng_.randrange(modulus) for _ in range(M * K)]
                b = [rng_.randrange(modulus) for _ in range(K * N)]
                got = unpack(ctx.matmul(ctx.from_numpy(pack(a, eb)), ctx.from_numpy(pack(b, eb)), M, K, N).to_numpy(), eb)
                assert got == [sum(a[i * K + k] * b[k * N + j] for k in range(K)) % modulus for i in range(M) for j in range(N)], \
                    (hex(modulus), M, K, N)
        for M in range(1, 9):
            for (K, N) in ((1000 + M, 130), (389, 261)):
                A = pack([modulus - 1] * (M * K), eb)
                B = pack([modulus - 1] * (K * N), eb)
                got = unpack(ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy(), eb)
                assert got == [K * (modulus - 1) ** 2 % modulus] * (M * N), (hex(modulus), M, K, N)
        # matrix x few columns (k_matvec_rows_col: one instantiation per N, one or two rows per workgroup, the last workgroup
        # ragged), K beyond one flush of a thread's column sums (192 terms x 256 threads)
        for N in range(2, 9):
            for (M, K) in ((71, 1000 + N), (130, 389)) + (((64, 60001),) if N in (3, 8) else ()):
                A = pack([modulus - 1] * (M * K), eb)
                B = pack([modulus - 1] * (K * N), eb)
                got = unpack(ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy(), eb)
                assert got == [K * (modulus - 1) ** 2 % modulus] * (M * N), (hex(modulus), M, K, N)
    # through the mirror: 2-D @ 1-D, 1-D @ 2-D and a batch of two rows
    from mpyc_amd import finfields
    Fm = finfields.GF(P61)
    rng = random.Random(8)
    W_ = [[rng.randrange(P61) for _ in range(300)] for _ in range(150)]
    x = [rng.randrange(P61) for _ in range(150)]
    y = [rng.randrange(P61) for _ in range(300)]
    got = Fm.array(x) @ Fm.array(W_)
    assert [int(v) for v in got.value] == [sum(x[k] * W_[k][j] for k in range(150)) % P61 for j in range(300)]
    got = Fm.array(W_) @ Fm.array(y)
    assert [int(v) for v in got.value] == [sum(W_[i][k] * y[k] for k in range(300)) % P61 for i in range(150)]


def test_dense_product_in_digits_multi_limb_primes(eng):
    """Round 6: the LDS-tiled dense product stages the multi-limb 2^k - c primes as 28-bit digits and accumulates column sums,
    reduced every 32 terms (k_matmul with LazyDot): every operand p - 1 (the largest columns), K around the flush and the
    k-step, split-K shapes, against Python integers -- the 80-, 96-, 128- and 136-bit primes."""
    rng = random.Random(2806)
    for modulus in (2**80 - 65, 2**96 - 17, P128, 2**136 - 113):
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        for (M, K, N) in ((33, 31, 35), (40, 32, 33), (20, 33, 40), (17, 200, 19), (64, 257, 48)):
            A = pack([modulus - 1] * (M * K), eb)
            B = pack([modulus - 1] * (K * N), eb)
            got = unpack(ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy(), eb)
            assert got == [K * (modulus - 1) ** 2 % modulus] * (M * N), (hex(modulus), M, K, N)
            a = [rng.randrange(modulus) for _ in range(M * K)]
            b = [rng.randrange(modulus) for _ in range(K * N)]
            got = unpack(ctx.matmul(ctx.from_numpy(pack(a, eb)), ctx.from_numpy(pack(b, eb)), M, K, N).to_numpy(), eb)
            assert got == [sum(a[i * K + k] * b[k * N + j] for k in range(K)) % modulus for i in range(M) for j in range(N)], \
                (hex(modulus), M, K, N)


def test_matrix_core_product(eng, coracle):
    """Large dense products over primes of up to 64 bits run as int8 limb GEMMs on the matrix cores
    (k_limb_gemm): bit-exact against the oracle for every limb count (5: 32-bit storage, 9: moduli below 2^63,
    10: 64-bit moduli), ragged shapes (padding), K beyond one 8192 chunk (accumulating launches), worst-case
    operands (p - 1 everywhere in a row and a column) and sub-matrix views; and equal to the VALU kernel
    (FFGPU_MM_MFMA=0 in a fresh context is not needed: the small shapes of test_matmul take that path)."""
    shapes = [(256, 300, 257), (65, 8300, 70), (130, 64, 2000), (64, 1030, 300),     # the last one: split-K slabs
              (16, 2100, 520), (9, 4100, 450), (300, 2000, 33), (40, 1000, 410)]     # 9..63 rows / columns: tiles padded to 64
    for modulus, binary in [(P61, False), (P64, False), (6616326157076047771, False), (2**31 - 1, False), (2**40 - 87, False),
                            (65537, False)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        for (M, K, N) in shapes:
            A, B = rand_np(F, eb, M * K, 71), rand_np(F, eb, K * N, 72)
            A[:K] = pack([F.order - 1] * K, eb)
            B[::N] = pack([F.order - 1] * K, eb)
            got = ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy()
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A, B, M, K, N)
            coracle.set_threads(1)
            assert (got == want).all(), (hex(modulus), M, K, N)
    # operands cycling through the values where the signed-digit representative switches (0x7f7f..7f and its
    # neighbours), p/2, and the ends of the range, for moduli at the top of each storage width
    for modulus in (P64, 2**64 - 59, 2**63 - 25, 2**32 - 5, 2**31 - 1, P61):
        F = po.Field(modulus, False)
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, False)
        T8 = 0x7f7f7f7f7f7f7f7f if eb == 8 else 0x7f7f7f7f
        edge = sorted({v % modulus for v in (0, 1, 127, 128, 255, 256, T8 - 1, T8, T8 + 1, T8 + 2, modulus // 2, modulus // 2 + 1,
                                              modulus - 1, modulus - 2, modulus - 128, modulus - 129, 2**31, 2**63 % modulus,
                                              0x8080808080808080 % modulus, 0x80 << 24)})
        # 256 rows: digit planes of both operands; 64 rows: the product kernel converts B itself (k_limb_gemm_glds<BRAW>)
        for (M, K, N) in ((256, 256, 256), (64, 1024, 256)):
            rng = random.Random(modulus % 1000)
            a = [edge[(i * 7 + k_ * 3) % len(edge)] if (i + k_) % 3 else rng.randrange(modulus) for i in range(M) for k_ in range(K)]
            b = [edge[(k_ * 5 + j) % len(edge)] if (j + k_) % 4 else rng.randrange(modulus) for k_ in range(K) for j in range(N)]
            A, B = pack(a, eb), pack(b, eb)
            got = ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy()
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A, B, M, K, N)
            coracle.set_threads(1)
            assert (got == want).all(), (hex(modulus), M, K, N)
    # all-(p-1) operands: every limb product at its maximum
    ctx = ctx_for(eng, P64, False)
    M = K = N = 256
    ones = pack([P64 - 1] * (M * K), 8)
    got = unpack(ctx.matmul(ctx.from_numpy(ones), ctx.from_numpy(ones), M, K, N).to_numpy(), 8)
    assert set(got) == {K * (P64 - 1) * (P64 - 1) % P64}


def test_matrix_core_product_equals_valu_product(eng, monkeypatch):
    """k_limb_gemm_glds (operand tiles streamed straight into LDS, two k-steps ahead, counted waits) against the VALU product
    (k_matmul: a context created with FFGPU_MM_MFMA=0 -- the library's switches are read per context): identical arrays for
    digit planes of both operands, for the raw right operand of the 64-row shapes (whole tiles, and ragged ones that go
    through the planes), split-K slabs, K beyond one 8192 chunk (accumulating launches) and one, two and three k-steps
    (the prologue / drain of the three-stage ring); 32-bit storage (k_limb_gemm_l4) as well."""
    for modulus in (P61, P64, 2**63 - 25, 2**31 - 1):
        F = po.Field(modulus, False)
        ctx = ctx_for(eng, modulus, False)
        monkeypatch.setenv('FFGPU_MM_MFMA', '0')
        valu = eng.FieldContext(modulus)
        monkeypatch.delenv('FFGPU_MM_MFMA')
        eb = ctx.elem_bytes
        # (tests/conftest.py lowers the matrix-core threshold to 1.6e7 multiply-accumulates)
        for (M, K, N) in ((256, 256, 256), (64, 1024, 256), (64, 4096, 512), (1024, 64, 512), (512, 96, 512), (128, 128, 1024),
                          (65, 8300, 70), (64, 1030, 300), (300, 9000, 130), (64, 8192 + 64, 128), (128, 8192 + 96, 64)):
            A, B = rand_np(F, eb, M * K, 5 + M), rand_np(F, eb, K * N, 7 + N)
            dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
            got = ctx.matmul(dA, dB, M, K, N)
            want = valu.matmul(eng.DevArray(valu, dA.t, M * K), eng.DevArray(valu, dB.t, K * N), M, K, N)
            assert torch.equal(got.t, want.t), (hex(modulus), M, K, N)


def test_matrix_core_product_two_limb_primes(eng, coracle):
    """Primes of 65..128 bits: 12 / 16 signed digits, the diagonals in 2 / 3 passes (k_limb_gemm_wide), K beyond one
    4096 chunk, worst-case operands, every two-limb reduction strategy (pseudo-Mersenne k = 128 and k < 128,
    Montgomery, 12-byte storage), and operands at the representative switch 0x7f7f..7f."""
    for modulus in (P128, 2**127 - 1, 258797994007609146293811961253269568351, 2**96 - 17, 2**80 - 65):
        F = po.Field(modulus, False)
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, False)
        nb = 12 if eb == 12 else 16
        T = int.from_bytes(b'\x7f' * nb, 'little')
        edge = sorted({v % modulus for v in (0, 1, T - 1, T, T + 1, modulus // 2, modulus // 2 + 1, modulus - 1, modulus - 2,
                                              2**64 - 1, 2**64, 2**127 % modulus, modulus - 128)})
        for (M, K, N) in ((128, 1000, 130), (65, 4200, 70), (20, 2100, 400), (700, 1300, 19)):
            A, B = rand_np(F, eb, M * K, 61), rand_np(F, eb, K * N, 62)
            a = unpack(A, eb)
            b = unpack(B, eb)
            for i in range(K):
                a[i] = modulus - 1                                   # row 0 of A and column 0 of B: worst case
                b[i * N] = modulus - 1
            for i in range(len(edge)):
                a[K + i] = edge[i]                                   # row 1 of A: the edge values
                b[N * i + 1] = edge[(3 * i) % len(edge)]             # column 1 of B
            A, B = pack(a, eb), pack(b, eb)
            got = ctx.matmul(ctx.from_numpy(A), ctx.from_numpy(B), M, K, N).to_numpy()
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A, B, M, K, N)
            coracle.set_threads(1)
            assert (got == want).all(), (hex(modulus), M, K, N)


def test_matmul_leading_dimensions(eng, coracle):
    """ffgpu_matmul on sub-matrix views: lda > K, ldb > N, ldc > N (the C ABI takes leading dimensions; the
    engine wrapper always passes contiguous operands), through every kernel family: skinny matvec / vecmat,
    tiled, split-K, and the matrix-core product whose split kernels read the strided operands."""
    from mpyc_amd import _ffi
    for modulus in (P61, P64, 2**31 - 1, P128):
        F = po.Field(modulus, False)
        ctx = ctx_for(eng, modulus, False)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, False)
        shapes = [(70, 33, 1), (1, 200, 130), (40, 50, 45), (9, 300, 20), (260, 270, 250), (70, 300, 3), (130, 1000, 8), (5, 700, 200)] if eb < 12 else \
            [(70, 33, 1), (1, 200, 130), (20, 50, 25)]
        for (M, K, N) in shapes:
            lda, ldb, ldc = K + 5, N + 3, N + 7
            Abig, Bbig = rand_np(F, eb, M * lda, 5), rand_np(F, eb, K * ldb, 6)
            dA, dB = ctx.from_numpy(Abig), ctx.from_numpy(Bbig)
            dC = ctx.from_numpy(rand_np(F, eb, M * ldc, 7))                   # pre-filled: the padding must survive
            before = dC.to_numpy().copy()
            _ffi.check(ctx._L.ffgpu_matmul(ctx._h, dA.ptr, lda, dB.ptr, ldb, dC.ptr, ldc, M, K, N, ctx._stream()), 'matmul')
            got = dC.to_numpy().reshape((M, ldc) + ((2,) if eb == 16 else ()))
            A = np.ascontiguousarray(Abig.reshape((M, lda) + ((2,) if eb == 16 else ()))[:, :K])
            B = np.ascontiguousarray(Bbig.reshape((K, ldb) + ((2,) if eb == 16 else ()))[:, :N])
            coracle.set_threads(coracle.max_threads())
            want = coracle.matmul(cf, A.reshape((M * K,) + ((2,) if eb == 16 else ())), B.reshape((K * N,) + ((2,) if eb == 16 else ())), M, K, N)
            coracle.set_threads(1)
            assert (got[:, :N].reshape(want.shape) == want).all(), (hex(modulus), M, K, N)
            keep = before.reshape(got.shape)[:, N:]
            assert (got[:, N:] == keep).all(), ('padding overwritten', hex(modulus), M, K, N)


def test_gf2n_table_multiplication(eng, coracle):
    """Large GF(2^n<=8) arrays multiply through log/antilog tables in LDS (k_gf8_mul_tab): same
    answers as the shift-xor kernel and the oracle, for every small binary field, all 256x256 pairs."""
    for mod in (0x11b, 0b10011, 0b111, 0b11, 0b1011, 0x12b):
        F = po.Field(mod, True)
        ctx = ctx_for(eng, mod, True)
        cf = coracle.CField(mod, True)
        q = F.order
        n = 1 << 19
        rng = np.random.default_rng(mod)
        A = rng.integers(0, q, size=n, dtype=np.uint8)
        B = rng.integers(0, q, size=n, dtype=np.uint8)
        pairs = np.array([(x, y) for x in range(q) for y in range(q)], dtype=np.uint8)
        A[:len(pairs)] = pairs[:, 0]
        B[:len(pairs)] = pairs[:, 1]
        for nn in (n, n - 5):                       # aligned + ragged tail
            got = ctx.mul(ctx.from_numpy(A[:nn]), ctx.from_numpy(B[:nn])).to_numpy()
            assert (got == cf.ew(coracle.MUL, A[:nn], B[:nn])).all(), (hex(mod), nn)
        small = ctx.mul(ctx.from_numpy(A[:4096]), ctx.from_numpy(B[:4096])).to_numpy()   # shift-xor path
        assert (small == got[:4096]).all()


KEY = bytes(range(100, 132))


@pytest.mark.parametrize('modulus,binary', FIELDS)
def test_device_rng(eng, coracle, modulus, binary):
    """On-device CSPRNG: coefficient matrix == oracle restatement of ChaCha + sampler + layout;
    fused split_rng == split(materialised coefficients), incl. tails, unaligned input, t > 4."""
    F = po.Field(modulus, binary)
    ctx = ctx_for(eng, modulus, binary)
    eb = ctx.elem_bytes
    cf = coracle.CField(modulus, binary)
    for (t, m, n, rounds, nonce) in [(1, 3, 4099, 20, 0), (3, 7, 2051, 20, 0xabcdef0123456789), (2, 5, 1025, 12, 3),
                                     (4, 9, 517, 8, 9), (6, 13, 300, 20, 0xffffffff00000001)]:
        if m >= F.order:
            continue
        C = ctx.rng_coeffs(KEY, nonce, t, n, rounds)
        want = coracle.rng_coeffs(cf, KEY, nonce, rounds, t, n)
        assert (C.to_numpy() == want).all(), (t, n)
        S, B = rand_np(F, eb, n, 61), rand_np(F, eb, n, 62)
        dS, dB = ctx.from_numpy(S), ctx.from_numpy(B)
        ref = ctx.split(dS, C, t, m)
        got = ctx.split_rng(dS, t, m, key=KEY, nonce=nonce, rounds=rounds)
        assert (got.to_numpy() == ref.to_numpy()).all(), (t, m, n)
        assert (got.to_numpy() == cf.split(S, want, t, m)).all(), (t, m, n)
        fused = ctx.split_rng(dS, t, m, key=KEY, nonce=nonce, rounds=rounds, mul_by=dB)
        assert (fused.to_numpy() == ctx.split(dS, C, t, m, mul_by=dB).to_numpy()).all(), (t, m, n)
        if eb < 16:
            # unaligned secrets pointer -> scalar path draws the same coefficients per element
            k_ = n - 1
            va = eng.DevArray(ctx, dS.t[1:], k_)
            got_u = ctx.split_rng(va, t, m, key=KEY, nonce=nonce, rounds=rounds)
            Ck = ctx.rng_coeffs(KEY, nonce, t, k_, rounds)
            assert (got_u.to_numpy() == ctx.split(va, Ck, t, m).to_numpy()).all(), (t, m, 'unaligned')
    # fresh key per call by default: two calls differ, both recombine to the secret
    if F.order > 3:
        n = 1000
        S = rand_np(F, eb, n, 63)
        dS = ctx.from_numpy(S)
        a, b = ctx.split_rng(dS, 1, 3), ctx.split_rng(dS, 1, 3)
        assert not (a.to_numpy() == b.to_numpy()).all()
        lam = po.recombination_vector(F, [1, 2], 0)
        for sh in (a, b):
            assert (ctx.recombine([sh.row(0), sh.row(1)], lam).to_numpy() == S).all()


def test_device_rng_rejection_branch(eng, coracle):
    """GPU vs oracle on the pseudo-Mersenne prime with the highest admissible rejection rate
    (p = 2^33 - c, c ~ 2^16): ~2^-17 of the samples take the spare path."""
    from mpyc_amd.finfields import is_prime
    c = 65535
    while not is_prime(2**33 - c):
        c -= 2
    p = 2**33 - c
    ctx = ctx_for(eng, p, False)
    assert ctx.reduction == 'pseudo-mersenne'
    cf = coracle.CField(p)
    n = 1 << 20
    for t in (1, 3):
        C = ctx.rng_coeffs(KEY, 5, t, n, rounds=8)
        assert (C.to_numpy() == coracle.rng_coeffs(cf, KEY, 5, 8, t, n)).all()
    S = rand_np(po.Field(p), 8, n, 5)
    dS = ctx.from_numpy(S)
    got = ctx.split_rng(dS, 3, 7, key=KEY, nonce=5, rounds=8)
    assert (got.to_numpy() == ctx.split(dS, ctx.rng_coeffs(KEY, 5, 3, n, rounds=8), 3, 7).to_numpy()).all()


def test_many_rows_and_outputs(eng, coracle):
    """k > 9 rows (generic kernel), w > 8 outputs, t > 4 (generic split)."""
    for modulus, binary in [(P64, False), (0x11b, True), (P128, False)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        n, t, m = 5003, 12, 25
        S = rand_np(F, eb, n, 41)
        Cn = rand_np(F, eb, t * n, 42).reshape(lshape(eb, t, n))
        sh = ctx.split(ctx.from_numpy(S), ctx.matrix_from_numpy(Cn), t, m)
        want = cf.split(S, Cn, t, m)
        assert (sh.to_numpy() == want).all()
        xs = list(range(2, 2 + t + 1))
        lam = po.recombination_vector(F, xs, 0)
        rec = ctx.recombine([sh.row(x - 1) for x in xs], lam)
        assert (rec.to_numpy() == S).all()
        # 11 recombination points at once from 3 rows of a degree-2 sharing
        Cn2 = rand_np(F, eb, 2 * n, 43).reshape(lshape(eb, 2, n))
        sh2 = ctx.split(ctx.from_numpy(S), ctx.matrix_from_numpy(Cn2), 2, 5)
        x_rs = list(range(0, 11)) if F.order > 11 else [0, 1, 2]
        xs = [1, 2, 3]
        lam = [v for xr in x_rs for v in po.recombination_vector(F, xs, xr)]
        out = ctx.recombine([sh2.row(x - 1) for x in xs], lam, w=len(x_rs))
        want2 = cf.recombine([cf.split(S, Cn2, 2, 5)[x - 1] for x in xs], lam, w=len(x_rs))
        assert (out.to_numpy() == want2).all()


def test_ragged_and_unaligned(eng, coracle):
    """Empty, single-element and 16-byte-misaligned inputs take the scalar path."""
    for modulus, binary in [(P61, False), (P128, False), (0x11b, True), (2**31 - 1, False)]:
        F = po.Field(modulus, binary)
        ctx = ctx_for(eng, modulus, binary)
        eb = ctx.elem_bytes
        cf = coracle.CField(modulus, binary)
        e = ctx.empty(0)
        assert ctx.mul(e, e).n == 0
        for n in (1, 2, 15, 17, 63, 257):
            A, B = rand_np(F, eb, n + 3, n), rand_np(F, eb, n + 3, n + 1)
            dA, dB = ctx.from_numpy(A), ctx.from_numpy(B)
            want = cf.ew(coracle.MUL, A, B)
            assert (ctx.mul(dA, dB).to_numpy() == want).all()
            if eb < 16:
                # views starting one element in: pointer not 16-byte aligned
                va = eng.DevArray(ctx, dA.t[1:1 + n], n)
                vb = eng.DevArray(ctx, dB.t[1:1 + n], n)
                out = ctx.mul(va, vb)
                assert (out.to_numpy() == want[1:1 + n]).all()
                sh = ctx.split(va, None, 0, 1)
                assert (sh.to_numpy()[0] == A[1:1 + n]).all()


# ---------------------------------------------------------------------------
# 3. BASELINE.json full sizes: 10^7 elements
# ---------------------------------------------------------------------------
N_FULL = 10_000_000


def test_full_size_modmul_p61(eng, coracle):
    """configs[1]: SecFld(GF(2^61-1)) array of 10^7 elements, element-wise modmul, bit-exact."""
    F = po.Field(P61)
    ctx = ctx_for(eng, P61, False)
    cf = coracle.CField(P61)
    A, B = rand_np(F, 8, N_FULL, 101), rand_np(F, 8, N_FULL, 102)
    got = ctx.mul(ctx.from_numpy(A), ctx.from_numpy(B)).to_numpy()
    coracle.set_threads(coracle.max_threads())
    want = cf.ew(coracle.MUL, A, B)
    coracle.set_threads(1)
    assert (got == want).all()
    assert int(got[12345]) == int(A[12345]) * int(B[12345]) % P61


def test_full_size_share_recombine_p64(eng):
    """configs[2]: m=7, t=3, 10^7 secrets, 64-bit prime: round trip, linearity, zero-coefficient."""
    F = po.Field(P64)
    ctx = ctx_for(eng, P64, False)
    t, m = 3, 7
    S1, S2 = rand_np(F, 8, N_FULL, 201), rand_np(F, 8, N_FULL, 202)
    C1 = rand_np(F, 8, t * N_FULL, 203).reshape(t, N_FULL)
    C2 = rand_np(F, 8, t * N_FULL, 204).reshape(t, N_FULL)
    dS1, dS2 = ctx.from_numpy(S1), ctx.from_numpy(S2)
    dC1, dC2 = ctx.matrix_from_numpy(C1), ctx.matrix_from_numpy(C2)
    sh1 = ctx.split(dS1, dC1, t, m)
    # round trip from any t+1 rows and from 2t+1 rows
    for xs in ([1, 2, 3, 4], [7, 5, 3, 1], [1, 2, 3, 4, 5, 6, 7]):
        lam = po.recombination_vector(F, xs, 0)
        rec = ctx.recombine([sh1.row(x - 1) for x in xs], lam)
        assert torch.equal(rec.t, dS1.t), xs
    # t rows are NOT enough to be the secret (degree really is t)
    lam3 = po.recombination_vector(F, [1, 2, 3], 0)
    assert not torch.equal(ctx.recombine([sh1.row(i) for i in range(3)], lam3).t, dS1.t)
    # linearity: split(s1+s2; c1+c2) == split(s1;c1) + split(s2;c2), row by row
    sh2 = ctx.split(dS2, dC2, t, m)
    dSs = ctx.add(dS1, dS2)
    dCs = ctx.empty_matrix(t, N_FULL)
    for j in range(t):
        ctx.add(dC1.row(j), dC2.row(j), out=dCs.row(j))
    shs = ctx.split(dSs, dCs, t, m)
    for i in range(m):
        assert torch.equal(ctx.add(sh1.row(i), sh2.row(i)).t, shs.row(i).t), i
    # spot-check a few columns against Python integers
    h = [0, 1, 2, 9_999_999, 5_000_001]
    for i in (0, 6):
        x = i + 1
        col = sh1.row(i).to_numpy()
        for hh in h:
            want = (int(S1[hh]) + sum(int(C1[j, hh]) * x**(j + 1) for j in range(t))) % P64
            assert int(col[hh]) == want


def test_more_than_2_to_32_elements(eng):
    """Index arithmetic beyond 32 bits: 2^32 + 1000 GF(2^8) elements (4.3 GB per array).  Checked on
    the device: the product of the whole array equals the products of its two halves, and spot values."""
    ctx = ctx_for(eng, 0x11b, True)
    n = (1 << 32) + 1000
    g = torch.Generator(device='cuda:0')
    g.manual_seed(5)
    a = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=g)
    b = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=g)
    A, B = eng.DevArray(ctx, a, n), eng.DevArray(ctx, b, n)
    full = ctx.mul(A, B)
    h = (1 << 31) + 16                                   # 16-byte aligned split point above 2^31
    lo = ctx.mul(eng.DevArray(ctx, a[:h], h), eng.DevArray(ctx, b[:h], h))
    hi = ctx.mul(eng.DevArray(ctx, a[h:], n - h), eng.DevArray(ctx, b[h:], n - h))
    assert torch.equal(full.t[:h], lo.t) and torch.equal(full.t[h:], hi.t)
    F = po.Field(0x11b, True)
    for i in (0, 1, h - 1, h, (1 << 32) - 1, 1 << 32, n - 1):
        assert int(full.t[i]) == po.mul(F, int(a[i]), int(b[i])), i
    s = ctx.split(A, None, 0, 1)                         # copy path (t = 0) over the same range
    assert torch.equal(s.row(0).t, a)
    del full, lo, hi, s, a, b
    torch.cuda.empty_cache()


def test_full_size_gate_p128(eng):
    """configs[3] shape on one GPU: 128-bit prime, gate = local product + reshare (m=7,t=3):
    recombining the 2t+1 re-shared rows at x=0 gives a*b."""
    F = po.Field(P128)
    ctx = ctx_for(eng, P128, False)
    n, t, m = 2_000_000, 3, 7
    A, B = rand_np(F, 16, n, 301), rand_np(F, 16, n, 302)
    C = rand_np(F, 16, t * n, 303).reshape(t, n, 2)
    dA, dB, dC = ctx.from_numpy(A), ctx.from_numpy(B), ctx.matrix_from_numpy(C)
    prod = ctx.mul(dA, dB)
    sh = ctx.split(dA, dC, t, m, mul_by=dB)
    xs = list(range(1, 2 * t + 2))
    rec = ctx.recombine([sh.row(x - 1) for x in xs], po.recombination_vector(F, xs, 0))
    assert torch.equal(rec.t, prod.t)
    pi = prod.to_numpy()
    for hh in (0, 1, n - 1, n // 2):
        a = int(A[hh, 0]) | (int(A[hh, 1]) << 64)
        b = int(B[hh, 0]) | (int(B[hh, 1]) << 64)
        assert (int(pi[hh, 0]) | (int(pi[hh, 1]) << 64)) == a * b % P128


def test_last_kernel_ms(eng):
    """Opt-in event timing of the most recent call (`last_kernel_ms`)."""
    from mpyc_amd import _ffi
    ctx = eng.FieldContext(P61, device=0)
    a = ctx.from_numpy(rand_np(po.Field(P61, False), 8, 1 << 22, 1))
    with pytest.raises(ValueError):
        ctx.last_kernel_ms()                                   # timing is off by default
    ctx.set_timing(True)
    out = ctx.mul(a, a)
    ms = ctx.last_kernel_ms()
    assert 0.0 < ms < 50.0
    big = ctx.mul(a, a, out=out)
    t1 = ctx.last_kernel_ms()
    small = eng.DevArray(ctx, a.t[:1024], 1024)
    ctx.mul(small, small)
    t2 = ctx.last_kernel_ms()
    assert t2 < t1                                             # 1024 elements take less than 4M
    ctx.set_timing(False)
    with pytest.raises(ValueError):
        ctx.last_kernel_ms()
