"""CPU check of the DEVICE arithmetic (mpyc_amd/csrc/fields.hpp compiled with g++) against
Python integers / the reference's golden outputs.  No GPU needed: this is how reduction
bugs are caught before spending GPU minutes.  The harness is test-only."""
import ctypes
import random

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle.coracle import elem_bytes
from fieldutil import cross, edge_values, field_of, pack, rand_values, unhex, unpack

HC_ADD, HC_SUB, HC_MUL, HC_NEG, HC_REDUCE, HC_MULADD, HC_MULADD_SMALL, HC_DOT, HC_SHARE, HC_LAZY, HC_COLDOT, HC_LDOT, HC_CHAIN = range(13)


def limbs3(x):
    return (ctypes.c_uint64 * 3)(*[(x >> (64 * i)) & (2**64 - 1) for i in range(3)])


def run(hc, F, op, a, b=None, c=None, x=0, lam=None, k=0, n=None, allow_rc=()):
    eb = elem_bytes(F.modulus, F.binary)
    A = pack(a, eb)
    n = n if n is not None else len(a)
    B = pack(b, eb) if b is not None else None
    C = pack(c, eb) if c is not None else None
    out = np.zeros_like(A[:n] if eb != 16 else A[:n])
    lam_arr = None
    if lam is not None:
        sl = 3 if eb == 24 else 2                        # limbs per host scalar (ffgpu_ctx_scalar_limbs)
        lam_arr = (ctypes.c_uint64 * (sl * len(lam)))()
        for i, v in enumerate(lam):
            for q in range(sl):
                lam_arr[sl * i + q] = (v >> (64 * q)) & (2**64 - 1)
    pk, ebo = ctypes.c_int(), ctypes.c_int()
    p = lambda z: z.ctypes.data_as(ctypes.c_void_p) if z is not None else None
    rc = hc.hc_run(int(F.binary), limbs3(F.modulus), 3, op, p(A), p(B), p(C), p(out), ctypes.c_size_t(n),
                   ctypes.c_uint32(x), lam_arr, k, ctypes.byref(pk), ctypes.byref(ebo))
    if rc in allow_rc:
        return None
    assert rc == 0, rc
    assert ebo.value == eb
    return unpack(out, eb), pk.value


def all_cases(golden_fields):
    return sorted(golden_fields)


def test_policy_selection(hostcheck, golden_fields):
    """Which reduction each modulus gets (policy_build.hpp)."""
    kinds = {}
    for name, case in golden_fields.items():
        F = field_of(case)
        _, pk = run(hostcheck, F, HC_ADD, [0], [0])
        kinds[name] = pk
    # PolicyKind enum order in policy_build.hpp
    PM64_MERS, PM64_K64, PM64_GEN, RC64, RC32, PM128_K128, PM128_GEN, PM96, MONT128, GF2P8, GF2W64, GF2W128 = range(1, 13)
    assert kinds['P61'] == PM64_MERS and kinds['P64'] == PM64_K64 and kinds['P40'] == PM64_GEN
    assert kinds['P63G'] == RC64 and kinds['P31'] == RC32 and kinds['GF19'] == RC32 and kinds['GF2'] == RC32
    assert kinds['P128'] == PM128_K128 and kinds['P127'] == PM128_GEN and kinds['P96'] == PM96
    assert kinds['P80'] == PM96 and kinds['P128G'] == MONT128 and kinds['P100G'] == MONT128
    assert kinds['GF2_8'] == GF2P8 and kinds['GF2_4'] == GF2P8 and kinds['GF2_1'] == GF2P8
    GF2W32 = 15                      # (round 6: appended to the enumeration, after the two three-limb prime policies)
    assert kinds['GF2_16'] == GF2W32 and kinds['GF2_64'] == GF2W64
    assert kinds['GF2_100'] == GF2W128 and kinds['GF2_128'] == GF2W128


def test_golden_elementwise(hostcheck, golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        a, b = unhex(case['a']), unhex(case['b'])
        for op, key in ((HC_ADD, 'add'), (HC_SUB, 'sub'), (HC_MUL, 'mul')):
            got, _ = run(hostcheck, F, op, a, b)
            assert got == unhex(case[key]), (name, key)
        got, _ = run(hostcheck, F, HC_NEG, a)
        assert got == unhex(case['neg']), name
        eb = elem_bytes(F.modulus, F.binary)
        if case['raw_width'] == 8 * eb:
            got, _ = run(hostcheck, F, HC_REDUCE, unhex(case['raw']))
            assert got == unhex(case['raw_reduced']), name


def test_edges_cross_product(hostcheck, golden_fields):
    """Every edge value against every edge value, + - * and fused multiply-add."""
    for name, case in golden_fields.items():
        F = field_of(case)
        ev = edge_values(F) + rand_values(F, 6, 1)
        a, b = cross(ev)
        for op, fn in ((HC_ADD, po.add), (HC_SUB, po.sub), (HC_MUL, po.mul)):
            got, _ = run(hostcheck, F, op, a, b)
            assert got == po.vec(fn, F, a, b), (name, op)
        c = list(reversed(a))
        got, _ = run(hostcheck, F, HC_MULADD, a, b, c)
        assert got == [po.add(F, po.mul(F, x, y), z) for x, y, z in zip(a, b, c)], name


def test_random_mul(hostcheck, golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        a, b = rand_values(F, 2000, 7), rand_values(F, 2000, 8)
        got, _ = run(hostcheck, F, HC_MUL, a, b)
        assert got == po.vec(po.mul, F, a, b), name


def test_muladd_small(hostcheck, golden_fields):
    """Horner step y*x + c with a public 32-bit x (party index)."""
    for name, case in golden_fields.items():
        F = field_of(case)
        ev = edge_values(F) + rand_values(F, 6, 2)
        y, c = cross(ev)
        xs = [1, 2, 3, 7, 255, 256, 65535, 2**31 - 1, 2**32 - 1]
        for x in xs:
            if F.binary and x >= F.order:
                continue
            got, _ = run(hostcheck, F, HC_MULADD_SMALL, y, None, c, x=x)
            xr = x if F.binary else x % F.modulus
            want = [po.add(F, po.mul(F, yy, xr), cc) for yy, cc in zip(y, c)]
            assert got == want, (name, x)


def test_dot(hostcheck, golden_fields):
    """Unreduced accumulation + single reduction (recombination inner loop)."""
    r = random.Random(5)
    for name, case in golden_fields.items():
        F = field_of(case)
        q = F.order
        for k in (1, 2, 3, 7, 9, 64, 255):
            n = 24
            extremes = [q - 1] * n
            rows = [extremes if j % 2 == 0 else rand_values(F, n, 100 + j) for j in range(k)]
            lam = [q - 1 if j % 3 == 0 else r.randrange(q) for j in range(k)]
            flat = [v for row in rows for v in row]
            got, _ = run(hostcheck, F, HC_DOT, flat, lam=lam, k=k, n=n)
            want = []
            for h in range(n):
                acc = 0
                for j in range(k):
                    acc = po.add(F, acc, po.mul(F, lam[j], rows[j][h]))
                want.append(acc)
            assert got == want, (name, k)


def test_share_evaluation_by_forward_differences(hostcheck, golden_fields):
    """share_x = s + sum_j C_j x^(j+1) reached by x steps of the forward-difference recurrence from f(0) = s
    (fields.hpp share_diff_init / share_diff_next -- the share loop of k_split), for every prime policy, t = 1..4,
    at extreme operands and up to 300 parties."""
    checked = 0
    for name, case in golden_fields.items():
        F = field_of(case)
        if F.binary:
            continue
        q = F.order
        n = 40
        ev = edge_values(F)
        for t in (1, 2, 3, 4):
            s = ([q - 1] * 4 + ev + rand_values(F, n, 3))[:n]
            rows = [([q - 1] * 4 + rand_values(F, n, 10 + j))[:n] for j in range(t)]
            flat = [v for row in rows for v in row]
            for party in (1, 2, 3, 7, 11, 300):
                if party >= q:
                    continue
                got, _ = run(hostcheck, F, HC_SHARE, s, None, flat, x=party, k=t, n=n)
                want = [(s[h] + sum(rows[j][h] * party**(j + 1) for j in range(t))) % q for h in range(n)]
                assert got == want, (name, t, party)
                checked += 1
    assert checked > 300


def test_dense_binary_moduli_take_the_bitserial_path(hostcheck):
    """GF(2^n) moduli whose low part r(x) is not short (>= 2^28) do not use the fold reduction;
    n <= 32 uses long division of the 64-bit carry-less product.  All against the bit-level oracle."""
    from mpyc_amd.gfpx import BinaryPolynomial
    mods = [int(BinaryPolynomial.next_irreducible((1 << 64) | (1 << 45))),       # dense, one limb
            int(BinaryPolynomial.next_irreducible((1 << 100) | (1 << 70))),      # dense, two limbs
            int(BinaryPolynomial.next_irreducible((1 << 128) | (1 << 100))),
            int(BinaryPolynomial.next_irreducible(1 << 9)), int(BinaryPolynomial.next_irreducible(1 << 31)),
            int(BinaryPolynomial.next_irreducible(1 << 32)), int(BinaryPolynomial.next_irreducible(1 << 33)),
            int(BinaryPolynomial.next_irreducible((1 << 40) | (1 << 27))),       # r just below 2^28: fold x several passes
            int(BinaryPolynomial.next_irreducible(1 << 65)), int(BinaryPolynomial.next_irreducible(1 << 127))]
    for mod in mods:
        F = po.Field(mod, True)
        ev = edge_values(F) + rand_values(F, 10, 1)
        a, b = cross(ev)
        got, _ = run(hostcheck, F, HC_MUL, a, b)
        assert got == po.vec(po.mul, F, a, b), hex(mod)
        a, b = rand_values(F, 1500, 2), rand_values(F, 1500, 3)
        got, _ = run(hostcheck, F, HC_MUL, a, b)
        assert got == po.vec(po.mul, F, a, b), hex(mod)


def test_small_binary_fields_every_degree_fold_and_long_division(hostcheck):
    """GF(2^n), 9 <= n <= 32 (they ride the one-limb policy): since round 6 sparse moduli -- the first irreducible of every
    degree, what mpyc's find_irreducible returns -- reduce by fold passes, moduli whose low part is long (a pass would not pay)
    keep the long division; both against the bit-level oracle, at extreme and random operands."""
    from mpyc_amd.gfpx import BinaryPolynomial
    for deg in range(9, 33):
        mods = [int(BinaryPolynomial.next_irreducible(1 << deg)),                                   # sparse: fold
                int(BinaryPolynomial.next_irreducible((1 << deg) | (1 << (deg - 1)) | (1 << (deg - 2))))]    # r of degree n-1: long division
        for mod in mods:
            assert mod.bit_length() == deg + 1
            F = po.Field(mod, True)
            ev = edge_values(F) + rand_values(F, 6, deg)
            a, b = cross(ev)
            got, _ = run(hostcheck, F, HC_MUL, a, b)
            assert got == po.vec(po.mul, F, a, b), hex(mod)
            a, b = rand_values(F, 400, 2 * deg), rand_values(F, 400, 3 * deg)
            got, _ = run(hostcheck, F, HC_MUL, a, b)
            assert got == po.vec(po.mul, F, a, b), hex(mod)


def test_pseudo_mersenne_with_the_largest_admissible_c(hostcheck):
    """The golden primes all have a tiny c = 2^k - p.  The fold-based policies are selected for c up to
    2^min((k-1)/2, 31) (k <= 64) resp. 2^31 (k > 64), and that is where their carry chains are tightest:
    for every width family take the prime with the LARGEST admissible c and the one just outside the range
    (which must fall back to the reciprocal / Montgomery policies), and check every arithmetic entry at
    extreme operands against Python integers."""
    from mpyc_amd.finfields import is_prime
    PM = {33: 3, 40: 3, 48: 3, 63: 3, 64: 2, 65: 8, 80: 8, 96: 8, 97: 7, 127: 7, 128: 6}      # expected policy kind
    for k, kind in PM.items():
        cb = min((k - 1) // 2, 31) if k <= 64 else 31
        c = (1 << cb) - 1
        while not is_prime((1 << k) - c):
            c -= 2
        p_in = (1 << k) - c
        c = (1 << cb) + 1
        while not is_prime((1 << k) - c):
            c += 2
        p_out = (1 << k) - c
        for p, inside in ((p_in, True), (p_out, False)):
            F = po.Field(p, False)
            _, pk = run(hostcheck, F, HC_ADD, [0], [0])
            assert (pk == kind) == inside, (k, hex(p), pk)
            ev = edge_values(F) + [p - 1, p - 2, p - c, (1 << (k - 1)) % p, ((1 << k) - 1) % p] + rand_values(F, 6, k)
            a, b = cross(sorted(set(ev)))
            for op, fn in ((HC_ADD, po.add), (HC_SUB, po.sub), (HC_MUL, po.mul)):
                got, _ = run(hostcheck, F, op, a, b)
                assert got == po.vec(fn, F, a, b), (k, hex(p), op)
            got, _ = run(hostcheck, F, HC_MULADD, a, b, a[::-1])
            assert got == [(x * y + z) % p for x, y, z in zip(a, b, a[::-1])], (k, hex(p))
            for x in (1, 3, 2**31 - 1, 2**32 - 1):
                got, _ = run(hostcheck, F, HC_MULADD_SMALL, a, None, b, x=x)
                assert got == [(y * x + cc) % p for y, cc in zip(a, b)], (k, hex(p), x)
            raw = [0, 1, p, p + 1, (1 << k) - 1, (1 << (8 * ((k + 7) // 8 if k > 64 else (4 if k <= 32 else 8)))) - 1]
            eb = elem_bytes(p, False)
            raw = [v for v in raw if v < (1 << (8 * eb))] + [(1 << (8 * eb)) - 1]
            got, _ = run(hostcheck, F, HC_REDUCE, raw)
            assert got == [v % p for v in raw], (k, hex(p))
            # recombination-style dot products of extreme values
            for kk in (3, 9, 255):
                n = 8
                rows = [[p - 1] * n if j % 2 == 0 else rand_values(F, n, j) for j in range(kk)]
                lam = [p - 1 if j % 2 == 0 else (p - c) % p for j in range(kk)]
                got, _ = run(hostcheck, F, HC_DOT, [v for row in rows for v in row], lam=lam, k=kk, n=n)
                assert got == [sum(lam[j] * rows[j][h] for j in range(kk)) % p for h in range(n)], (k, hex(p), kk)
            # share evaluation (forward differences) at the largest operands
            for (t, m) in [(1, 3), (3, 7), (4, 255)]:
                n = 12
                s = [p - 1] * n
                rows = [[p - 1] * n for _ in range(t)]
                got, _ = run(hostcheck, F, HC_SHARE, s, None, [v for row in rows for v in row], x=m, k=t, n=n)
                assert got == [(s[h] + sum(rows[j][h] * m**(j + 1) for j in range(t))) % p for h in range(n)], (k, hex(p), t, m)


def test_matrix_core_operand_digits(hostcheck):
    """limb_digits (fields.hpp): L signed base-256 digits in [-128, 127] that sum to x or x - p, for every x in
    [0, p) -- in particular around 0x7f7f..7f, where the representative switches, and for moduli at the very top
    of the 64-bit and 32-bit ranges, where the balanced residue p/2 would NOT be representable."""
    rng = random.Random(64)
    for L, moduli in ((8, [2**64 - 59, 2**64 - 189, 2**63 - 25, 2**61 - 1, 6616326157076047771, 2**40 - 87, 2**33 - 9]),
                      (4, [2**32 - 5, 2**31 - 1, 65537, 19, 2])):
        T = int.from_bytes(b'\x7f' * L, 'little')
        for p in moduli:
            xs = {0, 1, p - 1, p // 2, p // 2 + 1, (p - 2) % p} | {v % p for v in (T - 1, T, T + 1, T + 2, 255, 256, 2**31, 2**63)}
            xs |= {rng.randrange(p) for _ in range(300)}
            buf = (ctypes.c_int8 * L)()
            for x in xs:
                assert hostcheck.hc_limb_digits(ctypes.c_uint64(x), ctypes.c_uint64(p), L, buf) == 0
                d = list(buf)
                assert all(-128 <= v <= 127 for v in d)
                val = sum(v << (8 * i) for i, v in enumerate(d))
                assert val == (x if x <= T else x - p), (L, hex(p), hex(x), d)
                assert -128 * (256**L - 1) // 255 <= val <= 127 * (256**L - 1) // 255


def test_matrix_core_operand_digits_two_limbs(hostcheck):
    """limb_digits_wide: 12 digits for 65..96-bit primes, 16 for 97..128-bit primes (three-word arithmetic)."""
    rng = random.Random(128)
    for L, moduli in ((16, [2**128 - 173, 2**128 - 159, 2**127 - 1, 258797994007609146293811961253269568351, 2**97 - 141, 2**100 - 15]),
                      (12, [2**96 - 17, 2**80 - 65, 2**65 - 49, 2**89 - 1])):
        T = int.from_bytes(b'\x7f' * L, 'little')
        for p in moduli:
            xs = {0, 1, p - 1, p // 2, p // 2 + 1, p - 2, 2**64 - 1, 2**64, 2**64 + 1} | \
                 {v % p for v in (T - 1, T, T + 1, T + 2, 255, 256, 2**63, 2**127, (T >> 64) << 64, ((T >> 64) << 64) - 1)}
            xs |= {rng.randrange(p) for _ in range(300)}
            buf = (ctypes.c_int8 * L)()
            for x in xs:
                x2 = (ctypes.c_uint64 * 2)(x & (2**64 - 1), x >> 64)
                p2 = (ctypes.c_uint64 * 2)(p & (2**64 - 1), p >> 64)
                assert hostcheck.hc_limb_digits_wide(x2, p2, L, buf) == 0
                d = list(buf)
                val = sum(v << (8 * i) for i, v in enumerate(d))
                assert val == (x if x <= T else x - p), (L, hex(p), hex(x), d)



# ---- three-limb pseudo-Mersenne primes (PM192: SecInt(97..160) default fields) -----------------------------------
def _pm192_primes():
    from mpyc_amd.finfields import find_prime_root, is_prime
    ps = [find_prime_root(l)[0] for l in (129, 136, 160, 191, 192)]          # the primes MPyC picks by default
    c = 2**31 - 1                                                            # the largest admissible fold constant
    while not is_prime(2**150 - c):
        c -= 2
    return ps + [2**150 - c]


def test_pm192_policy_and_arithmetic(hostcheck):
    """Every operation of the three-limb policy against Python integers: edge values crossed, random values,
    all raw 192-bit patterns for the reduction, Horner steps with 32-bit multipliers, dot products at the
    accumulator's declared bound (192 terms of (p-1)^2)."""
    PM192, MONT192 = 13, 14
    rng = random.Random(192)
    from mpyc_amd.finfields import find_prime_root, next_prime
    # three-limb Montgomery policy: the "root of unity" primes of SecInt(l, n=N) (finfields.py:332-343; the 136-bit one
    # is the field of demos/np_lpsolver.py -i5), a prime just above 2^128, one just below 2^192 that is not 2^192 - c
    generic = [find_prime_root(136, n=118)[0], find_prime_root(160, n=5)[0], next_prime(2**128), next_prime(2**191 + 2**100),
               next_prime(2**192 - 2**40)]
    assert all(129 <= q.bit_length() <= 192 for q in generic)
    for p in _pm192_primes() + generic:
        want_kind = MONT192 if p in generic else PM192
        F = po.Field(p, False)
        assert elem_bytes(p, False) == 24
        k = p.bit_length()
        ev = sorted(v for v in {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 2**64 - 1, 2**64, 2**128 - 1, 2**128,
                                2**(k - 1), 2**(k - 1) - 1, 2**127, 2**63, p - 2**64, p - 2**128} if 0 <= v < p)
        a, b = cross(ev)
        a += [rng.randrange(p) for _ in range(300)]
        b += [rng.randrange(p) for _ in range(300)]
        cc = [rng.randrange(p) for _ in range(len(a))]
        got, pk = run(hostcheck, F, HC_ADD, a, b)
        assert pk == want_kind and got == [(x + y) % p for x, y in zip(a, b)], hex(p)
        assert run(hostcheck, F, HC_SUB, a, b)[0] == [(x - y) % p for x, y in zip(a, b)], hex(p)
        assert run(hostcheck, F, HC_MUL, a, b)[0] == [(x * y) % p for x, y in zip(a, b)], hex(p)
        assert run(hostcheck, F, HC_NEG, a)[0] == [(-x) % p for x in a], hex(p)
        assert run(hostcheck, F, HC_MULADD, a, b, cc)[0] == [(x * y + z) % p for x, y, z in zip(a, b, cc)], hex(p)
        raw = [2**192 - 1, 2**192 - 2, p, p + 1, 2 * p % 2**192, 2**191, 2**k % 2**192] + [rng.randrange(2**192) for _ in range(200)]
        assert run(hostcheck, F, HC_REDUCE, raw)[0] == [x % p for x in raw], hex(p)
        for x in (0, 1, 2, 7, 255, 65536, 2**31, 2**32 - 1):
            assert run(hostcheck, F, HC_MULADD_SMALL, a, c=cc, x=x)[0] == [(y * x + z) % p for y, z in zip(a, cc)], (hex(p), x)
        for kk in (1, 2, 7, 64, 192):
            n = 40
            rows = [[p - 1] * 3 + [rng.randrange(p) for _ in range(n - 3)] for _ in range(kk)]
            lam = [p - 1] * kk if kk != 7 else [rng.randrange(p) for _ in range(kk)]
            flat = [v for r in rows for v in r]
            got, _ = run(hostcheck, F, HC_DOT, flat, lam=lam, k=kk, n=n)
            assert got == [sum(lam[j] * rows[j][i] for j in range(kk)) % p for i in range(n)], (hex(p), kk)
    # even moduli of three limbs are refused
    rc = hostcheck.hc_run(0, limbs3(2**140 + 2), 3, HC_ADD, None, None, None, None, ctypes.c_size_t(0), ctypes.c_uint32(0), None, 0, None, None)
    assert rc == 104                                                         # 100 + PB_EMODULUS


def test_wide_primes_device_header_against_reference_vectors(hostcheck, golden_wide):
    """PM192 (fields.hpp, compiled for the host) against the reference's outputs for its 129..192-bit default primes, then
    the same cross-product / random / multiply-add / dot checks as every other policy"""
    test_golden_elementwise(hostcheck, golden_wide)
    test_edges_cross_product(hostcheck, golden_wide)
    test_random_mul(hostcheck, golden_wide)
    test_muladd_small(hostcheck, golden_wide)
    test_dot(hostcheck, golden_wide)
    for name, case in golden_wide.items():                  # share evaluation over the three-limb policies
        F = field_of(case)
        q = F.order
        for t in (1, 2, 3, 4):
            s_ = [q - 1, 0, 1] + rand_values(F, 9, t)
            rows = [[q - 1] + rand_values(F, 11, 20 + j) for j in range(t)]
            for party in (1, 5, 40):
                got, _ = run(hostcheck, F, HC_SHARE, s_, None, [v for r in rows for v in r], x=party, k=t, n=12)
                assert got == [(s_[h] + sum(rows[j][h] * party**(j + 1) for j in range(t))) % q for h in range(12)], (name, t, party)


def test_mersenne_policy_and_lazy_chains(hostcheck):
    """PM64<false,true> (p = 2^k - 1, 33 <= k <= 61; fields.hpp presum / fold64): the product that never forms the 128-bit
    number, and the partially reduced chains of ff_pow / k_inv_batch, for EVERY k the policy accepts (only 2^61 - 1 is
    prime; the arithmetic must hold for the others too, the C ABI takes any modulus) -- and k = 62, 63 go to the general
    2^k - c policy."""
    from types import SimpleNamespace
    PM64_MERS, PM64_GEN = 1, 3
    rng = random.Random(61)
    for k in list(range(33, 64)):
        p = 2**k - 1
        F = SimpleNamespace(modulus=p, binary=False, order=p)
        edge = [0, 1, 2, 3, p - 1, p - 2, 2**32 - 1, 2**32, 2**32 + 1, (p >> 1), (p >> 1) + 1, 2**(k - 1), 2**(k - 1) - 1,
                (2**32 - 1) << (k - 32) & p, p - 2**32, p - 2**32 + 1]
        edge = [e % p for e in edge] + [rng.randrange(p) for _ in range(12)]
        a, b = cross(edge)
        got, pk = run(hostcheck, F, HC_MUL, a, b)
        assert pk == (PM64_MERS if k <= 61 else PM64_GEN), (k, pk)
        assert got == [x * y % p for x, y in zip(a, b)], k
        c = list(reversed(a))
        got, _ = run(hostcheck, F, HC_MULADD, a, b, c)
        assert got == [(x * y + z) % p for x, y, z in zip(a, b, c)], k
        for noncanon in (0, 1):
            got, _ = run(hostcheck, F, HC_LAZY, a, b, x=noncanon)
            want = []
            for u, v in zip(a, b):
                r = u * v % p
                for j in range(6):
                    r = r * r % p
                    r = r * (u if j & 1 else v) % p
                want.append(r)
            assert got == want, (k, noncanon)
        ra = [rng.randrange(p) for _ in range(3000)]
        rb = [rng.randrange(p) for _ in range(3000)]
        got, _ = run(hostcheck, F, HC_MUL, ra, rb)
        assert got == [x * y % p for x, y in zip(ra, rb)], k


def test_k64_lazy_chains(hostcheck):
    """PM64<true,false> (p = 2^64 - c): the partially reduced product chains of ff_pow / the batched inverse (fields.hpp
    red128_lazy: any 64-bit word is a representative, p is subtracted once at the end), entered with canonical operands and
    with the second representative u + p of every u < c, for small, typical and the largest admissible c."""
    from types import SimpleNamespace
    rng = random.Random(64189)
    for c in (3, 59, 189, 2**16 + 1, 2**31 - 1, 2**31 - 19):
        p = 2**64 - c
        F = SimpleNamespace(modulus=p, binary=False, order=p)
        edge = [0, 1, 2, c - 1, c, c + 1, p - 1, p - 2, p - c, 2**32 - 1, 2**32, 2**63, 2**63 - 1, p >> 1, (p >> 1) + 1, p - 2**32]
        edge = [e % p for e in edge] + [rng.randrange(p) for _ in range(10)] + [rng.randrange(c) for _ in range(4)]
        a, b = cross(edge)
        for noncanon in (0, 1):
            got, pk = run(hostcheck, F, HC_LAZY, a, b, x=noncanon)
            want = []
            for u, v in zip(a, b):
                r = u * v % p
                for j in range(6):
                    r = r * r % p
                    r = r * (u if j & 1 else v) % p
                want.append(r)
            assert got == want, (c, noncanon)


def test_column_accumulators_of_the_skinny_products(hostcheck):
    """fields.hpp ColAcc (k_vecmat_partial_col): the shared operand in three limbs of 22 / 22 / 20 bits, six column sums of
    partial products, rebuilt into the 192-bit sum that acc_reduce takes -- equal to the plain dot product modulo p for every
    one-word prime policy (Mersenne, 2^64 - c, 2^k - c, reciprocal), below and beyond the 192 terms of one flush (then the
    residue re-enters the column sums as the term 1 x residue, as in the kernels), with all operands p - 1, with limb patterns
    of all ones, and random."""
    from types import SimpleNamespace
    rng = random.Random(2264)
    for p in (2**61 - 1, 2**64 - 189, 2**64 - 59, 2**40 - 87, 2**63 - 25, 2**33 - 9, 6616326157076047771, 18446744073709551557,
              (1 << 62) + 135, 4294967311):
        F = SimpleNamespace(modulus=p, binary=False, order=p)
        for k in (1, 2, 5, 64, 191, 192, 193, 383, 1000):      # from 192 terms on: the kernels' flush, the residue re-enters as a term
            n = 6
            special = [p - 1, (2**22 - 1) % p, ((2**22 - 1) << 22) % p, (2**64 - 2**44) % p, (2**32 - 1) % p, (2**64 - 2**32) % p]
            lam = [p - 1 if k >= 192 or j % 3 == 0 else special[j % 6] if j % 3 == 1 else rng.randrange(p) for j in range(k)]
            rows = [[p - 1] * n if k >= 192 or j % 2 == 0 else [special[(j + h) % 6] if h % 2 else rng.randrange(p) for h in range(n)]
                    for j in range(k)]
            got, _ = run(hostcheck, F, HC_COLDOT, [v for row in rows for v in row], lam=lam, k=k, n=n)
            want = [sum(lam[j] * rows[j][h] for j in range(k)) % p for h in range(n)]
            assert got == want, (hex(p), k)


def test_pm64_general_policy_all_widths(hostcheck):
    """PM64<false,false> (p = 2^k - c, 33 <= k <= 63, 1 < c < 2^min((k-1)/2, 31)): the two folds are written on the words
    (fields.hpp fold128: `(hi << (64 - k)) | (lo >> k)`), so every k must be exercised -- edge values, values around the fold
    boundaries and random ones, for small, large and borderline c; product, fused multiply-add, the 32-bit Horner step, the
    unreduced accumulation."""
    from types import SimpleNamespace
    PM64_GEN = 3
    rng = random.Random(3363)
    for k in range(33, 64):
        cb = min((k - 1) // 2, 31)
        for c in sorted({3, 5, 87, (1 << cb) - 1, (1 << cb) - 3, rng.randrange(2, 1 << cb) | 1}):
            p = (1 << k) - c
            F = SimpleNamespace(modulus=p, binary=False, order=p)
            edge = [0, 1, 2, c, c + 1, p - 1, p - 2, p - c, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, p >> 1, (p >> 1) + 1,
                    (1 << (k - 1)) - 1, 1 << (k - 1), p - (1 << 32), ((1 << 32) - 1) << (k - 32) & ((1 << k) - 1)]
            edge = [e % p for e in edge] + [rng.randrange(p) for _ in range(8)]
            a, b = cross(edge)
            got, pk = run(hostcheck, F, HC_MUL, a, b)
            assert pk == PM64_GEN, (k, c, pk)
            assert got == [x * y % p for x, y in zip(a, b)], (k, c)
            z = list(reversed(a))
            got, _ = run(hostcheck, F, HC_MULADD, a, b, z)
            assert got == [(x * y + w) % p for x, y, w in zip(a, b, z)], (k, c)
            for x in (3, 255, 2**31 - 1, 2**32 - 1):
                got, _ = run(hostcheck, F, HC_MULADD_SMALL, a, None, z, x=x)
                assert got == [(u * (x % p) + w) % p for u, w in zip(a, z)], (k, c, x)
            rows = [[rng.randrange(p) for _ in range(64)] for _ in range(7)]
            lam = [rng.randrange(p) for _ in range(7)]
            got, _ = run(hostcheck, F, HC_DOT, [v for row in rows for v in row], lam=lam, k=7, n=64)
            assert got == [sum(l * rows[j][i] for j, l in enumerate(lam)) % p for i in range(64)], (k, c)


def test_bitsliced_gf2_64_product(hostcheck):
    """mpyc_amd/csrc/bitslice.hpp (the arithmetic of k_gf2w64_mul_bitsliced) compiled for the host: the 32 x 32 bit
    transpose against its definition, and the product of the 16 elements of a lane -- one transpose per operand (operand halves
    in register halves), packed Karatsuba down to 8 x 8 leaves, fold modulo x^64 + x^4 + x^3 + x + 1, transpose out -- against
    the oracle's GF(2^64) product
    (gfpx.py:988-1045 restated) for random and extreme operands."""
    import random
    from oracle import pyoracle as po
    rng = random.Random(64)
    for _ in range(5):
        words = [rng.getrandbits(32) for _ in range(32)]
        out = (ctypes.c_uint32 * 32)()
        assert hostcheck.hc_bs64_transpose32((ctypes.c_uint32 * 32)(*words), out) == 0
        assert list(out) == [sum(((words[e] >> i) & 1) << e for e in range(32)) for i in range(32)]
    F = po.Field((1 << 64) | 0x1b, True)
    full = (1 << 64) - 1
    extreme = [0, 1, 2, 3, full, full - 1, 1 << 63, (1 << 63) | 1, 0x1b, 1 << 32, (1 << 32) - 1, 0x8000000080000000, 0x5555555555555555,
               0xAAAAAAAAAAAAAAAA, 7 << 61, 0xFFFFFFFF00000000]
    for rnd in range(20):
        a = [rng.getrandbits(64) for _ in range(32)]
        b = [rng.getrandbits(64) for _ in range(32)]
        if rnd < 4:
            a[:16] = extreme
            b[:16] = extreme[::-1] if rnd % 2 else extreme
            b[16:32] = extreme[rnd:] + extreme[:rnd]
        for off in (0, 16):
            o16 = (ctypes.c_uint64 * 16)()
            assert hostcheck.hc_bs64_mul16_packed((ctypes.c_uint64 * 16)(*a[off:off + 16]), (ctypes.c_uint64 * 16)(*b[off:off + 16]), o16) == 0
            assert list(o16) == [po.mul(F, x, y) for x, y in zip(a[off:off + 16], b[off:off + 16])], (rnd, off)


def test_dot_products_in_28_bit_digits(hostcheck):
    """Round 6: the recombination kernels of the multi-limb 2^k - c primes accumulate sum_j lam_j * x_j in 28-bit digits
    (fields.hpp LazyDot: column sums of digit products, one carry pass at the end).  Every bit length from 65 to 192 that
    has such a policy, up to the declared bound of 32 terms, operands at the extremes (p - 1 everywhere: the largest
    columns; single bits around every digit and word boundary) and random."""
    from mpyc_amd.finfields import find_prime_root
    rng = random.Random(28)
    seen = set()
    for bits in list(range(65, 193, 3)) + [80, 96, 97, 112, 113, 128, 129, 136, 160, 191, 192]:
        p = find_prime_root(bits)[0]
        if p in seen:
            continue
        seen.add(p)
        F = po.Field(p, False)
        k = p.bit_length()
        marks = sorted({v % p for e in (27, 28, 29, 31, 32, 33, 55, 56, 57, 63, 64, 65, 83, 84, 85, 95, 96, 111, 112, 113, 127, 128, 139,
                                        140, 141, 167, 168, 169, k - 1) if e < k for v in (1 << e, (1 << e) - 1)})
        for kk in (1, 2, 3, 4, 7, 9, 32):
            n = 12 + len(marks)
            rows = [([p - 1] * 6 + [rng.randrange(p) for _ in range(6)] + marks) if j % 2 == 0 else
                    ([p - 1] * 3 + [rng.randrange(p) for _ in range(9)] + marks[::-1]) for j in range(kk)]
            for lam in ([p - 1] * kk, [rng.randrange(p) for _ in range(kk)], [marks[(3 * j) % len(marks)] for j in range(kk)]):
                flat = [v for r in rows for v in r]
                rc = run(hostcheck, F, HC_LDOT, flat, lam=lam, k=kk, n=n, allow_rc=(2,))
                assert rc is not None, (bits, 'no digit accumulator for a multi-limb 2^k - c prime')
                got, _ = rc
                assert got == [sum(lam[j] * rows[j][i] for j in range(kk)) % p for i in range(n)], (hex(p), kk)


def test_product_chains_in_digits(hostcheck):
    """Round 6: sqrt / inverse sqrt / pow over the two-limb 2^k - c primes run their product chain in NL digits of
    ceil(k / NL) bits (fields.hpp DigitChain), partially reduced until the end.  Every bit length 65..192, every digit
    count that fits, chains of 13 products from extreme and random operands, against Python integers."""
    from mpyc_amd.finfields import find_prime_root
    rng = random.Random(2806)
    seen = set()
    for bits in range(65, 193):
        p = find_prime_root(bits)[0]
        if p in seen:
            continue
        seen.add(p)
        F = po.Field(p, False)
        k = p.bit_length()
        c = (1 << k) - p
        ev = sorted({v % p for v in (0, 1, 2, p - 1, p - 2, (p - 1) // 2, 2**64 - 1, 2**64, 2**(k - 1), 2**(k - 1) - 1, c, c + 1, p - c,
                                      2**27, 2**28 - 1, 2**54, 2**56 - 1, 2**81, 2**84 - 1, 2**128, 2**140 - 1, 2**168)})
        a, b = cross(ev)
        a += [rng.randrange(p) for _ in range(200)]
        b += [rng.randrange(p) for _ in range(200)]
        def chain(u, v):
            t = u * v % p
            for j in range(6):
                t = t * t % p
                t = t * (u if j & 1 else v) % p
            return t
        want = [chain(u, v) for u, v in zip(a, b)]
        ran = 0
        for nl in (3, 4, 5, 6, 7):
            w = -(-k // nl)
            rc = run(hostcheck, F, HC_CHAIN, a, b, x=nl, allow_rc=(3,))
            if rc is None:
                lo_nl, hi_nl = (3, 4) if k <= 96 else (3, 5) if k <= 128 else (5, 7)
                assert not (lo_nl <= nl <= hi_nl and 22 <= w <= 28 and (c << (w * nl - k)) < 2**20), (bits, nl)
                continue
            assert rc[0] == want, (hex(p), nl)
            ran += 1
        assert ran >= 1, (bits, 'no digit chain for a default prime')
