"""Pin the oracle (oracle/pyoracle.py, oracle/fforacle.c) to the REAL reference:
tests/golden/*.json were produced by tests/golden/make_golden.py running lschoe/mpyc."""
import pytest
import json
import os

import numpy as np

from oracle import pyoracle as po
from oracle.coracle import elem_bytes
from fieldutil import field_of, pack, unhex, unpack, lshape

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_py_elementwise(golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        a, b = unhex(case['a']), unhex(case['b'])
        assert po.vec(po.add, F, a, b) == unhex(case['add']), name
        assert po.vec(po.sub, F, a, b) == unhex(case['sub']), name
        assert po.vec(po.mul, F, a, b) == unhex(case['mul']), name
        assert [po.neg(F, x) for x in a] == unhex(case['neg']), name
        sc = int(case['scalar'], 16)
        assert [po.add(F, x, sc) for x in a] == unhex(case['add_scalar']), name
        assert [po.mul(F, x, sc) for x in a] == unhex(case['mul_scalar']), name
        assert [po.sub(F, sc, x) for x in a] == unhex(case['rsub_scalar']), name
        assert [po.reduce(F, x) for x in unhex(case['raw'])] == unhex(case['raw_reduced']), name
        if 'neg_in' in case:
            assert [po.reduce(F, x) for x in case['neg_in']] == unhex(case['neg_in_reduced']), name


def test_py_sharing(golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        a = unhex(case['a'])
        s = a[:13] + a[-5:]
        n = len(s)
        for sc in case['sharing']:
            t, m, draws = sc['t'], sc['m'], unhex(sc['draws'])
            np_sh = po.np_random_split(F, s, t, m, draws)
            assert np_sh == [unhex(r) for r in sc['np_shares']], (name, t, m)
            li_sh = po.random_split(F, s, t, m, draws)
            assert li_sh == [unhex(r) for r in sc['list_shares']], (name, t, m)
            # list path == np path fed with permuted draws
            assert po.np_random_split(F, s, t, m, po.list_to_np_draws(draws, t, n)) == li_sh
            for rec in sc['recombine']:
                xs = rec['xs']
                assert po.recombination_vector(F, xs, 0) == unhex(rec['vector']), (name, xs)
                pts = [(x, np_sh[x - 1]) for x in xs]
                assert po.np_recombine(F, pts) == unhex(rec['np_out']), (name, xs)
                assert po.recombine_unreduced(F, pts) == unhex(rec['list_out_unreduced']), (name, xs)
            mu = sc['multi']
            pts = [(x, np_sh[x - 1]) for x in mu['xs']]
            assert [po.recombination_vector(F, mu['xs'], xr) for xr in mu['x_rs']] == \
                [unhex(v) for v in mu['vectors']]
            assert po.np_recombine(F, pts, mu['x_rs']) == [unhex(r) for r in mu['out']], name


def test_py_known_answers(golden_sbox):
    """Reference KATs: tests/test_finfields.py:29-30,94-99; FIPS-197."""
    F = po.Field(0x11b, binary=True)
    assert po.mul(F, 16, 16) == 27 and po.mul(F, 32, 16) == 54 and po.mul(F, 57, 67) == 137
    assert po.mul(F, 137, po.inv(F, 57)) == 67 and po.mul(F, 3, 3) == 5 and po.mul(F, 48, 16) == 45
    kat = golden_sbox['kat']
    assert kat == {'16*16': 27, '32*16': 54, '57*67': 137, '137/57': 67, '3*3': 5, '48*16': 45}
    P61 = po.Field(2**61 - 1)
    p = P61.modulus
    assert po.recombination_vector(P61, (1, 2), 0) == [2, p - 1]
    assert po.recombination_vector(P61, (1, 2, 3), 0) == [3, p - 3, 1]
    assert po.recombination_vector(P61, (2, 3, 1), 0) == [p - 3, 1, 3]
    assert po.recombination_vector(F, (1, 2, 3), 0) == [1, 1, 1]
    # GF(2^8), draws 3,200,17,99, secrets 57,67,0,255, t=1, m=3
    assert po.np_random_split(F, [57, 67, 0, 255], 1, 3, [3, 200, 17, 99]) == \
        [[58, 139, 17, 156], [63, 200, 34, 57], [60, 0, 51, 90]]
    with open(os.path.join(GOLDEN, 'convention.json')) as fh:
        conv = json.load(fh)
    draws = list(range(1000, 1006))
    assert po.random_split(P61, [5, 7, 11], 2, 3, draws) == conv['list'] and conv['list'][0][0] == 2006
    assert po.np_random_split(P61, [5, 7, 11], 2, 3, draws) == conv['np'] and conv['np'][0][0] == 2008
    # S-box
    rows8, b = po.aes_affine_rows()
    assert rows8 == golden_sbox['rows8'] and b == golden_sbox['b'] == 0x63
    assert po.sbox(range(256)) == golden_sbox['table']
    assert golden_sbox['table'][:4] == [0x63, 0x7c, 0x77, 0x7b]
    assert [po.pow254(F, v) for v in range(256)] == golden_sbox['pow254']
    for v in range(1, 256):
        assert po.mul(F, po.pow254(F, v), v) == 1          # tests/test_runtime.py:818-843 a**254*a == 1


def test_c_oracle(coracle, golden_fields, golden_sbox):
    for name, case in golden_fields.items():
        F = field_of(case)
        eb = elem_bytes(F.modulus, F.binary)
        cf = coracle.CField(F.modulus, F.binary)
        a, b = unhex(case['a']), unhex(case['b'])
        A, B = pack(a, eb), pack(b, eb)
        assert unpack(cf.ew(coracle.ADD, A, B), eb) == unhex(case['add']), name
        assert unpack(cf.ew(coracle.SUB, A, B), eb) == unhex(case['sub']), name
        assert unpack(cf.ew(coracle.MUL, A, B), eb) == unhex(case['mul']), name
        assert unpack(cf.ew(coracle.NEG, A), eb) == unhex(case['neg']), name
        if case['raw_width'] == 8 * eb:
            assert unpack(cf.ew(coracle.REDUCE, pack(unhex(case['raw']), eb)), eb) == unhex(case['raw_reduced'])
        s = a[:13] + a[-5:]
        n = len(s)
        for sc in case['sharing']:
            t, m, draws = sc['t'], sc['m'], unhex(sc['draws'])
            S = pack(s, eb)
            C = pack(draws, eb).reshape(lshape(eb, t, n))
            sh = cf.split(S, C, t, m)
            got = [unpack(sh[i], eb) for i in range(m)]
            assert got == [unhex(r) for r in sc['np_shares']], (name, t, m)
            for rec in sc['recombine']:
                xs = rec['xs']
                rows = [sh[x - 1] for x in xs]
                out = cf.recombine(rows, unhex(rec['vector']))
                assert unpack(out, eb) == unhex(rec['np_out']), (name, xs)
            mu = sc['multi']
            rows = [sh[x - 1] for x in mu['xs']]
            lam = [v for vv in mu['vectors'] for v in unhex(vv)]
            out = cf.recombine(rows, lam, w=len(mu['x_rs']))
            assert [unpack(out[r], eb) for r in range(len(mu['x_rs']))] == [unhex(r) for r in mu['out']]
    x = np.arange(256, dtype=np.uint8)
    assert list(coracle.sbox(x, golden_sbox['rows8'], golden_sbox['b'])) == golden_sbox['table']


def test_c_oracle_threads(coracle):
    """OpenMP legs give the same answer as one thread (used for the N-core CPU baseline)."""
    import random
    r = random.Random(3)
    p = 2**61 - 1
    a = np.array([r.randrange(p) for _ in range(5000)], dtype=np.uint64)
    b = np.array([r.randrange(p) for _ in range(5000)], dtype=np.uint64)
    cf = coracle.CField(p)
    one = cf.ew(coracle.MUL, a, b)
    coracle.set_threads(max(2, coracle.max_threads()))
    many = cf.ew(coracle.MUL, a, b)
    coracle.set_threads(1)
    assert (one == many).all()
    assert [int(v) for v in one[:50]] == [int(x) * int(y) % p for x, y in zip(a[:50], b[:50])]


def test_py_linalg():
    """np.linalg.det / inv / solve of the reference (tests/golden/linalg.json) vs the restatement."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'linalg.json')))
    for name, fc in g.items():
        F = po.Field(int(fc['modulus'], 16), fc['binary'])
        ux = lambda m: [[int(v, 16) for v in row] for row in m]
        red = lambda v: int(v, 16) if fc['binary'] else int(v, 16) % F.modulus
        for c in fc['cases']:
            A, B = ux(c['A']), ux(c['B'])
            assert po.gauss_det(F, A) == red(c['det']), (name, c['kind'])
            if 'error' in c:
                with pytest.raises(ZeroDivisionError, match='no inverse exists'):
                    po.gauss_solve(F, A, B)
                continue
            eye = [[int(i == j) for j in range(c['n'])] for i in range(c['n'])]
            assert po.gauss_solve(F, A, eye) == [[red(v) for v in row] for row in c['inv']], (name, c['kind'])
            assert po.gauss_solve(F, A, B) == [[red(v) for v in row] for row in c['solve']], (name, c['kind'])
        dets = [red(v) for row in fc['stack_det'] for v in row]
        assert [po.gauss_det(F, ux(m)) for m in fc['stack']] == dets


def test_py_sqrt():
    g = json.load(open(os.path.join(GOLDEN, 'sqrt.json')))
    for name, c in g.items():
        F = po.Field(int(c['modulus'], 16), False)
        assert [po.sqrt_prime(F, int(v, 16)) for v in c['a']] == [int(v, 16) for v in c['sqrt']], name
        assert [po.inv(F, po.sqrt_prime(F, int(v, 16))) for v in c['sq']] == [int(v, 16) for v in c['inv_sqrt']]


def test_py_aes128_fips197():
    """FIPS-197 appendix C.1 (the vector the reference demo reproduces, docs/demos.rst:611) and appendix B."""
    key = list(range(16))
    pt = [17 * i for i in range(16)]
    assert bytes(po.aes128_encrypt(key, pt)).hex() == '69c4e0d86a7b0430d8cdb78070b4c55a'
    key = list(bytes.fromhex('2b7e151628aed2a6abf7158809cf4f3c'))
    pt = list(bytes.fromhex('3243f6a8885a308d313198a2e0370734'))
    assert bytes(po.aes128_encrypt(key, pt)).hex() == '3925841d02dc09fbdc118597196a0b32'


def test_py_wide_primes(golden_wide):
    """the three-limb primes (129..192 bits; reference outputs in wide.json): the Python-integer oracle is the checker
    for them on the GPU (tests/test_gpu_pm192.py) -- the C oracle stops at 128 bits"""
    assert sorted(golden_wide) == ['P129', 'P129G', 'P136', 'P136R', 'P160', 'P192']
    test_py_elementwise(golden_wide)
    test_py_sharing(golden_wide)


def test_two_limb_mulmod_long_division_vs_python_and_shift_add(coracle):
    """oracle/fforacle.c computes two-limb products as a schoolbook 256-bit product followed by Knuth long division
    (fast enough for 10^7-element parity runs over 128-bit fields); cross-checked here against Python integers AND
    against the bit-serial shift-and-add product of rounds 1-2, on extreme and random operands."""
    import ctypes
    import random
    L = coracle.lib()
    rnd = random.Random(20260925)

    def lim(x):
        return (ctypes.c_uint64 * 2)(x & (2**64 - 1), x >> 64)
    primes = [2**128 - 173, 2**127 - 1, 258797994007609146293811961253269568351, 2**96 - 17, 2**80 - 65, 2**65 + 131,
              2**97 - 141, (1 << 64) + 13]
    for p in primes:
        edge = [v % p for v in (0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 2**64 - 1, 2**64, 2**64 + 1, p >> 1,
                                (1 << (p.bit_length() - 1)) - 1, (1 << (p.bit_length() - 1)), 2**32, 2**96)]
        pairs = [(a, b) for a in edge for b in edge] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(1500)]
        for a, b in pairs:
            f, s = (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 2)()
            L.orc_mulmod_pair(lim(a), lim(b), lim(p), f, s)
            want = a * b % p
            assert f[0] | (f[1] << 64) == want and s[0] | (s[1] << 64) == want, (hex(p), hex(a), hex(b))
