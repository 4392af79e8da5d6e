#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (lschoe/mpyc, pure Python).

Run in the build container only (the reference does not travel to the GPU box):

    cd /tmp && PYTHONPATH=/root/reference python3 /root/repo/tests/golden/make_golden.py

Everything written here is an input/output pair of a reference function on the
hot path:
    finfields.FiniteFieldArray + - * neg, ctor reduction     finfields.py:717-725,1056-1124,1189
    thresha.np_random_split / random_split                   thresha.py:47-64 / 23-44
    thresha._recombination_vector                            thresha.py:67-85
    thresha.np_recombine / recombine                         thresha.py:119-132 / 88-116
    GF(2^8) S-box on public values                           demos/np_aes.py:37-43
Share generation is made reproducible by replacing secrets.randbelow (looked up at
call time, thresha.py:37,58) with a replay of a recorded draw list.

Integers are stored as hex strings; the files are small (n ~ 40 per case).
"""
import json
import os
import random
import secrets
import sys

import numpy as np

from mpyc import finfields, gfpx, thresha

OUT = os.environ.get('GOLDEN_OUT') or os.path.dirname(os.path.abspath(__file__))     # GOLDEN_OUT: regenerate elsewhere (tests)
rng = random.Random(20260925)

P61 = 2**61 - 1
P64 = 2**64 - 189
P128 = 2**128 - 173
P127 = 2**127 - 1
P96 = 2**96 - 17
P80 = int(finfields.find_prime_root(80)[0])          # default SecFxp-size prime
P63G = 0x5BD1E995C6A4A793                            # generic 63-bit, checked prime below
P128G = None                                         # generic 128-bit prime, found below
P31 = 2**31 - 1
P32G = 4294967291                                    # 2^32 - 5
P40 = int(finfields.find_prime_root(40)[0])


def next_prime_from(x):
    from mpyc import gmpy as g
    x |= 1
    while not g.is_prime(x):
        x += 2
    return int(x)


P63G = next_prime_from(P63G)
P128G = next_prime_from(0xC2B2AE3D27D4EB4F165667B19E3779F9)   # "random" odd 128-bit start
P100G = next_prime_from(0x9E3779B97F4A7C15F39CC0605)          # generic ~100-bit

PRIMES = {
    'P61': P61, 'P64': P64, 'P128': P128, 'P127': P127, 'P96': P96, 'P80': P80,
    'P63G': P63G, 'P128G': P128G, 'P100G': P100G, 'P31': P31, 'P32G': P32G, 'P40': P40,
    'GF19': 19, 'GF101': 101, 'GF2': 2, 'GF3': 3, 'GF65537': 65537,
}

GF2X = gfpx.GFpX(2)
BINARIES = {
    'GF2_8': int(finfields.find_irreducible(2, 8)),       # 0x11b, AES
    'GF2_128': int(finfields.find_irreducible(2, 128)),   # x^128+x^7+x^2+x+1
    'GF2_4': int(finfields.find_irreducible(2, 4)),
    'GF2_2': int(finfields.find_irreducible(2, 2)),
    'GF2_1': int(finfields.find_irreducible(2, 1)),
    'GF2_16': int(finfields.find_irreducible(2, 16)),
    'GF2_64': int(finfields.find_irreducible(2, 64)),
    'GF2_100': int(finfields.find_irreducible(2, 100)),
}


def hx(v):
    return hex(int(v))


def hxl(a):
    return [hx(v) for v in a]


def edge_values(q, bits):
    e = [0, 1, 2, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, 2**32 - 1, 2**32, 2**63, 2**64 - 1,
         2**bits - 1, 2**(bits - 1)]
    return [v % q for v in e]


class Replay:
    def __init__(self, order, count):
        self.draws = [rng.randrange(order) for _ in range(count)]
        # sprinkle extremes
        for i, v in zip(range(0, count, 7), (0, order - 1, 1, order - 2)):
            self.draws[i] = v % order
        self.i = 0

    def __call__(self, bound):
        v = self.draws[self.i]
        self.i += 1
        assert v < bound
        return v


def field_case(name, F, is_binary, raw_width=None):
    q = F.order
    bits = (q - 1).bit_length() if q > 2 else 1
    ev = edge_values(q, max(bits, 1))
    a = ev + [rng.randrange(q) for _ in range(27)]
    b = list(reversed(ev)) + [rng.randrange(q) for _ in range(27)]
    n = len(a)
    A, B = F.array(a), F.array(b)

    def ints(arr):
        return [int(v) for v in arr.value.reshape(-1)]

    case = {'name': name, 'binary': is_binary, 'modulus': hx(int(F.modulus)), 'order': hx(q),
            'a': hxl(a), 'b': hxl(b)}
    case['add'] = hxl(ints(A + B))
    case['sub'] = hxl(ints(A - B))
    case['mul'] = hxl(ints(A * B))
    case['neg'] = hxl(ints(-A))
    sc = a[5] if a[5] else 3 % q
    case['scalar'] = hx(sc)
    case['add_scalar'] = hxl(ints(A + F(sc)))
    case['mul_scalar'] = hxl(ints(A * F(sc)))
    case['rsub_scalar'] = hxl(ints(F(sc) - A))
    # raw (unreduced) constructor inputs: finfields.py:724 `value %= modulus`
    if not is_binary:
        width = 8 * ((bits + 7) // 8)
        width = 32 if width <= 32 else 64 if width <= 64 else 128
        width = raw_width or width
        raw = [2**width - 1, 2**width - 2, q, q + 1, 2 * q % 2**width] + [rng.randrange(2**width) for _ in range(11)]
        case['raw_width'] = width
        case['raw'] = hxl(raw)
        case['raw_reduced'] = hxl(ints(F.array(raw)))
        case['neg_in'] = [-1, -2, -(q // 2), -q, -(q + 1)]
        case['neg_in_reduced'] = hxl(ints(F.array(case['neg_in'])))
    else:
        width = 8 if bits <= 8 else 64 if bits <= 64 else 128
        raw = [2**width - 1, 2**width - 2, q % 2**width, 1 << (width - 1)] + [rng.randrange(2**width) for _ in range(12)]
        case['raw_width'] = width
        case['raw'] = hxl(raw)
        case['raw_reduced'] = hxl(ints(F.array(raw)))

    # ---- sharing -------------------------------------------------------
    shares_cases = []
    s = a[: 13] + a[-5:]
    n = len(s)
    tms = [(0, 1), (1, 3), (3, 7), (1, 2), (4, 9)]
    if name == 'P64':
        tms += [(2, 5), (5, 11), (1, 4)]
    for (t, m) in tms:
        if m >= q:       # thresha needs m < order (sectypes.py:639-647)
            continue
        rp = Replay(q, t * n)
        old = secrets.randbelow
        secrets.randbelow = rp
        try:
            sh_np = thresha.np_random_split(F, F.array(s), t, m)
        finally:
            secrets.randbelow = old
        rp2 = Replay.__new__(Replay)
        rp2.draws, rp2.i = rp.draws, 0
        secrets.randbelow = rp2
        try:
            sh_list = thresha.random_split(F, list(s), t, m)
        finally:
            secrets.randbelow = old
        sc_ = {'t': t, 'm': m, 'draws': hxl(rp.draws),
               'np_shares': [hxl(int(v) for v in row) for row in sh_np],
               'list_shares': [hxl(int(v) for v in row) for row in sh_list]}
        # recombination from several point sets (orders matter: thresha.py:67-85)
        recs = []
        xsets = [tuple(range(1, t + 2)), tuple(range(m - t, m + 1)), tuple((j % m) + 1 for j in range(1, t + 2))]
        if 2 * t + 1 <= m:
            xsets.append(tuple(range(1, 2 * t + 2)))
            xsets.append(tuple(((1 + j) % m) + 1 for j in range(2 * t + 1)))
        for xs in xsets:
            if len(set(xs)) != len(xs):
                continue
            vec = thresha._recombination_vector(F, xs, 0)
            pts = [(x, sh_np[x - 1]) for x in xs]
            y = thresha.np_recombine(F, pts)
            assert [int(v) for v in y.value] == [int(F(v).value) for v in s], (name, t, m, xs)
            pts_l = [(x, [int(v) for v in sh_np[x - 1]]) for x in xs]
            y_l = thresha.recombine(F, pts_l)       # unreduced for raw ints (thresha.py:109)
            recs.append({'xs': list(xs), 'vector': hxl(int(v) for v in vec),
                         'np_out': hxl(int(v) for v in y.value),
                         'list_out_unreduced': hxl(int(v) for v in y_l)})
        # multi-point recombination (x_rs list, thresha.py:96-97,125-131)
        xs = tuple(range(1, t + 2))
        x_rs = [0, m + 1 if m + 1 < q else 0, 1]
        pts = [(x, sh_np[x - 1]) for x in xs]
        yw = thresha.np_recombine(F, pts, x_rs)
        vecs = [thresha._recombination_vector(F, xs, xr) for xr in x_rs]
        sc_['multi'] = {'xs': list(xs), 'x_rs': x_rs, 'vectors': [hxl(int(v) for v in vv) for vv in vecs],
                        'out': [hxl(int(v) for v in row) for row in yw.value]}
        sc_['recombine'] = recs
        shares_cases.append(sc_)
    case['sharing'] = shares_cases
    return case


def sbox_case():
    f256 = finfields.GF(GF2X(BINARIES['GF2_8']))
    # demos/np_aes.py:23-33: A = circulant([1,0,0,0,1,1,1,1]), B = [1,1,0,0,0,1,1,0]
    r = [1, 0, 0, 0, 1, 1, 1, 1]
    A = [[r[(c - j) % 8] for c in range(8)] for j in range(8)]   # np.roll(r, j): row j
    B = [1, 1, 0, 0, 0, 1, 1, 0]
    table = []
    x = f256.array(list(range(256)))
    inv = x**254
    for v in inv.value:
        bits = [(int(v) >> i) & 1 for i in range(8)]
        y = [(sum(A[rr][c] & bits[c] for c in range(8)) + B[rr]) & 1 for rr in range(8)]
        table.append(sum(y[i] << i for i in range(8)))
    assert table[:4] == [0x63, 0x7c, 0x77, 0x7b], table[:4]     # FIPS-197
    rows8 = [sum(A[rr][c] << c for c in range(8)) for rr in range(8)]
    bbyte = sum(B[i] << i for i in range(8))
    kat = {'16*16': int((f256(16) * f256(16)).value), '32*16': int((f256(32) * f256(16)).value),
           '57*67': int((f256(57) * f256(67)).value), '137/57': int((f256(137) / f256(57)).value),
           '3*3': int((f256(3) * f256(3)).value), '48*16': int((f256(48) * f256(16)).value)}
    return {'modulus': hx(BINARIES['GF2_8']), 'rows8': rows8, 'b': bbyte, 'table': table,
            'pow254': [int(v) for v in inv.value], 'kat': kat}


def main():
    cases = {}
    for name, p in PRIMES.items():
        F = finfields.GF(p)
        cases[name] = field_case(name, F, False)
    for name, mod in BINARIES.items():
        F = finfields.GF(GF2X(mod))
        cases[name] = field_case(name, F, True)
    with open(os.path.join(OUT, 'fields.json'), 'w') as fh:
        json.dump(cases, fh, separators=(',', ':'))
    with open(os.path.join(OUT, 'sbox.json'), 'w') as fh:
        json.dump(sbox_case(), fh, separators=(',', ':'))
    # Appendix A.1 spot values recorded in SURVEY.md (list vs np coefficient convention)
    F = finfields.GF(P61)
    old = secrets.randbelow
    it = iter(range(1000, 2000))
    secrets.randbelow = lambda b: next(it)
    l_ = thresha.random_split(F, [5, 7, 11], 2, 3)
    it = iter(range(1000, 2000))
    n_ = thresha.np_random_split(F, F.array([5, 7, 11]), 2, 3)
    secrets.randbelow = old
    with open(os.path.join(OUT, 'convention.json'), 'w') as fh:
        json.dump({'list': [[int(v) for v in r] for r in l_], 'np': [[int(v) for v in r] for r in n_]}, fh)
    print('wrote', len(cases), 'field cases; sizes:',
          {f: os.path.getsize(os.path.join(OUT, f)) for f in ('fields.json', 'sbox.json', 'convention.json')})



def prss_cases(fields=None, fname='prss.json'):
    """PRSS golden vectors (thresha.py:135-266): PRF outputs and every party's shares."""
    from itertools import combinations
    out = {}
    key0 = int('0x00112233445566778899aabbccddeeff', 16).to_bytes(16, byteorder='little')   # tests/test_thresha.py:43
    uci = 'test uci'.encode()
    n = 9
    for name, F in fields or (('P61', finfields.GF(P61)), ('P64', finfields.GF(P64)), ('P128', finfields.GF(P128)),
                              ('P80', finfields.GF(P80)), ('GF19', finfields.GF(19)), ('P63G', finfields.GF(P63G)),
                              ('P128G', finfields.GF(P128G)), ('GF2_8', finfields.GF(GF2X(BINARIES['GF2_8']))),
                              ('GF2_128', finfields.GF(GF2X(BINARIES['GF2_128'])))):
        case = {'modulus': hx(int(F.modulus)), 'binary': not isinstance(F.modulus, int), 'uci': uci.hex(), 'n': n,
                'settings': []}
        for (m, t) in ((1, 0), (3, 1), (5, 2), (4, 1)):
            if m >= F.order:
                continue
            for bound in (F.order, 1 << max(1, (F.order - 1).bit_length() - 3)):
                subsets = list(combinations(range(m), m - t))
                keys = {S: bytes((b + 17 * k) & 0xff for b in key0) for k, S in enumerate(subsets)}
                setting = {'m': m, 't': t, 'bound': hx(bound), 'keys': {','.join(map(str, S)): k.hex() for S, k in keys.items()},
                           'prf0': hxl(thresha.PRF(keys[subsets[0]], bound)(uci, n)), 'parties': []}
                for i in range(m):
                    prfs = {S: thresha.PRF(k, bound) for S, k in keys.items() if i in S}
                    sh = thresha.np_pseudorandom_share(F, m, i, prfs, uci, n)
                    sh_l = thresha.pseudorandom_share(F, m, i, prfs, uci, n)
                    assert [int(v) for v in sh.value] == [int(v.value) for v in sh_l]
                    party = {'share': hxl(int(v) for v in sh.value)}
                    if bound == F.order and t > 0:
                        z = thresha.np_pseudorandom_share_0(F, m, i, prfs, uci, n)
                        zl = thresha.pseudorandom_share_zero(F, m, i, prfs, uci, n)
                        party['zero_np'] = hxl(int(v) for v in z.value)
                        party['zero_list'] = hxl(int(v.value) for v in zl)
                    setting['parties'].append(party)
                # the shares of all parties are a degree-t sharing of sum_S prl_S (resp. of 0)
                pts = [(i + 1, [int(x, 16) for x in setting['parties'][i]['share']]) for i in range(t + 1)]
                sec = thresha.recombine(F, pts)
                setting['secret'] = hxl(int(F(int(v)).value) for v in sec)
                case['settings'].append(setting)
        out[name] = case
    with open(os.path.join(OUT, fname), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', fname, os.path.getsize(os.path.join(OUT, fname)))


def matmul_cases():
    """F.array @ F.array from the reference (finfields.py:1126-1135; tests/test_finfields.py:389-404)."""
    out = {}
    rng = random.Random(79)          # own stream: this file does not depend on what ran before it
    for name, F in (('P61', finfields.GF(P61)), ('P64', finfields.GF(P64)), ('P127', finfields.GF(P127)),
                    ('P128', finfields.GF(P128)), ('P128G', finfields.GF(P128G)), ('P63G', finfields.GF(P63G)),
                    ('GF19', finfields.GF(19)), ('P31', finfields.GF(P31)),
                    ('GF2_8', finfields.GF(GF2X(BINARIES['GF2_8']))), ('GF2_64', finfields.GF(GF2X(BINARIES['GF2_64']))),
                    ('GF2_128', finfields.GF(GF2X(BINARIES['GF2_128'])))):
        q = F.order
        cases = []
        for (M, K, N) in ((1, 1, 1), (5, 7, 3), (3, 1, 4), (2, 20, 2)):
            A = [[rng.randrange(q) for _ in range(K)] for _ in range(M)]
            B = [[rng.randrange(q) for _ in range(N)] for _ in range(K)]
            A[0][0] = q - 1
            B[0][0] = q - 1
            C = F.array(A) @ F.array(B)
            v = (F.array(A[0]) @ F.array(B)).value          # 1-D @ 2-D
            cases.append({'M': M, 'K': K, 'N': N, 'A': [hxl(r) for r in A], 'B': [hxl(r) for r in B],
                          'C': [hxl(int(x) for x in r) for r in C.value], 'row0': hxl(int(x) for x in v)})
        out[name] = {'modulus': hx(int(F.modulus)), 'binary': not isinstance(F.modulus, int), 'cases': cases}
    with open(os.path.join(OUT, 'matmul.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote matmul.json', os.path.getsize(os.path.join(OUT, 'matmul.json')))


def linalg_cases():
    """np.linalg.det / inv / solve / matrix_power from the reference (finfields.py:872-978;
    tests/test_finfields.py:405-431), incl. matrices that need row swaps (zero pivots) and singular ones."""
    import numpy as np
    r = random.Random(77)
    out = {}
    for name, F in (('P61', finfields.GF(P61)), ('P64', finfields.GF(P64)), ('P128', finfields.GF(P128)),
                    ('P128G', finfields.GF(P128G)), ('P63G', finfields.GF(P63G)), ('GF19', finfields.GF(19)),
                    ('GF2', finfields.GF(2)), ('P31', finfields.GF(P31)),
                    ('GF2_8', finfields.GF(GF2X(BINARIES['GF2_8']))), ('GF2_64', finfields.GF(GF2X(BINARIES['GF2_64']))),
                    ('GF2_128', finfields.GF(GF2X(BINARIES['GF2_128'])))):
        q = F.order
        cases = []
        for n, kind in ((1, 'rand'), (2, 'swap'), (3, 'rand'), (3, 'swap'), (4, 'singular'), (5, 'swap2'), (6, 'rand'),
                        (9, 'rand'), (3, 'zero')):
            A = [[r.randrange(q) for _ in range(n)] for _ in range(n)]
            if kind.startswith('swap'):
                A[0][0] = 0
                if n > 2:
                    A[1][0] = 0                    # first usable pivot is two rows down
                if kind == 'swap2' and n > 3:
                    A[3][3] = 0
            elif kind == 'singular':
                A[n - 1] = [int(v) for v in (F.array(A[0]) + F.array(A[1])).value]    # dependent row
            elif kind == 'zero':
                A = [[0] * n for _ in range(n)]
            a = F.array(A)
            B = [[r.randrange(q) for _ in range(2)] for _ in range(n)]
            case = {'n': n, 'kind': kind, 'A': [hxl(row) for row in A], 'B': [hxl(row) for row in B],
                    'det': hx(np.linalg.det(a))}
            try:
                case['inv'] = [hxl(int(x) for x in row) for row in np.linalg.inv(a).value]
                case['solve'] = [hxl(int(x) for x in row) for row in np.linalg.solve(a, F.array(B)).value]
                case['pow_m3'] = [hxl(int(x) for x in row) for row in np.linalg.matrix_power(a, -3).value]
            except ZeroDivisionError as exc:
                case['error'] = str(exc)
            case['pow5'] = [hxl(int(x) for x in row) for row in np.linalg.matrix_power(a, 5).value]
            cases.append(case)
        stack = [[[r.randrange(q) for _ in range(3)] for _ in range(3)] for _ in range(4)]
        stack[2][0][0] = 0
        stack[3][2] = stack[3][1]
        dets = np.linalg.det(F.array(stack).reshape(2, 2, 3, 3))
        out[name] = {'modulus': hx(int(F.modulus)), 'binary': not isinstance(F.modulus, int), 'cases': cases,
                     'stack': [[hxl(row) for row in m] for m in stack],
                     'stack_det': [hxl(int(x) for x in row) for row in dets.value]}
    with open(os.path.join(OUT, 'linalg.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote linalg.json', os.path.getsize(os.path.join(OUT, 'linalg.json')))


def sqrt_cases():
    """F.array.sqrt() for p = 1 mod 4 (Cipolla-Lehmer, finfields.py:447-470 via :1459-1460), incl. zero,
    non-residues (the reference returns a deterministic non-root for those) and INV=True."""
    import numpy as np
    from mpyc import gmpy as g
    r = random.Random(79)
    out = {}

    def prime_1mod4_below(x):
        x -= 1
        while not (x % 4 == 1 and g.is_prime(x)):
            x -= 1
        return x

    for name, start in (('p5', 6), ('p13', 14), ('p17', 18), ('P31', 2**31), ('P40', 2**40), ('P61', 2**61),
                        ('P64', 2**64), ('P63G', 6616326157076047771), ('P96', 2**96), ('P128', 2**128),
                        ('P128G', 258797994007609146293811961253269568351)):
        p = prime_1mod4_below(start)
        F = finfields.GF(p)
        vals = [0, 1, 4 % p, p - 1, 2 % p, 3 % p] + [r.randrange(p) for _ in range(10)]
        sq = [pow(r.randrange(1, p), 2, p) for _ in range(8)]
        a = F.array(vals + sq)
        roots = a.sqrt()
        inv_roots = F.array(sq + [1]).sqrt(INV=True)
        out[name] = {'modulus': hx(p), 'a': hxl(vals + sq), 'sqrt': hxl(int(v) % p for v in roots.value),
                     'sq': hxl(sq + [1]), 'inv_sqrt': hxl(int(v) % p for v in inv_roots.value),
                     'is_sqr': [bool(b) for b in a.is_sqr()]}
    with open(os.path.join(OUT, 'sqrt.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote sqrt.json', os.path.getsize(os.path.join(OUT, 'sqrt.json')))


def npfunc_cases():
    """NumPy functions with arithmetic routed through __array_function__ (finfields.py:766-819, 1332-1356):
    convolve (runtime.np_convolve's local part), outer, prod, trace, sum along an axis."""
    import numpy as np
    r = random.Random(78)
    out = {}
    for name, F in (('P61', finfields.GF(P61)), ('P128', finfields.GF(P128)), ('GF19', finfields.GF(19)),
                    ('P31', finfields.GF(P31)), ('GF2_8', finfields.GF(GF2X(BINARIES['GF2_8']))),
                    ('GF2_64', finfields.GF(GF2X(BINARIES['GF2_64'])))):
        q = F.order
        a = [r.randrange(q) for _ in range(9)]
        v = [r.randrange(q) for _ in range(4)]
        m = [[r.randrange(q) for _ in range(5)] for _ in range(3)]
        fa, fv, fm = F.array(a), F.array(v), F.array(m)
        L = lambda x: hxl(int(e) for e in np.asarray(x.value).reshape(-1))
        out[name] = {'modulus': hx(int(F.modulus)), 'binary': not isinstance(F.modulus, int),
                     'a': hxl(a), 'v': hxl(v), 'm': [hxl(row) for row in m],
                     'conv_full': L(np.convolve(fa, fv)), 'conv_same': L(np.convolve(fa, fv, 'same')),
                     'conv_valid': L(np.convolve(fa, fv, 'valid')), 'conv_swapped': L(np.convolve(fv, fa)),
                     'outer': L(np.outer(fa, fv)), 'prod': hx(np.prod(fa)), 'prod_m': hx(fm.prod()),
                     'trace': hx(np.trace(fm)), 'sum0': L(np.sum(fm, axis=0)), 'sum1': L(fm.sum(axis=1)),
                     'sum': hx(np.sum(fm))}
    with open(os.path.join(OUT, 'npfuncs.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote npfuncs.json', os.path.getsize(os.path.join(OUT, 'npfuncs.json')))


def wire_cases():
    """The reference's marshalled forms of share rows:
      * field.to_bytes / from_bytes (finfields.py:91-102): byte_length little-endian bytes per element, incl. the
        2-byte elements of GF(2^8) (order.bit_length() = 9, finfields.py:524) and the 12-byte ones of 2^96-17;
      * the message frame `<qI{n}s` = pc (8 bytes signed) | payload size (4 bytes) | payload written by
        asyncoro.MessageExchanger.send (asyncoro.py:54-64), produced here by the reference's own send() on a
        recording transport, and its reassembly by data_received (asyncoro.py:66-106) from odd-sized chunks."""
    from mpyc import asyncoro
    r = random.Random(80)
    out = {'fields': {}, 'frames': []}
    for name, F in (('P61', finfields.GF(P61)), ('P64', finfields.GF(P64)), ('P96', finfields.GF(P96)),
                    ('P80', finfields.GF(P80)), ('P128', finfields.GF(P128)), ('P128G', finfields.GF(P128G)),
                    ('P31', finfields.GF(P31)), ('P40', finfields.GF(P40)), ('GF19', finfields.GF(19)),
                    ('GF2_8', finfields.GF(GF2X(BINARIES['GF2_8']))), ('GF2_16', finfields.GF(GF2X(BINARIES['GF2_16']))),
                    ('GF2_64', finfields.GF(GF2X(BINARIES['GF2_64']))), ('GF2_100', finfields.GF(GF2X(BINARIES['GF2_100']))),
                    ('GF2_128', finfields.GF(GF2X(BINARIES['GF2_128'])))):
        q = F.order
        vals = [0, 1, q - 1, q - 2, q >> 1] + [r.randrange(q) for _ in range(12)]
        data = F.to_bytes(vals)
        assert F.from_bytes(data) == vals
        # the same bytes from a field array's representation (what runtime.py:480,567,650 marshal in mix32-64bit mode)
        arr = F.array(vals)
        assert F.to_bytes([int(v) for v in arr.value]) == data
        out['fields'][name] = {'modulus': hx(int(F.modulus)), 'binary': not isinstance(F.modulus, int),
                               'byte_length': F.byte_length, 'values': hxl(vals), 'bytes': data.hex()}

    class Recorder:
        def __init__(self):
            self.chunks = []

        def write(self, data):
            self.chunks.append(bytes(data))

    class FakeRuntime:
        class options:
            no_prss = True

        def set_protocol(self, pid, proto):
            pass

    tx = asyncoro.MessageExchanger(FakeRuntime(), peer_pid=1)
    tx.transport = Recorder()
    msgs = [(0, b''), (1, b'\x00'), (-1, bytes(range(7))), (2**63 - 1, bytes(r.randrange(256) for _ in range(40))),
            (-2**63, out['fields']['P64']['bytes'] and bytes.fromhex(out['fields']['P64']['bytes'])),
            (123456789, bytes.fromhex(out['fields']['GF2_8']['bytes']))]
    for pc, payload in msgs:
        tx.send(pc, payload)
    stream = b''.join(tx.transport.chunks)
    assert tx.nbytes_sent == len(stream)
    # reassembly by the reference's receiver from chunks of 1, 2, 3, ... bytes
    rx = asyncoro.MessageExchanger(FakeRuntime(), peer_pid=0)
    pos, step = 0, 1
    while pos < len(stream):
        rx.data_received(stream[pos:pos + step])
        pos += step
        step += 1
    assert {pc: bytes(p) for pc, p in rx.buffers.items()} == dict(msgs)
    out['frames'] = [{'pc': pc, 'payload': payload.hex(), 'frame': chunk.hex()}
                     for (pc, payload), chunk in zip(msgs, tx.transport.chunks)]
    with open(os.path.join(OUT, 'wire.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote wire.json', os.path.getsize(os.path.join(OUT, 'wire.json')))


def roots_cases():
    """finfields.find_prime_root (finfields.py:311-344) incl. n-th roots of unity for n > 2 (:331-343), the primes
    SecFld / SecInt pick by default."""
    out = []
    for l, blum, n in ((8, True, 1), (8, False, 1), (16, True, 2), (61, True, 1), (64, True, 1), (80, True, 1), (128, True, 1),
                       (2, True, 1), (2, False, 1), (10, True, 3), (16, True, 5), (32, True, 7), (61, True, 4), (64, True, 17),
                       (100, True, 257), (128, True, 1000)):
        p, nn, w = finfields.find_prime_root(l, blum, n)
        out.append({'l': l, 'blum': blum, 'n': n, 'p': hx(int(p)), 'n_out': int(nn), 'w': hx(int(w))})
    with open(os.path.join(OUT, 'roots.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote roots.json', os.path.getsize(os.path.join(OUT, 'roots.json')))


ALL = ('main', 'prss', 'matmul', 'linalg', 'npfuncs', 'sqrt', 'wire', 'roots')

def wide_cases():
    """Three-limb primes (129..192 bits: the default fields of SecInt(97..160), finfields.py:311-344): the same
    case as fields.json holds for every other field, in a file of its own with its own random stream (the streams
    of the older fixtures do not move)."""
    global rng
    saved, rng = rng, random.Random(20260925 + 192)
    try:
        cases = {}
        for bits in (129, 136, 160, 192):
            p = int(finfields.find_prime_root(bits)[0])
            assert p.bit_length() == bits
            cases[f'P{bits}'] = field_case(f'P{bits}', finfields.GF(p), False, raw_width=192)
        # primes of no special shape: the root-of-unity prime SecInt(104, n=118) asks for (demos/np_lpsolver.py:74 with
        # its largest dataset; finfields.py:332-343), and the first prime above 2^128
        from mpyc import gmpy as g
        for name, p in (('P136R', int(finfields.find_prime_root(136, n=118)[0])), ('P129G', int(g.next_prime(2**128)))):
            cases[name] = field_case(name, finfields.GF(p), False, raw_width=192)
    finally:
        rng = saved
    with open(os.path.join(OUT, 'wide.json'), 'w') as fh:
        json.dump(cases, fh, separators=(',', ':'))
    print('wrote wide.json:', {k: v['modulus'] for k, v in cases.items()})
    # PRSS over three-limb fields (np_random_bits draws these in np_lpsolver -i5: runtime.py:4247-4250)
    prss_cases([(nm, finfields.GF(int(cases[nm]['modulus'], 16))) for nm in ('P136', 'P136R', 'P192')], 'prss_wide.json')


if __name__ == '__main__':
    args = sys.argv[1:]
    if 'wide' in args:
        wide_cases()
        sys.exit(0)
    if 'wire' in args:
        wire_cases()
        sys.exit(0)
    if 'roots' in args:
        roots_cases()
        sys.exit(0)
    if 'matmul' in args:
        matmul_cases()
        sys.exit(0)
    if 'sqrt' in sys.argv[1:]:
        sqrt_cases()
        sys.exit(0)
    if 'npfuncs' in sys.argv[1:]:
        npfunc_cases()
        sys.exit(0)
    if 'linalg' in sys.argv[1:]:
        linalg_cases()
        sys.exit(0)
    main()
    prss_cases()
    matmul_cases()
    linalg_cases()
    npfunc_cases()
    sqrt_cases()
    wire_cases()
    roots_cases()
    wide_cases()
