"""oracle/pyoracle.py -- CPU restatement of the reference's hot path with Python ints.

TEST INFRASTRUCTURE ONLY.  Nothing under mpyc_amd/ may import this module; it is
used by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as
the CHECKER of the HIP path, never as a computation path of the product.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.json, which tests/golden/make_golden.py produced by
running the real reference (lschoe/mpyc v0.11.2, pure Python) in the build
container, including the reference's own known-answer values
(tests/test_finfields.py:94-99, SURVEY.md appendix A.1/A.3/A.6, FIPS-197 S-box).

Each function cites the reference lines it restates (paths relative to the mpyc
checkout).  The code is a restatement, not a copy: plain integers and lists, no
field/element classes.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple


class Field:
    """Minimal description of a field: prime GF(p) or binary GF(2^n).

    modulus: the prime p, or the bit pattern of the irreducible polynomial.
    (finfields.py:347-363 pGF, :508-525 xGF)
    """

    def __init__(self, modulus: int, binary: bool = False):
        self.modulus = int(modulus)
        self.binary = bool(binary)
        if binary:
            self.n = self.modulus.bit_length() - 1
            self.order = 1 << self.n
        else:
            self.n = 1
            self.order = self.modulus

    def __repr__(self):
        return f"Field({'2^%d' % self.n if self.binary else self.modulus})"


# --------------------------------------------------------------------------
# GF(2)[x] on bit patterns
# --------------------------------------------------------------------------
def clmul(a: int, b: int) -> int:
    """Carry-less product of two bit patterns (gfpx.py:988-1003 _mul, incl. the
    squaring shortcut :1005-1015 which computes the same value)."""
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return r


def clmod(a: int, f: int) -> int:
    """Remainder of bit pattern a modulo f (gfpx.py:1025-1045 _mod)."""
    df = f.bit_length()
    while a.bit_length() >= df:
        a ^= f << (a.bit_length() - df)
    return a


def gf2_inv(a: int, f: int, n: int) -> int:
    """Inverse in GF(2^n) = a^(2^n - 2); equals gfpx.py:1084-1096 _invert."""
    if a == 0:
        raise ZeroDivisionError('inverse does not exist')
    r = 1
    e = (1 << n) - 2
    base = a
    while e:
        if e & 1:
            r = clmod(clmul(r, base), f)
        base = clmod(clmul(base, base), f)
        e >>= 1
    return r


# --------------------------------------------------------------------------
# scalar field operations (canonical ints in, canonical ints out)
# --------------------------------------------------------------------------
def reduce(F: Field, x: int) -> int:
    """Constructor reduction: finfields.py:724 `value %= modulus` (prime: Python
    modulo, negative wraps; binary: finfields.py:537-541 with gfpx.py:879-880
    _from_int = abs)."""
    if F.binary:
        return clmod(abs(int(x)), F.modulus)
    return int(x) % F.modulus


def add(F: Field, a: int, b: int) -> int:
    """finfields.py:1056-1063 / gfpx.py:982-984."""
    return (a ^ b) if F.binary else (a + b) % F.modulus


def sub(F: Field, a: int, b: int) -> int:
    """finfields.py:1075-1082 (char 2: gfpx.py:986 _sub = _add)."""
    return (a ^ b) if F.binary else (a - b) % F.modulus


def neg(F: Field, a: int) -> int:
    """finfields.py:1189-1192."""
    return a if F.binary else (-a) % F.modulus


def mul(F: Field, a: int, b: int) -> int:
    """finfields.py:1105-1112: product then reduction (binary: gfpx _mul, _mod)."""
    if F.binary:
        return clmod(clmul(a, b), F.modulus)
    return (a * b) % F.modulus


def inv(F: Field, a: int) -> int:
    """finfields.py:1416-1422 / gfpx.py:1084-1096."""
    if F.binary:
        return gf2_inv(a, F.modulus, F.n)
    if a % F.modulus == 0:
        raise ZeroDivisionError('inverse does not exist')
    return pow(a, -1, F.modulus)


def sqrt_prime(F: Field, a: int) -> int:
    """finfields.py:440-470: 0 -> 0; p = 3 mod 4: a^((p+1)/4); p = 1 mod 4: Cipolla-Lehmer with the
    smallest b such that b^2 - 4a is a non-residue, X^((p+1)/2) mod X^2 - bX + a by the same ladder."""
    p = F.modulus
    if a == 0 or p == 2:
        return a
    if p & 3 == 3:
        return pow(a, (p + 1) >> 2, p)
    b = 1
    while pow((b * b - 4 * a) % p, (p - 1) >> 1, p) != p - 1:
        b += 1
    u, v = 0, 1
    e = (p + 1) >> 1
    for i in range(e.bit_length() - 1, -1, -1):
        u2 = u * u % p
        u, v = ((u << 1) * v + b * u2) % p, (v * v - a * u2) % p
        if (e >> i) & 1:
            u, v = (v + b * u) % p, (-a * u) % p
    return v


def vec(op, F: Field, a: Sequence[int], b: Sequence[int]) -> List[int]:
    return [op(F, x, y) for x, y in zip(a, b)]


# --------------------------------------------------------------------------
# Gaussian elimination (finfields.py:872-955)
# --------------------------------------------------------------------------
def gauss_solve(F: Field, A: Sequence[Sequence[int]], B: Sequence[Sequence[int]]) -> List[List[int]]:
    """finfields.py:872-908: LU in place with the reciprocal of the pivot stored on the diagonal, first
    nonzero entry at or below the diagonal as pivot, back substitution.  Raises ZeroDivisionError."""
    n = len(A)
    M = [list(ra) + list(rb) for ra, rb in zip(A, B)]
    for k in range(n):
        if M[k][k] == 0:                                           # :885-893
            for x in range(k + 1, n):
                if M[x][k] != 0:
                    break
            else:
                raise ZeroDivisionError('no inverse exists')
            M[k], M[x] = M[x], M[k]
        M[k][k] = inv(F, M[k][k])                                  # :894
        for i in range(k + 1, n):                                  # :895-896
            M[i][k] = mul(F, M[i][k], M[k][k])
            for j in range(k + 1, len(M[i])):
                M[i][j] = sub(F, M[i][j], mul(F, M[i][k], M[k][j]))
    for i in range(n - 1, -1, -1):                                 # :899-901
        for j in range(n, len(M[i])):
            acc = M[i][j]
            for c in range(i + 1, n):
                acc = sub(F, acc, mul(F, M[i][c], M[c][j]))
            M[i][j] = mul(F, acc, M[i][i])
    return [row[n:] for row in M]


def gauss_det(F: Field, A: Sequence[Sequence[int]]) -> int:
    """finfields.py:931-949: product of the pivots; row swaps do NOT change the sign (reference behaviour)."""
    n = len(A)
    M = [list(r) for r in A]
    for k in range(n):
        if M[k][k] == 0:
            for x in range(k + 1, n):
                if M[x][k] != 0:
                    break
            else:
                return 0
            M[k], M[x] = M[x], M[k]
        inv_k = inv(F, M[k][k])
        for i in range(k + 1, n):
            M[i][k] = mul(F, M[i][k], inv_k)
            for j in range(k + 1, n):
                M[i][j] = sub(F, M[i][j], mul(F, M[i][k], M[k][j]))
    d = 1
    for k in range(n):
        d = mul(F, d, M[k][k])
    return d


def matmul(F: Field, A, B):
    """finfields.py:1126-1135."""
    return [[_dot(F, ra, [rb[j] for rb in B]) for j in range(len(B[0]))] for ra in A]


def _dot(F: Field, a, b):
    acc = 0
    for x, y in zip(a, b):
        acc = add(F, acc, mul(F, x, y))
    return acc


# --------------------------------------------------------------------------
# Shamir share generation
# --------------------------------------------------------------------------
def np_random_split(F: Field, s: Sequence[int], t: int, m: int, draws: Sequence[int]) -> List[List[int]]:
    """thresha.py:47-64.  `draws` is the sequence secrets.randbelow(order) returned,
    in call order: C[j][h] = draws[j*n + h] (row-major (t, n), thresha.py:60).
    share_i[h] = (s[h] + sum_j C[j][h] * x_i^(j+1)) mod modulus with x_i = i+1
    (for GF(2^n): the polynomial with bit pattern i+1, thresha.py:61)."""
    n = len(s)
    assert 0 <= t < m and len(draws) >= t * n
    out = [[0] * n for _ in range(m)]
    for i in range(m):
        x = i + 1
        for h in range(n):
            if F.binary:
                acc = s[h]
                xp = 1
                for j in range(t):
                    xp = clmul(xp, x)                    # unreduced powers, as np.vander does
                    acc ^= clmul(draws[j * n + h], xp)
                out[i][h] = clmod(acc, F.modulus)
            else:
                acc = s[h]
                xp = 1
                for j in range(t):
                    xp *= x
                    acc += draws[j * n + h] * xp
                out[i][h] = acc % F.modulus
    return out


def random_split(F: Field, s: Sequence[int], t: int, m: int, draws: Sequence[int]) -> List[List[int]]:
    """thresha.py:23-44 (list path).  For secret h the t draws are consecutive,
    c = draws[h*t:(h+1)*t], and Horner y = (y + c_j) * x puts c[0] on X^t."""
    n = len(s)
    out = [[0] * n for _ in range(m)]
    for h in range(n):
        c = draws[h * t:(h + 1) * t]
        for i in range(m):
            x = i + 1
            y = 0
            for cj in c:
                y = clmul(y ^ cj, x) if F.binary else (y + cj) * x
            out[i][h] = clmod(y ^ s[h], F.modulus) if F.binary else (y + s[h]) % F.modulus
    return out


def list_to_np_draws(draws: Sequence[int], t: int, n: int) -> List[int]:
    """Permutation that feeds the np-convention kernel so that it reproduces the
    list path: np coefficient row j (X^(j+1)) must be list draw c_h[t-1-j]
    (SURVEY.md appendix A.1)."""
    return [draws[h * t + (t - 1 - j)] for j in range(t) for h in range(n)]


# --------------------------------------------------------------------------
# Lagrange recombination
# --------------------------------------------------------------------------
def recombination_vector(F: Field, xs: Sequence[int], x_r: int) -> List[int]:
    """thresha.py:67-85: lambda_i = prod_{j != i} (x_r - x_j) / (x_i - x_j), in the
    order of xs."""
    xs = [reduce(F, x) for x in xs]
    x_r = reduce(F, x_r)
    out = []
    for i, xi in enumerate(xs):
        num, den = 1, 1
        for j, xj in enumerate(xs):
            if i != j:
                num = mul(F, num, sub(F, x_r, xj))
                den = mul(F, den, sub(F, xi, xj))
        out.append(mul(F, num, inv(F, den)))
    return out


def np_recombine(F: Field, points: Sequence[Tuple[int, Sequence[int]]], x_rs=0):
    """thresha.py:119-132: sums = Lambda @ rows, reduced once at the end
    (finfields.py:1126-1135).  Returns a list (x_rs scalar) or list of lists."""
    xs = [p[0] for p in points]
    rows = [[reduce(F, v) for v in p[1]] for p in points]     # field.array(shares), :128
    scalar = not isinstance(x_rs, list)
    xr_list = [x_rs] if scalar else x_rs
    n = len(rows[0])
    outs = []
    for x_r in xr_list:
        lam = recombination_vector(F, xs, x_r)
        row = []
        for h in range(n):
            if F.binary:
                acc = 0
                for j in range(len(rows)):
                    acc ^= clmul(lam[j], rows[j][h])
                row.append(clmod(acc, F.modulus))
            else:
                acc = 0
                for j in range(len(rows)):
                    acc += lam[j] * rows[j][h]
                row.append(acc % F.modulus)
        outs.append(row)
    return outs[0] if scalar else outs


def recombine_unreduced(F: Field, points: Sequence[Tuple[int, Sequence[int]]], x_r=0) -> List[int]:
    """thresha.py:88-116 on raw ints: the sums are NOT reduced (:109); callers
    reduce afterwards (runtime.py:588,682)."""
    xs = [p[0] for p in points]
    lam = recombination_vector(F, xs, x_r)
    n = len(points[0][1])
    out = []
    for h in range(n):
        acc = 0
        for j, (_, row) in enumerate(points):
            acc = (acc ^ clmul(row[h], lam[j])) if F.binary else acc + row[h] * lam[j]
        out.append(acc)
    return out


# --------------------------------------------------------------------------
# GF(2^8) S-box on public bytes (demos/np_aes.py:37-43)
# --------------------------------------------------------------------------
AES_MOD = 0x11b


def aes_affine_rows() -> Tuple[List[int], int]:
    """A = circulant([1,0,0,0,1,1,1,1]) (np_aes.py:23-30: row j is the first row
    rolled right by j), B = [1,1,0,0,0,1,1,0] little-endian -> 0x63."""
    r = [1, 0, 0, 0, 1, 1, 1, 1]
    rows = []
    for j in range(8):
        rows.append(sum(r[(c - j) % 8] << c for c in range(8)))
    b = sum(bit << i for i, bit in enumerate([1, 1, 0, 0, 0, 1, 1, 0]))
    return rows, b


def pow254(F: Field, a: int) -> int:
    """runtime.py:1356-1367 addition chain (11 multiplications)."""
    d = a
    c = mul(F, d, d)
    c = mul(F, c, c)
    c = mul(F, c, c)
    c = mul(F, c, d)
    c = mul(F, c, c)
    c, d = mul(F, c, c), mul(F, c, d)
    c, d = mul(F, c, c), mul(F, c, d)
    c = mul(F, c, d)
    return mul(F, c, c)


def sbox(x: Iterable[int], rows8: Sequence[int] = None, b: int = None) -> List[int]:
    """np_aes.py:37-43: bits(x^254) -> A @ bits + B -> byte."""
    F = Field(AES_MOD, binary=True)
    if rows8 is None:
        rows8, b = aes_affine_rows()
    out = []
    for v in x:
        iv = pow254(F, v)
        y = 0
        for r in range(8):
            y |= (bin(iv & rows8[r]).count('1') & 1) << r
        out.append(y ^ b)
    return out


# --------------------------------------------------------------------------
# Pseudorandom secret sharing (thresha.py:135-266)
# --------------------------------------------------------------------------
def prf_values(key: bytes, bound: int, s: bytes, n: int) -> List[int]:
    """thresha.PRF.__call__ (thresha.py:238-266): n values in range(bound) from SHAKE128(key+s)."""
    from hashlib import shake_128
    l = ((bound - 1).bit_length() + 7) // 8
    if bound & (bound - 1):
        l += len(key)
    if n == 0:
        return []
    if l == 0:
        return [0] * n
    dk = shake_128(key + s).digest(n * l)
    return [int.from_bytes(dk[i:i + l], 'little') % bound for i in range(0, n * l, l)]


def f_S_i(F: Field, m: int, i: int, S) -> int:
    """thresha.py:135-141: value at x = i+1 of the polynomial that is 1 at 0 and 0 at every party
    outside S (reduced; the reference keeps the unreduced integer, congruent)."""
    pts = [(0, [1])] + [(x + 1, [0]) for x in range(m) if x not in S]
    return reduce(F, recombine_unreduced(F, pts, i + 1)[0])


def np_pseudorandom_share(F: Field, m: int, i: int, keys: dict, bound: int, uci: bytes, n: int) -> List[int]:
    """thresha.py:163-173."""
    out = [0] * n
    for S, key in keys.items():
        f = f_S_i(F, m, i, S)
        prl = prf_values(key, bound, uci, n)
        for h in range(n):
            out[h] = add(F, out[h], mul(F, reduce(F, prl[h]), f))
    return out


def np_pseudorandom_share_0(F: Field, m: int, i: int, keys: dict, bound: int, uci: bytes, n: int,
                            list_convention: bool = False) -> List[int]:
    """thresha.py:201-217 (np: draw j of secret h multiplies (i+1)^(j+1)); with list_convention the
    Horner order of thresha.py:176-198 (draw j multiplies (i+1)^(d-j))."""
    d = m - len(next(iter(keys)))
    i1 = reduce(F, i + 1)
    out = [0] * n
    for S, key in keys.items():
        f = f_S_i(F, m, i, S)
        prl = prf_values(key, bound, uci, n * d)
        for h in range(n):
            acc = 0
            for j in range(d):
                power = d - j if list_convention else j + 1
                w = 1
                for _ in range(power):
                    w = mul(F, w, i1)
                acc = add(F, acc, mul(F, reduce(F, prl[h * d + j]), w))
            out[h] = add(F, out[h], mul(F, acc, f))
    return out


# --------------------------------------------------------------------------
# PRSS in production mode: the reference's combination (thresha.py:163-173, 201-217) over a COUNTER-MODE PRF -- one
# ChaCha stream per subset key (mpyc_amd/csrc/kernels.hpp k_prss_chacha).  No reference counterpart for the PRF: pinned
# to RFC 8439 (test vector 2.3.2 in tests/test_prss_chacha.py) and to the layout restated here and in fforacle.c.
# --------------------------------------------------------------------------
def chacha_block(key32: bytes, counter: int, nonce8: bytes, rounds: int = 20) -> bytes:
    """RFC 8439 section 2.3 block function; state words 12, 13 = 64-bit block counter, 14, 15 = nonce."""
    M = 0xffffffff
    init = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574]
    init += [int.from_bytes(key32[4 * i:4 * i + 4], 'little') for i in range(8)]
    init += [counter & M, (counter >> 32) & M, int.from_bytes(nonce8[:4], 'little'), int.from_bytes(nonce8[4:8], 'little')]
    x = list(init)

    def rot(v, c):
        return ((v << c) & M) | (v >> (32 - c))

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return b''.join(((a + b) & M).to_bytes(4, 'little') for a, b in zip(x, init))


def prss_chacha_layout(l: int) -> Tuple[int, int]:
    """(TB, DPT): blocks per tile and draws per tile for l-byte draws (LW = ceil(l/4) words each): TB in {1,2,3} with
    the most draws per block, DPT = min(8, 16 TB // LW)."""
    lw = (l + 3) // 4
    best = None
    for tb in (1, 2, 3):
        dpt = min(8, 16 * tb // lw)
        if best is None or dpt * best[0] > best[1] * tb:
            best = (tb, dpt)
    return best


def prss_chacha_stream_key(key: bytes, s: bytes) -> bytes:
    """key (32) + nonce (8) of the stream of PRF key `key` on common input `s` (mpyc_amd/thresha.py)."""
    from hashlib import shake_128
    return shake_128(b'mpyc_amd prss chacha v1\0' + len(key).to_bytes(2, 'little') + key + s).digest(40)


def prf_values_chacha(key: bytes, bound: int, s: bytes, n: int, d: int = 1, rounds: int = 20) -> List[int]:
    """The n*d draws (h, j) -> index h*d + j of one subset key in production mode: l bytes of keystream `% bound`, l as in
    thresha.py:232-236."""
    l = ((bound - 1).bit_length() + 7) // 8
    if bound & (bound - 1):
        l += len(key)
    if l == 0:
        return [0] * (n * d)
    k40 = prss_chacha_stream_key(key, s)
    tb, dpt = prss_chacha_layout(l)
    lw = (l + 3) // 4
    out = []
    cache = {}
    for h in range(n):
        tile, slot = divmod(h, dpt)
        for j in range(d):
            if (tile, j) not in cache:
                cache = {kk: vv for kk, vv in cache.items() if kk[0] == tile}
                cache[(tile, j)] = b''.join(chacha_block(k40[:32], (tile * d + j) * tb + b, k40[32:], rounds) for b in range(tb))
            ksb = cache[(tile, j)]
            out.append(int.from_bytes(ksb[4 * slot * lw:4 * slot * lw + l], 'little') % bound)
    return out


def np_pseudorandom_share_chacha(F: Field, m: int, i: int, keys: dict, bound: int, uci: bytes, n: int, rounds: int = 20):
    """thresha.py:163-173 with the production PRF."""
    out = [0] * n
    for S, key in keys.items():
        f = f_S_i(F, m, i, S)
        prl = prf_values_chacha(key, bound, uci, n, 1, rounds)
        for h in range(n):
            out[h] = add(F, out[h], mul(F, reduce(F, prl[h]), f))
    return out


def np_pseudorandom_share_0_chacha(F: Field, m: int, i: int, keys: dict, bound: int, uci: bytes, n: int, rounds: int = 20,
                                   list_convention: bool = False):
    """thresha.py:201-217 with the production PRF (draw j of secret h multiplies (i+1)^(j+1))."""
    d = m - len(next(iter(keys)))
    i1 = reduce(F, i + 1)
    out = [0] * n
    for S, key in keys.items():
        f = f_S_i(F, m, i, S)
        prl = prf_values_chacha(key, bound, uci, n, d, rounds)
        for h in range(n):
            acc = 0
            for j in range(d):
                power = d - j if list_convention else j + 1
                w = 1
                for _ in range(power):
                    w = mul(F, w, i1)
                acc = add(F, acc, mul(F, reduce(F, prl[h * d + j]), w))
            out[h] = add(F, out[h], mul(F, acc, f))
    return out


# --------------------------------------------------------------------------
# AES-128 (FIPS-197), as demos/np_aes.py:55-86 computes it on public values
# --------------------------------------------------------------------------
def aes128_encrypt(key: Sequence[int], block: Sequence[int]) -> List[int]:
    """key, block: 16 bytes each (FIPS-197 byte order: byte p = s[p % 4][p // 4]).  S-box = sbox() above
    (x^254 + affine map, np_aes.py:37-43); MixColumns = circulant([2,3,1,1]) (np_aes.py:33)."""
    F = Field(0x11b, True)
    sb = sbox(range(256))
    w = [list(key[4 * c:4 * c + 4]) for c in range(4)]                  # columns
    for i in range(4, 44):                                              # np_aes.py:55-72
        tcol = list(w[i - 1])
        if i % 4 == 0:
            tcol = [sb[v] for v in tcol]
            tcol = tcol[1:] + tcol[:1]
            rc = 1
            for _ in range(i // 4 - 1):
                rc = mul(F, rc, 2)
            tcol[0] ^= rc
        w.append([a ^ b for a, b in zip(tcol, w[i - 4])])
    K = [[w[4 * j + c][r] for c in range(4) for r in range(4)] for j in range(11)]
    s = [a ^ b for a, b in zip(block, K[0])]
    C = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]
    for rnd in range(1, 11):                                            # np_aes.py:75-86
        s = [sb[v] for v in s]
        sh = [[s[r + 4 * ((c + r) % 4)] for c in range(4)] for r in range(4)]     # sh[r][c]
        if rnd < 10:
            sh = [[_xor_all(mul(F, C[r][k], sh[k][c]) for k in range(4)) for c in range(4)] for r in range(4)]
        s = [sh[r][c] ^ K[rnd][r + 4 * c] for c in range(4) for r in range(4)]
    return s


def _xor_all(it):
    acc = 0
    for v in it:
        acc ^= v
    return acc
