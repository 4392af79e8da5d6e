"""Local parts of the reference's secure-array protocols, composed from the kernels, for ALL m parties of a
computation held on one GPU.

MPyC runs one process per party and moves shares over TCP (runtime.py); what each party computes between
two messages is exactly the kernels of this package.  The functions below chain those kernels in the order
the reference's protocols do, for every party at once, with the "network" replaced by handing device rows
from one party's share matrix to the other parties' recombination.  They exist to (a) test the gate
compositions end to end (open(result) must equal the plaintext function) and (b) measure the compute of a
whole protocol (bench.py: the np_aes S-box layer over secure bytes, BASELINE config 5).

A secret-shared array is a list of m DevArrays: entry i is the share of party i+1 (x-coordinate i+1,
thresha.py:61).  Nothing here runs on the CPU except Lagrange vectors (a handful of scalars, as in the
reference, thresha.py:67-85).
"""
from typing import List, Optional, Sequence

from . import thresha
from .engine import DevArray, FieldContext

Shares = List[DevArray]


def _lagrange(field, xs: Sequence[int]) -> List[int]:
    return [int(v) for v in thresha._recombination_vector(field, tuple(xs), 0)]


def share(ctx: FieldContext, x: DevArray, t: int, m: int, rng=None) -> Shares:
    """np_random_split (thresha.py:47-64) with coefficients from the device CSPRNG; row i -> party i+1.
    rng: an engine.RngState (device-resident generator, required inside HIP-graph capture) or None (fresh
    host key per call)."""
    mtx = ctx.split_rng(x, t, m, state=rng)
    return [mtx.row(i) for i in range(m)]


def open_(ctx: FieldContext, field, xs: Shares, t: int, degree: Optional[int] = None) -> DevArray:
    """runtime.output (runtime.py:560-600): recombine the first degree+1 shares at x = 0."""
    k = (t if degree is None else degree) + 1
    return ctx.recombine(xs[:k], _lagrange(field, range(1, k + 1)))


def add_public(ctx: FieldContext, xs: Shares, c: DevArray) -> Shares:
    """Shamir: every party adds the public value to its share."""
    return [ctx.add(x, c) for x in xs]


def multiply(ctx: FieldContext, field, xs: Shares, ys: Shares, t: int, rng=None) -> Shares:
    """runtime.np_multiply (runtime.py:1096-1141) = local product of shares (degree 2t) + _reshare
    (runtime.py:603-689): each of the first 2t+1 parties re-shares its product with a fresh degree-t
    polynomial -- the product is formed inside the share-generation kernel (ffgpu_mul_split_rng) --, then
    every party recombines the 2t+1 sub-shares it received with the Lagrange vector for x = 0."""
    m = len(xs)
    k = 2 * t + 1
    if m < k:
        raise ValueError('multiplication needs m >= 2t+1 parties')
    lam = _lagrange(field, range(1, k + 1))
    sub = [ctx.split_rng(xs[i], t, m, mul_by=ys[i], state=rng) for i in range(k)]            # sender i -> row j for party j
    return [ctx.recombine([sub[i].row(j) for i in range(k)], lam) for j in range(m)]


# ---- chains of multiplications with the recombined share kept in registers ------------------------------
class Pending:
    """A secret-shared value each party holds as a RECOMBINATION it has not performed yet: party j's share
    is sum_i lam[i] * rows[j][i] (rows[j] = the sub-shares party j received).  A materialised sharing is the
    1-row case with lam = [1]."""

    __slots__ = ('rows', 'lam')

    def __init__(self, rows, lam):
        self.rows, self.lam = rows, lam

    @classmethod
    def of(cls, xs: Shares) -> 'Pending':
        return cls([[x] for x in xs], [1])


def materialize(ctx: FieldContext, x) -> Shares:
    """Pending -> the parties' shares (the recombination of runtime._reshare, runtime.py:677-689)."""
    if not isinstance(x, Pending):
        return x
    if len(x.lam) == 1 and x.lam[0] == 1:
        return [r[0] for r in x.rows]
    return [ctx.recombine(r, x.lam) for r in x.rows]


def multiply_pending(ctx: FieldContext, field, x, y, t: int, rng=None) -> Pending:
    """multiply() for operands that may be Pending (y is x: a squaring): the first 2t+1 parties recombine both
    factors, multiply and re-share in ONE kernel (ffgpu_gate_rng); the result stays Pending, so the next
    multiplication consumes the new sub-shares directly.  Needs t <= 3."""
    px = x if isinstance(x, Pending) else Pending.of(x)
    py = None if y is x else (y if isinstance(y, Pending) else Pending.of(y))
    m = len(px.rows)
    k = 2 * t + 1
    if m < k:
        raise ValueError('multiplication needs m >= 2t+1 parties')
    lam = _lagrange(field, range(1, k + 1))
    sub = [ctx.gate(px.rows[i], px.lam, py.rows[i] if py else None, py.lam if py else None, t, m, state=rng)
           for i in range(k)]
    return Pending([[sub[i].row(j) for i in range(k)] for j in range(m)], lam)


def pow254_fused(ctx: FieldContext, field, xs: Shares, t: int, rng=None) -> Shares:
    """pow254() with every intermediate share kept Pending: 11 fused gate kernels + 1 recombination per party
    instead of 11 x (share generation + recombination)."""
    mul = lambda a, b: multiply_pending(ctx, field, a, b, t, rng)
    d = Pending.of(xs)
    c = mul(d, d)
    c = mul(c, c)
    c = mul(c, c)
    c = mul(c, d)
    c = mul(c, c)
    c, d = mul(c, c), mul(c, d)
    c, d = mul(c, c), mul(c, d)
    c = mul(c, d)
    return materialize(ctx, mul(c, c))


def matmul(ctx: FieldContext, field, xs: Shares, ys: Shares, M: int, K: int, N: int, t: int, rng=None) -> Shares:
    """runtime.np_matmul (runtime.py:2481-2541): every party multiplies its share matrices locally
    (`A @ B` then one reduction, :2531 -- the dense-product kernel) and the degree-2t result is re-shared as
    in multiply().  xs: (M, K) and ys: (K, N) row-major share matrices; returns shares of the (M, N) product."""
    m = len(xs)
    k = 2 * t + 1
    if m < k:
        raise ValueError('multiplication needs m >= 2t+1 parties')
    lam = _lagrange(field, range(1, k + 1))
    sub = [ctx.split_rng(ctx.matmul(xs[i], ys[i], M, K, N), t, m, state=rng) for i in range(k)]
    return [ctx.recombine([sub[i].row(j) for i in range(k)], lam) for j in range(m)]


def pow254(ctx: FieldContext, field, xs: Shares, t: int, rng=None) -> Shares:
    """x^254 by the reference's addition chain (runtime.py:1356-1367): 11 secure multiplications.  The
    reference stacks (c, d) in two rounds to halve the number of MESSAGES; locally the stacked product is two
    products, issued here as such (no concatenation traffic)."""
    mul = lambda a, b: multiply(ctx, field, a, b, t, rng)
    d = xs
    c = mul(d, d)
    c = mul(c, c)
    c = mul(c, c)
    c = mul(c, d)
    c = mul(c, c)
    c, d = mul(c, c), mul(c, d)
    c, d = mul(c, c), mul(c, d)
    c = mul(c, d)
    return mul(c, c)


def to_bits_gf256(ctx: FieldContext, field, xs: Shares, rbits: Shares, t: int) -> Shares:
    """runtime.np_to_bits for a binary field (runtime.py:4411-4423): with shared random bits r_j,
    r = sum_j r_j 2^j, open c = x + r, return bits(c) + r_bits.  xs: n bytes, rbits: 8n bit shares."""
    weights = [[1 << j for j in range(8)]]
    masked = [ctx.add(x, ctx.group_matvec(r, weights)) for x, r in zip(xs, rbits)]     # a + r_modl
    c = open_(ctx, field, masked, t)
    return [ctx.to_bits(c, addend=r) for r in rbits]                               # c_bits + r_bits


def from_bits(ctx: FieldContext, bits: Shares, l: int = 8) -> Shares:
    """runtime.np_from_bits (runtime.py:4475-4484): sum_j x_j 2^j over the last axis (local)."""
    weights = [[1 << j for j in range(l)]]
    return [ctx.group_matvec(b, weights) for b in bits]


# ---- the same layer with ALL parties in every launch ---------------------------------------------------------
# The per-party functions above issue one launch per party and step (what each MPyC party does in its own
# process).  When all m parties of a computation sit on one GPU the parties' launches of a step are identical
# up to pointers, so they become grid rows of ONE launch: shares of all parties are the rows of one DevMatrix,
# sub-shares live in an (m * k)-row block used as [recipient][sender], and a layer of the x^254 chain is one
# ffgpu_gate_rng_batch.  The two local steps around the opening inside np_to_bits are one kernel each
# (ffgpu_gf256_mask_open, ffgpu_gf256_bits_affine_fold).  13 launches per S-box layer instead of 49.
def as_matrix(ctx: FieldContext, xs: Shares):
    """The parties' shares as the rows of one DevMatrix: a view when they already are equally spaced rows of one
    allocation (what share() / split return), else a copy."""
    import torch
    from .engine import DevMatrix, limbs_of
    m, n, eb = len(xs), xs[0].n, ctx.elem_bytes
    unit = xs[0].t.element_size()                     # bytes of one tensor element (8 for the 16- and 24-byte layouts)
    step = (xs[1].ptr - xs[0].ptr) if m > 1 else 0
    lb = limbs_of(eb)
    same_storage = all(x.t.untyped_storage().data_ptr() == xs[0].t.untyped_storage().data_ptr() for x in xs)
    if m > 1 and same_storage and step > 0 and step % eb == 0 and step // eb >= n and \
            all(xs[i].ptr - xs[0].ptr == i * step for i in range(m)) and all(x.n == n for x in xs) and \
            xs[0].t.storage_offset() * unit + (m - 1) * step + n * eb <= xs[0].t.untyped_storage().nbytes():
        stride = step // eb
        shape, strides = ((m, stride, lb), (stride * lb, lb, 1)) if lb else ((m, stride), (stride, 1))
        span = xs[0].t.storage_offset() + ((m - 1) * stride + stride) * (lb or 1)
        if span * unit <= xs[0].t.untyped_storage().nbytes():
            return DevMatrix(ctx, torch.as_strided(xs[0].t, shape, strides, xs[0].t.storage_offset()), m, n, stride)
    mtx = ctx.empty_matrix(m, n)
    for i, x in enumerate(xs):
        mtx.row(i).t.copy_(x.t)
    return mtx


class _Block:
    """Sub-shares of one re-sharing round, all parties: row j * k + s of `mtx` = the sub-share sender s+1 dealt to
    party j+1."""

    __slots__ = ('mtx', 'k', 'lam')

    def __init__(self, mtx, k, lam):
        self.mtx, self.k, self.lam = mtx, k, lam

    def rows_of(self, j: int):
        return [self.mtx.row(j * self.k + s) for s in range(self.k)]


def _gate_all(ctx: FieldContext, x, y, t: int, m: int, lam, rng):
    """One secure multiplication (runtime.py:1096-1141 + :603-689) for all parties: the k = 2t+1 senders recombine
    their factors (pending _Blocks or plain share matrices), multiply and re-share in ONE launch."""
    k = 2 * t + 1

    def operand(v):
        if isinstance(v, _Block):
            return v.rows_of(0), v.lam, v.k * v.mtx.stride
        return [v.row(0)], [1], v.stride
    ra, la, sa = operand(x)
    rb, lb, sb = (None, None, 0) if y is x else operand(y)
    n = ra[0].n
    out = ctx.empty_matrix(m * k, n)
    ctx.gate_batch(ra, la, sa, rb, lb, sb, t, m, k, out, k, state=rng)
    return _Block(out, k, lam)


def sbox_layer_all(ctx: FieldContext, field, xs, rbits, t: int, A: Sequence[Sequence[int]], B: Sequence[int], rng=None,
                   fused: bool = True):
    """sbox_layer() with all parties in every launch (see above): xs / rbits are Shares lists or DevMatrix (one row
    per party); returns a DevMatrix of the parties' shares of S-box(x).  GF(2^8), t <= 3.  fused (default): the
    whole layer as ONE kernel (ffgpu_gf256_sbox_layer) where it applies; fused=False: the 13-launch composition
    (11 batched chain gates + masked opening + bits/affine/fold)."""
    from .engine import DevMatrix
    X = xs if isinstance(xs, DevMatrix) else as_matrix(ctx, xs)
    R = rbits if isinstance(rbits, DevMatrix) else as_matrix(ctx, rbits)
    m, k = X.rows, 2 * t + 1
    if m < k or R.rows != m:
        raise ValueError('multiplication needs m >= 2t+1 parties, and bit shares for each of them')
    lam = _lagrange(field, range(1, k + 1))
    if fused and ctx.elem_bytes == 1:
        # everything below is element-wise: one kernel carries the parties' shares through the whole layer in registers
        try:
            out = ctx.gf256_sbox_layer(X, R, t, lam, _lagrange(field, range(1, t + 2)), A, B, state=rng)
            if rng is not None:
                rng.commit()
            return out
        except NotImplementedError:
            pass                                        # shape not covered: per-step kernels
    mul = lambda a, b: _gate_all(ctx, a, b, t, m, lam, rng)
    d = X                                               # x^254 by the reference's addition chain, runtime.py:1356-1367
    c = mul(d, d)
    c = mul(c, c)
    c = mul(c, c)
    c = mul(c, d)
    c = mul(c, c)
    c, d = mul(c, c), mul(c, d)
    c, d = mul(c, c), mul(c, d)
    c = mul(c, d)
    y = mul(c, c)                                       # pending: party j holds sum_s lam[s] * y.rows_of(j)[s]
    mu = _lagrange(field, range(1, t + 2))              # opening from the first t+1 parties (runtime.py:582-585)
    rows, coefs = [], []
    for p in range(t + 1):
        for s, r in enumerate(y.rows_of(p)):
            rows.append(r)
            coefs.append(int(field(mu[p]) * field(lam[s])))
    if rng is not None:
        rng.commit()                                    # one nonce update for the 11 gates (deferred advance)
    opened = ctx.gf256_mask_open(rows, coefs, [R.row(p) for p in range(t + 1)], mu[:t + 1])
    return ctx.gf256_bits_affine_fold(opened, R, A, B)


def sbox_layer(ctx: FieldContext, field, xs: Shares, rbits: Shares, t: int, A: Sequence[Sequence[int]],
               B: Sequence[int], fused: bool = True, rng=None, chain: bool = True) -> Shares:
    """The AES S-box on secret-shared bytes, as demos/np_aes.py:37-43:
    x = np_to_bits(x**254); x = A @ x + B over GF(2) (on bit shares, local); x = np_from_bits(x)."""
    y = pow254_fused(ctx, field, xs, t, rng) if chain and t <= 3 else pow254(ctx, field, xs, t, rng)
    bits = to_bits_gf256(ctx, field, y, rbits, t)
    if fused:
        return [ctx.bit_affine(b, A, B, from_bits=True) for b in bits]           # both local steps in one pass
    bits = [ctx.group_matvec(b, A, B) for b in bits]
    return from_bits(ctx, bits)


def _sbox_any(ctx: FieldContext, field, xs: Shares, rbits: Shares, t: int, A, B, rng) -> Shares:
    """S-box layer for the AES functions below: all parties per launch when the fused chain applies (t <= 3)."""
    if t <= 3:
        out = sbox_layer_all(ctx, field, xs, rbits, t, A, B, rng)
        return [out.row(i) for i in range(out.rows)]
    return sbox_layer(ctx, field, xs, rbits, t, A, B, rng=rng)


# ---- AES-128 on secret-shared blocks (demos/np_aes.py:55-86) ------------------------------------------
# Layout: position-major.  A batch of nblk blocks is ONE array of 16*nblk bytes per party; byte position
# p = r + 4c of the AES state s[r][c] (= input byte p of the block, FIPS-197 sec. 3.4) occupies
# buf[p*nblk : (p+1)*nblk].  SubBytes is then one S-box layer over the whole array, ShiftRows is a relabelling
# of row views (no data movement) and MixColumns is the small-public-matrix-times-rows kernel
# (ffgpu_recombine with w = k = 4, finfields.py:1126-1146 `C @ s`).
_MIX = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]          # circulant([2, 3, 1, 1]), np_aes.py:33


def _xpow(modulus: int, k: int) -> int:
    """x^k in GF(2)[x] / modulus (bit patterns)."""
    v, deg = 1, modulus.bit_length() - 1
    for _ in range(k):
        v <<= 1
        if v >> deg:
            v ^= modulus
    return v


def _row(ctx: FieldContext, buf: DevArray, p: int, nblk: int) -> DevArray:
    return DevArray(ctx, buf.t[p * nblk:(p + 1) * nblk], nblk)


def _cat(ctx: FieldContext, rows: Sequence[DevArray]) -> DevArray:
    import torch
    t = torch.cat([r.t for r in rows])
    return DevArray(ctx, t, t.shape[0])


def aes128_key_expansion(ctx: FieldContext, field, key: Shares, nblk: int, rbits_fn, t: int, A, B, rng=None):
    """key_expansion for Nk = 4 (np_aes.py:55-72).  key: 16*nblk bytes per party (one key per block,
    position-major).  Returns the 11 round keys, each a Shares of 16*nblk bytes.  rbits_fn(nbytes) supplies
    shares of 8*nbytes random bits (np_random_bits) for each S-box call."""
    m = len(key)
    w = [[[_row(ctx, key[i], r + 4 * c, nblk) for r in range(4)] for i in range(m)] for c in range(4)]   # w[c][party][r]
    for i in range(4, 44):
        prev = w[i - 1]
        if i % 4 == 0:
            sub = _sbox_any(ctx, field, [_cat(ctx, prev[pi]) for pi in range(m)], rbits_fn(4 * nblk), t, A, B, rng)
            tcol = [[_row(ctx, sub[pi], (r + 1) % 4, nblk) for r in range(4)] for pi in range(m)]        # RotWord
            rcon = _xpow(ctx.modulus, i // 4 - 1)                   # f256(1) << i//Nk - 1  (np_aes.py:67)
            for pi in range(m):
                tcol[pi][0] = ctx.add_scalar(tcol[pi][0], rcon)                                          # every party adds the public constant
        else:
            tcol = prev
        w.append([[ctx.add(tcol[pi][r], w[i - 4][pi][r]) for r in range(4)] for pi in range(m)])
    return [[_cat(ctx, [w[4 * j + c][pi][r] for c in range(4) for r in range(4)]) for pi in range(m)] for j in range(11)]


def aes128_encrypt(ctx: FieldContext, field, K, state: Shares, nblk: int, rbits_fn, t: int, A, B, rng=None) -> Shares:
    """encrypt (np_aes.py:75-86): AddRoundKey, 9 x (SubBytes, ShiftRows, MixColumns, AddRoundKey), final round
    without MixColumns.  state: 16*nblk bytes per party, position-major."""
    from .engine import DevMatrix
    m = len(state)
    lam = [v for row in _MIX for v in row]
    s = [ctx.add(state[pi], K[0][pi]) for pi in range(m)]
    for rnd in range(1, 11):
        s = _sbox_any(ctx, field, s, rbits_fn(16 * nblk), t, A, B, rng)
        nxt = []
        for pi in range(m):
            # ShiftRows: s'[r][c] = s[r][(c + r) % 4]  (np.roll(s[r], -r))
            shifted = lambda r, c: _row(ctx, s[pi], r + 4 * ((c + r) % 4), nblk)
            if rnd < 10:
                out = ctx.empty(16 * nblk)
                for c in range(4):
                    view = DevMatrix(ctx, out.t[4 * c * nblk:(4 * c + 4) * nblk].view(4, nblk), 4, nblk, nblk)
                    ctx.recombine([shifted(k, c) for k in range(4)], lam, w=4, out=view)                 # C @ column c
            else:
                out = _cat(ctx, [shifted(r, c) for c in range(4) for r in range(4)])
            nxt.append(ctx.add(out, K[rnd][pi]))
        s = nxt
    return s


# ---- inverse cipher (demos/np_aes.py:46-52, 89-99) ------------------------------------------------------
_MIX_INV = [[14, 11, 13, 9], [9, 14, 11, 13], [13, 9, 14, 11], [11, 13, 9, 14]]     # np.linalg.inv(C), np_aes.py:34


def _gf2_inverse(A: Sequence[Sequence[int]]) -> List[List[int]]:
    """Inverse of an 8x8 0/1 matrix over GF(2) (np.linalg.inv(A), np_aes.py:31): host scalars."""
    n = len(A)
    M = [list(r) + [int(i == j) for j in range(n)] for i, r in enumerate(A)]
    for c in range(n):
        p = next(r for r in range(c, n) if M[r][c])
        M[c], M[p] = M[p], M[c]
        for r in range(n):
            if r != c and M[r][c]:
                M[r] = [x ^ y for x, y in zip(M[r], M[c])]
    return [row[n:] for row in M]


def sbox1_layer(ctx: FieldContext, field, xs: Shares, rbits: Shares, t: int, A, B, rng=None) -> Shares:
    """AES inverse S-box on secret-shared bytes (np_aes.py:46-52): x = np_to_bits(x); x += B; x = A1 @ x;
    x = np_from_bits(x) ** 254.  The two local steps are one pass: A1 (x + B) = A1 x + A1 B."""
    A1 = _gf2_inverse(A)
    bias = [0] * 8
    for r in range(8):
        for c in range(8):
            bias[r] ^= A1[r][c] & B[c]
    bits = to_bits_gf256(ctx, field, xs, rbits, t)
    y = [ctx.bit_affine(b, A1, bias, from_bits=True) for b in bits]
    return pow254(ctx, field, y, t, rng)


def aes128_decrypt(ctx: FieldContext, field, K, state: Shares, nblk: int, rbits_fn, t: int, A, B, rng=None) -> Shares:
    """decrypt (np_aes.py:89-99): for i = 10..1: AddRoundKey(K[i]); InvMixColumns unless i = 10; InvShiftRows;
    InvSubBytes; finally AddRoundKey(K[0]).  Same position-major layout as aes128_encrypt."""
    from .engine import DevMatrix
    m = len(state)
    lam = [v for row in _MIX_INV for v in row]
    s = list(state)
    for rnd in range(10, 0, -1):
        nxt = []
        for pi in range(m):
            x = ctx.add(s[pi], K[rnd][pi])
            if rnd < 10:
                mixed = ctx.empty(16 * nblk)
                for c in range(4):
                    view = DevMatrix(ctx, mixed.t[4 * c * nblk:(4 * c + 4) * nblk].view(4, nblk), 4, nblk, nblk)
                    ctx.recombine([_row(ctx, x, k + 4 * c, nblk) for k in range(4)], lam, w=4, out=view)     # C1 @ column c
                x = mixed
            # InvShiftRows: s'[r][c] = s[r][(c - r) % 4]  (np.roll(s[r], r))
            nxt.append(_cat(ctx, [_row(ctx, x, r + 4 * ((c - r) % 4), nblk) for c in range(4) for r in range(4)]))
        s = sbox1_layer(ctx, field, nxt, rbits_fn(16 * nblk), t, A, B, rng=rng)
    return [ctx.add(s[pi], K[0][pi]) for pi in range(m)]
