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


def sbox_layer(ctx: FieldContext, field, xs: Shares, rbits: Shares, t: int, A: Sequence[Sequence[int]],
               B: Sequence[int], fused: bool = True, rng=None) -> Shares:
    """The AES S-box on secret-shared bytes, as demos/np_aes.py:37-43:
    x = np_to_bits(x**254); x = A @ x + B over GF(2) (on bit shares, local); x = np_from_bits(x)."""
    y = pow254(ctx, field, xs, t, rng)
    bits = to_bits_gf256(ctx, field, y, rbits, t)
    if fused:
        return [ctx.bit_affine(b, A, B, from_bits=True) for b in bits]           # both local steps in one pass
    bits = [ctx.group_matvec(b, A, B) for b in bits]
    return from_bits(ctx, bits)
