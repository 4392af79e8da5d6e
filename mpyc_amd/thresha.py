"""Host-side mirror of mpyc.thresha (threshold secret sharing) on the MI355X engine.

Same names, argument meaning and return shapes as the reference (mpyc/thresha.py):

    random_split(field, s, t, m)          thresha.py:23-44    list path (lists of ints in/out)
    np_random_split(field, s, t, m)       thresha.py:47-64    array path -> (m, n) share matrix
    _recombination_vector(field, xs, x_r) thresha.py:67-85    host, cached (k <= m scalars)
    recombine(field, points, x_rs=0)      thresha.py:88-116   list path
    np_recombine(field, points, x_rs=0)   thresha.py:119-132  array path -> field.array

Randomness.  The reference draws every coefficient with secrets.randbelow (thresha.py:37,58-60).
Here the default is the on-device CSPRNG keyed from `secrets` (csrc/rng.hpp).  For bit-exact
parity with the reference set the module attribute `randbelow` to a callable (the same trick the
reference's tests use by patching secrets.randbelow): coefficients are then drawn through it on
the host in the reference's order and convention, and uploaded.

All share arithmetic runs on the GPU; the only host arithmetic is the k-entry Lagrange vector.
"""
from __future__ import annotations

import functools
import secrets
from hashlib import shake_128
from math import prod

import numpy as np

from .engine import DevArray, DevMatrix
from .finfields import FieldArray, HostView, _Rec, _context, _fops, _matrix_to_array, _src_dev

__all__ = ['random_split', 'recombine', 'np_random_split', 'np_recombine', '_recombination_vector',
           'pseudorandom_share', 'pseudorandom_share_zero', 'np_pseudorandom_share', 'np_pseudorandom_share_0', 'PRF']

randbelow = None      # parity hook: callable(order) -> int, else device CSPRNG
rng_rounds = 20       # ChaCha rounds of the device CSPRNG (20, 12 or 8)
_nonce = 0


device_rng_state = False  # True: share generation draws from ONE device-resident generator state per context
#                           (engine.RngState) instead of a fresh host key per call, so that code using this module
#                           can be captured in a HIP graph (engine.CapturedLaunches) and still get fresh randomness
#                           on every replay
lazy_recombine = True     # np_recombine defers single-target recombinations of <= 7 rows (see FieldArray._source)


def _ctx_state(ctx):
    st = getattr(ctx, '_thresha_rng_state', None)
    if st is None:
        st = ctx.rng_state(rounds=rng_rounds)            # key from the host CSPRNG, once
        ctx._thresha_rng_state = st
    return st


def _next_nonce():
    global _nonce
    _nonce += 1
    return _nonce


def _as_field_array(field, s) -> FieldArray:
    if isinstance(s, HostView):
        s = s._fa
    if isinstance(s, FieldArray):
        return s
    vals = [v if isinstance(v, (int, np.integer)) else int(getattr(v, 'value', v)) for v in s] \
        if not isinstance(s, np.ndarray) else s
    return field.array(vals)


def _split_device(field, S: FieldArray, t, m, np_convention: bool) -> DevMatrix:
    ctx = S.ctx
    n = S.size
    lazy = S._take_lazy_product()          # unmaterialised a*b: fuse the product into the split kernel
    if lazy is not None:
        a, b = lazy
        if (b is not None and randbelow is None and 1 <= t <= 3 and n
                and (isinstance(a, _Rec) or isinstance(b, _Rec))
                and all(len(src.rows) <= 7 for src in (a, b) if isinstance(src, _Rec))):
            # a chain of multiplications: the factors are still the sub-shares received in the previous gate
            # -> recombine in registers, multiply, re-share (ffgpu_gate_rng); nothing is written in between
            ra, la = (a.rows, a.lam) if isinstance(a, _Rec) else ([a], [1])
            rb, lb = (None, None) if b is a else ((b.rows, b.lam) if isinstance(b, _Rec) else ([b], [1]))
            if device_rng_state:
                return ctx.gate(ra, la, rb, lb, t, m, state=_ctx_state(ctx))
            return ctx.gate(ra, la, rb, lb, t, m, key=secrets.token_bytes(32), nonce=_next_nonce(), rounds=rng_rounds)
        dev = _src_dev(a)
        mul_by = None if b is None else (dev if b is a else _src_dev(b))
    else:
        dev, mul_by = (S.device_array if S.ndim == 1 else S.reshape(-1).device_array), None
    if t == 0 or n == 0:
        return ctx.split(dev, None, 0, m, mul_by=mul_by)
    if randbelow is None:
        if device_rng_state:
            return ctx.split_rng(dev, t, m, mul_by=mul_by, state=_ctx_state(ctx))
        return ctx.split_rng(dev, t, m, key=secrets.token_bytes(32), nonce=_next_nonce(), rounds=rng_rounds,
                             mul_by=mul_by)
    order = field.order
    draws = [randbelow(order) for _ in range(t * n)]
    if np_convention:
        rows = [draws[j * n:(j + 1) * n] for j in range(t)]                    # C[j][h] = d[j*n+h], thresha.py:60
    else:
        # list path: secret h uses d[h*t : (h+1)*t], c[0] on X^t ... c[t-1] on X (thresha.py:37-43)
        rows = [[draws[h * t + (t - 1 - j)] for h in range(n)] for j in range(t)]
    from .engine import ints_to_np
    eb = ctx.elem_bytes
    C = ctx.empty_matrix(t, n)
    for j in range(t):
        C.row(j).t.copy_(ctx.from_numpy(ints_to_np(rows[j], eb)).t)
    return ctx.split(dev, C, t, m, mul_by=mul_by)


class ShareMatrix:
    """Result of np_random_split: behaves like the reference's (m, n) ndarray for what the
    runtime does with it (iterate rows, index rows, len, .shape, pickle a row's .value), but the
    rows stay on the GPU until someone asks for Python ints."""

    def __init__(self, field, mtx: DevMatrix):
        self.field, self._mtx = field, mtx
        self.shape = (mtx.rows, mtx.n)

    def __len__(self):
        return self._mtx.rows

    def __getitem__(self, i) -> FieldArray:
        if isinstance(i, tuple):
            return self[i[0]][i[1:]] if len(i) > 1 else self[i[0]]
        if i < 0:
            i += self._mtx.rows
        if not 0 <= i < self._mtx.rows:
            raise IndexError(i)
        return self.field.array._wrap(self._mtx.row(i), (self._mtx.n,))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def value(self) -> np.ndarray:
        """The reference's return value: plain object ndarray (m, n) of canonical ints."""
        out = np.empty(self.shape, dtype=object)
        for i in range(self.shape[0]):
            out[i] = self[i]._host_value()
        return out

    def __array__(self, dtype=None, copy=None):
        return self.value


def np_random_split(field, s, t, m) -> ShareMatrix:
    """Split each secret in s into m random Shamir shares of degree t (0 <= t < m); one row per
    party (thresha.py:47-64)."""
    if not 0 <= t < m:
        raise ValueError('need 0 <= t < m')
    if _ipcwire.ENABLED:
        _ipcwire.ensure_runtime_hooks()
    S = _as_field_array(field, s)
    return ShareMatrix(field, _split_device(field, S, t, m, np_convention=True))


def random_split(field, s, t, m):
    """List path (thresha.py:23-44): s is a list of ints or field elements; returns m lists of ints."""
    if not 0 <= t < m:
        raise ValueError('need 0 <= t < m')
    if len(s) == 0:
        return [[] for _ in range(m)]
    S = _as_field_array(field, list(s))
    mtx = _split_device(field, S, t, m, np_convention=False)
    rows = mtx.to_ints()
    ops = _fops(field)
    if ops.binary:
        rows = [[ops.box(v) for v in r] for r in rows]
    return rows


@functools.lru_cache(maxsize=None)
def _recombination_vector(field, xs, x_r):
    """Lagrange coefficients for interpolation points xs evaluated at x_r, in the order of xs
    (thresha.py:67-85).  Host scalars; cached per (field, xs, x_r) like the reference."""
    ops = _fops(field)
    xs = [ops.reduce_int(int(x)) for x in xs]
    x_r = ops.reduce_int(int(x_r))
    vector = []
    for i, x_i in enumerate(xs):
        num, den = 1, 1
        for j, x_j in enumerate(xs):
            if i != j:
                num = ops.mul(num, ops.sub(x_r, x_j))
                den = ops.mul(den, ops.sub(x_i, x_j))
        vector.append(ops.mul(num, ops.inv(den)))
    return vector


def _rows_on_device(field, shares):
    rows = []
    for sh in shares:
        if isinstance(sh, HostView):
            sh = sh._fa
        if isinstance(sh, FieldArray):
            rows.append((sh if sh.ndim == 1 else sh.reshape(-1)).device_array)
        elif isinstance(sh, DevArray):
            rows.append(sh)
        else:
            rows.append(field.array(sh).reshape(-1).device_array)               # reduces, like :128
    return rows


def np_recombine(field, points, x_rs=0):
    """Recombine shares given by points [(x_j, row_j)] into secrets at x-coordinate(s) x_rs
    (thresha.py:119-132).  Returns field.array of shape (n,), or (w, n) if x_rs is a list."""
    xs, shares = list(zip(*points))
    scalar = not isinstance(x_rs, list)
    xr = (x_rs,) if scalar else tuple(x_rs)
    lam = [v for x_r in xr for v in _recombination_vector(field, tuple(xs), x_r)]
    rows = _rows_on_device(field, shares)
    ctx = rows[0].ctx
    if scalar and lazy_recombine and 1 < len(rows) <= 7 and rows[0].n > 1:
        # deferred: a product + share generation that follows recombines in registers; any other use
        # performs the recombination then (FieldArray._dev)
        return field.array._wrap_lazy_rec(rows, lam, (rows[0].n,))
    out = ctx.recombine(rows, lam, w=len(xr))
    if scalar:
        return field.array._wrap(out, (rows[0].n,))
    if len(xr) == 1:
        return field.array._wrap(out, (1, rows[0].n))
    return _matrix_to_array(field.array, out)


def recombine(field, points, x_rs=0):
    """List path (thresha.py:88-116).  The reference returns UNREDUCED integer sums for raw-int
    shares (:109) and field elements for field-element shares (:110-113); callers reduce the
    former (runtime.py:588,682).  This mirror returns canonical values in both cases: ints for
    int shares (congruent to the reference's sums modulo the field modulus) and field elements
    for field-element shares."""
    xs, shares = list(zip(*points))
    if len(shares[0]) == 0:
        return [] if not isinstance(x_rs, list) else [[] for _ in x_rs]
    T_is_field = isinstance(shares[0][0], field)
    rows = [[v if isinstance(v, (int, np.integer)) else int(getattr(v, 'value', v)) for v in sh] for sh in shares]
    out = np_recombine(field, list(zip(xs, rows)), x_rs)
    vals = out.unsigned_().tolist()

    def conv(row):
        if T_is_field:
            return [field(v) for v in row]
        ops = _fops(field)
        if ops.binary:
            return [ops.box(v) for v in row]
        return row
    return conv(vals) if not isinstance(x_rs, list) else [conv(r) for r in vals]


# --------------------------------------------------------------------------------------------
# Pseudorandom secret sharing (thresha.py:135-266).  The SHAKE128 XOF is sequential per key, so it
# runs on the host (hashlib); its raw output is uploaded and everything per element -- wide
# reduction of each l-byte draw, multiplication by f_S(i) (and the powers of i+1 for zero
# sharings), summation over the subsets -- is one kernel (ffgpu_prss_combine).
#
# Production mode (opt-in: prss_prf = 'chacha', or MPYC_AMD_PRSS_PRF=chacha in the environment of EVERY party): the subset
# keys' draws come from ChaCha streams expanded by the lanes that consume them (ffgpu_prss_chacha) -- same subsets, same
# f_S(i) weights, same l-byte-then-`% bound` sampling rule (thresha.py:234-266), another PRF: the values differ from the
# reference's, the distribution and the sharing structure do not.  SHAKE128 stays the default and the parity mode.
# --------------------------------------------------------------------------------------------
import os as _os

prss_prf = _os.environ.get('MPYC_AMD_PRSS_PRF', 'shake')         # 'shake' (reference PRF, bit-exact) | 'chacha' (device PRF)
prss_rounds = int(_os.environ.get('MPYC_AMD_PRSS_ROUNDS', '20'))   # ChaCha rounds of the device PRF: 20 or 12 (8: see below)
# ChaCha8 has no published attack but a thin margin: it is for measurements, admitted only on request
prss_allow_chacha8 = _os.environ.get('MPYC_AMD_PRSS_ALLOW_CHACHA8', '0') == '1'
PRSS_MAX_DRAW_BYTES = 64        # ffgpu_prss_chacha: bytes per draw l = byte length of the bound (+ key length against bias)


def prss_mode():
    """The validated (prf, rounds) pair of this party.  EVERY party of a computation must run the same pair: the shares
    of one subset key are only consistent when all its holders expand it with the same PRF (install() makes the parties
    confirm it at start-up, `prss_mode_tag`)."""
    if prss_prf not in ('shake', 'chacha'):
        raise ValueError(f"prss_prf (MPYC_AMD_PRSS_PRF) must be 'shake' or 'chacha', not {prss_prf!r}")
    if prss_rounds not in (20, 12, 8):
        raise ValueError(f'prss_rounds (MPYC_AMD_PRSS_ROUNDS) must be 20, 12 or 8, not {prss_rounds!r}')
    if prss_prf == 'chacha' and prss_rounds == 8 and not prss_allow_chacha8:
        raise ValueError('ChaCha8 as the PRSS PRF needs prss_allow_chacha8 = True (MPYC_AMD_PRSS_ALLOW_CHACHA8=1): '
                         'reduced-round variant, for measurements')
    return prss_prf, prss_rounds


def prss_mode_tag() -> str:
    """What the parties compare at start-up: 'shake' or 'chacha<rounds>/<domain tag version>'."""
    prf, rounds = prss_mode()
    return 'shake' if prf == 'shake' else f'chacha{rounds}/v1'


prss_mode()                      # a bad environment value fails at import, not at the first PRSS call
PRSS_CHACHA_DOMAIN = b'mpyc_amd prss chacha v1\0'


def prss_chacha_stream_key(key: bytes, s: bytes) -> bytes:
    """32-byte ChaCha key + 8-byte nonce of the stream that replaces shake_128(key + s) (thresha.py:255) in production
    mode: a KDF call per (PRF key, common input), microseconds on the host; the expansion runs on the device."""
    key = bytes(key)
    return shake_128(PRSS_CHACHA_DOMAIN + len(key).to_bytes(2, 'little') + key + bytes(s)).digest(40)


class PRF:
    """A pseudorandom function determined by a key and a public bound (thresha.py:220-266)."""

    def __init__(self, key, bound):
        self.key = key
        self.max = bound
        self.byte_length = ((bound - 1).bit_length() + 7) // 8
        if bound & (bound - 1):                       # not a power of 2: extra bytes against bias (:235-236)
            self.byte_length += len(self.key)

    def raw(self, s, n):
        """The n*l raw XOF bytes the reference chops into draws (thresha.py:255)."""
        return shake_128(self.key + s).digest(n * self.byte_length) if n and self.byte_length else b''

    def __call__(self, s, n=None):
        if isinstance(n, tuple):
            shape, n = n, prod(n)
        else:
            shape = None
        n_ = 1 if n is None else n
        l, bound = self.byte_length, self.max
        if n_ == 0:
            x = []
        elif not l:
            x = [0] * n_
        else:
            dk = self.raw(s, n_)
            x = [int.from_bytes(dk[i:i + l], 'little') % bound for i in range(0, n_ * l, l)]
        if shape is not None:
            return np.fromiter(x, object, count=n_).reshape(shape)
        return x[0] if n is None else x


# PRF classes whose draws ARE shake_128(key + s) chopped into byte_length-sized chunks: their streams may be expanded by
# libffgpu's host threads instead of by calling the object.  install() adds the reference's own class (the runtime makes
# its PRFs with mpyc.thresha.PRF); anything else -- a subclass, a user's PRF -- is called as it is.
SHAKE_PRF_TYPES = {PRF}


def register_shake_prf(cls) -> bool:
    """Admit a foreign PRF class (the installed mpyc's thresha.PRF) to the engine's own XOF expansion -- only after a
    known-answer check that ITS draws are shake_128(key + s) chopped by the same byte_length rule (thresha.py:220-266):
    another mpyc version with a different PRF keeps being called as the object it is."""
    if cls in SHAKE_PRF_TYPES:
        return True
    key, s = bytes(range(16)), b'\x07\x00\x01'
    try:
        for bound in (2**61 - 1, 1 << 64, 256, 2**127 - 1, 3):
            theirs, ours = cls(key, bound), PRF(key, bound)
            if getattr(theirs, 'byte_length', None) != ours.byte_length or getattr(theirs, 'max', None) != bound or \
                    bytes(theirs.key) != key or list(theirs(s, 5)) != ours(s, 5) or theirs(s) != ours(s):
                return False
    except Exception:          # noqa: BLE001 -- a class with another constructor or call signature is simply not admitted
        return False
    SHAKE_PRF_TYPES.add(cls)
    return True


@functools.lru_cache(maxsize=None)
def _f_S_i(field, m, i, S):
    """f_S(i+1) for the polynomial with f_S(0) = 1 and f_S(j+1) = 0 for all parties j outside S
    (thresha.py:135-141), as a canonical scalar."""
    ops = _fops(field)
    xs = (0,) + tuple(x + 1 for x in range(m) if x not in S)
    return _recombination_vector(field, xs, i + 1)[0]       # only the point at x = 0 carries a 1


from . import ipcwire as _ipcwire  # noqa: E402

PRSS_STREAM_MIN = 1 << 30        # XOF bytes of a call (all subset keys together) above which it is squeezed, uploaded and
#                                  combined in slices (bounded pinned memory, host and device overlap).  Below it the keys are
#                                  expanded in one piece each, in parallel, by libcrypto when it can be loaded -- its sponge
#                                  squeezes 0.9-1.2 GB/s per stream against 0.75 of the library's resumable one.


def _prss_device(field, m, i, prfs, uci, n, zero: bool, np_convention: bool):
    mode, rounds = prss_mode()
    ops = _fops(field)
    ctx = _context(field)
    items = list(prfs.items())
    first = items[0][1]
    bound, l = first.max, first.byte_length
    if any(prf.max != bound for _, prf in items):
        raise ValueError('all PRFs must share one bound')
    if bound & (bound - 1) == 0:
        mask_bits = bound.bit_length() - 1
    elif bound == ops.order:
        mask_bits = 0
    else:
        raise NotImplementedError('PRF bound must be the field order or a power of two (runtime.py:4062-4076)')
    out = ctx.empty(n)
    if n == 0:
        return out
    if l == 0 or (mask_bits == 0 and bound & (bound - 1) == 0):   # bound == 1: all draws are 0
        return ctx.mul_scalar(ctx.from_ints([0] * n), 0)
    d = (m - len(items[0][0])) if zero else 1
    if d == 0:                                   # t = 0: the zero sharing has no random part (thresha.py:208-216)
        out.t.zero_()
        return out
    i1 = ops.reduce_int(i + 1)
    weights = []
    for S, prf in items:
        f = _f_S_i(field, m, i, S)
        for j in range(d):
            if not zero:
                weights.append(f)
            else:
                # np path: draw j multiplies (i+1)^(j+1) (thresha.py:209); list path: Horner, draw j
                # multiplies (i+1)^(d-j) (thresha.py:193-195)
                power = j + 1 if np_convention else d - j
                w = f
                for _ in range(power):
                    w = ops.mul(w, i1)
                weights.append(w)
    if mode == 'chacha':
        # production mode: one ChaCha stream per subset key, expanded on the device (32 streams / 64 weights per launch).
        # Only PRF objects known to BE "shake_128(key + s) chopped into l-byte draws" are replaced (the admission rule of the
        # XOF path below); anything else has a __call__ of its own that a silent swap would change -- and a silent fall back
        # to SHAKE would disagree with the other parties.
        foreign = [type(prf).__name__ for _, prf in items if type(prf) not in SHAKE_PRF_TYPES]
        if foreign:
            raise TypeError(f"prss_prf = 'chacha' replaces mpyc's SHAKE128 PRF objects only; got {sorted(set(foreign))} "
                            '(register_shake_prf admits a class after a known-answer check)')
        if d > 64:
            raise NotImplementedError('zero sharing of degree > 64')
        if l > PRSS_MAX_DRAW_BYTES:
            raise NotImplementedError(f'device PRF: {l} bytes per draw (byte length of the bound + key length) exceeds the '
                                      f'kernel limit of {PRSS_MAX_DRAW_BYTES} (ffgpu_prss_chacha)')
        per = max(1, min(32, 64 // d))
        for k0 in range(0, len(items), per):
            chunk = items[k0:k0 + per]
            ctx.prss_chacha([prss_chacha_stream_key(prf.key, uci) for _, prf in chunk], d, l, weights[k0 * d:(k0 + per) * d], n,
                            mask_bits=mask_bits, rounds=rounds, out=out, accumulate=k0 > 0)
        return out
    # kernel argument limits: 48 streams / 96 weights per launch.  The XOF streams of a chunk (one per subset
    # key; each is inherently sequential) are expanded in parallel on host threads into pinned buffers.
    per = max(1, min(48, 96 // d))
    first_launch = True
    for k0 in range(0, len(items), per):
        chunk = items[k0:k0 + per]
        if all(type(prf) in SHAKE_PRF_TYPES for _, prf in chunk):
            if len(chunk) * n * d * l > PRSS_STREAM_MIN and hasattr(ctx, 'prss_streamed'):
                # long streams: squeeze, upload and combine slice by slice (bounded pinned memory, host and device overlap)
                ctx.prss_streamed([bytes(prf.key) + bytes(uci) for _, prf in chunk], d, l, weights[k0 * d:(k0 + per) * d], n,
                                  mask_bits, out, not first_launch)
                first_launch = False
                continue
            # (bytes(...): the runtime stores the PRSS keys it RECEIVED as bytearray slices, runtime.py:139)
            streams = ctx.shake128_streams([bytes(prf.key) + bytes(uci) for _, prf in chunk], n * d * l)
        else:                                   # foreign PRF objects (e.g. the reference's own class)
            streams = [prf.raw(uci, n * d) if hasattr(prf, 'raw') else shake_128(prf.key + uci).digest(n * d * l)
                       for _, prf in chunk]
        ctx.prss_combine(streams, d, l, weights[k0 * d:(k0 + per) * d], n, mask_bits=mask_bits,
                         out=out, accumulate=not first_launch)
        first_launch = False
    return out


def np_pseudorandom_share(field, m, i, prfs, uci, n):
    """Pseudorandom Shamir shares for party i of n random numbers (thresha.py:163-173)."""
    return field.array._wrap(_prss_device(field, m, i, prfs, uci, n, False, True), (n,))


def np_pseudorandom_share_0(field, m, i, prfs, uci, n):
    """Pseudorandom Shamir shares for party i of n sharings of 0 (thresha.py:201-217)."""
    return field.array._wrap(_prss_device(field, m, i, prfs, uci, n, True, True), (n,))


def pseudorandom_share(field, m, i, prfs, uci, n):
    """List version (thresha.py:144-160): list of field elements."""
    return [field(v) for v in _prss_device(field, m, i, prfs, uci, n, False, False).to_ints()]


def pseudorandom_share_zero(field, m, i, prfs, uci, n):
    """List version (thresha.py:176-198)."""
    return [field(v) for v in _prss_device(field, m, i, prfs, uci, n, True, False).to_ints()]
