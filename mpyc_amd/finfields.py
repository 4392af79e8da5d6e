"""Host-side mirror of mpyc.finfields on the MI355X engine.

Same names and argument meaning as the reference (mpyc/finfields.py) for the part of the
module that sits on the hot path:

    GF(modulus)                      finfields.py:23-42    field type + `.array` type
    find_prime_root, find_irreducible  :311-344, :502-505 the default moduli MPyC picks
    PrimeFieldElement / BinaryFieldElement   :366-, :528-  scalar elements (host, Python ints)
    FieldArray (= field.array)       :695-1368              arrays: device-resident limbs, every
                                                            element-wise op is a HIP kernel launch

Scalars stay on the host (there is nothing to parallelise in one field element); arrays never
compute on the host: an operation the engine cannot run raises NotImplementedError, it does not
fall back to NumPy object arrays.  `.value` materialises the reference's representation (object
ndarray of canonical Python ints, finfields.py:703-709) lazily and read-only, for the callers
that read it directly (runtime.py:561,643,...) or pickle it for the wire.
"""
from __future__ import annotations

import collections
import functools
import os
import random as _random
import sys
import weakref
from math import prod as _mprod
from typing import Optional

import numpy as np
import torch

try:
    from xxhash import xxh64 as _xxh64
except ImportError:                      # (no digest, no cache of public uploads: every use converts and uploads)
    _xxh64 = None

from .engine import np_to_objects, DevArray, DevMatrix, FieldContext, ints_to_np
from .gfpx import BinaryPolynomial, _clinvert, _clmod, _clmul


def _prod(shape) -> int:
    """number of elements of a shape (Python ints; () -> 1)"""
    return int(_mprod(int(s_) for s_ in shape))

__all__ = ['GF', 'find_prime_root', 'find_irreducible', 'FieldArray', 'PrimeFieldElement', 'BinaryFieldElement']


# --------------------------------------------------------------------------------------------
# number theory helpers (mpyc/gmpy.py stubs :122-290 do the same in pure Python)
# --------------------------------------------------------------------------------------------
_SMALL_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for q in _SMALL_PRIMES:
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    # deterministic for n < 3.3e24 with the first 12 primes; add random bases beyond that
    bases = list(_SMALL_PRIMES)
    if n.bit_length() > 81:
        rng = _random.Random(n)
        bases += [rng.randrange(2, n - 1) for _ in range(24)]
    for a in bases:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def prev_prime(n: int) -> int:
    n -= 1
    while n >= 2 and not is_prime(n):
        n -= 1
    return n


def next_prime(x):
    """smallest prime above x (gmpy2.next_prime / mpyc.gmpy stub)"""
    x = max(int(x) + 1, 2)
    if x > 2 and x % 2 == 0:
        x += 1
    while not is_prime(x):
        x += 1 if x == 2 else 2
    return x


def find_prime_root(l, blum=True, n=1):
    """finfields.py:311-344: prime of bit length l (largest [Blum] prime below 2^l for n <= 2)."""
    if l <= 2:
        if not blum:
            return 2, 1, 1
        return 3, 2, 2
    if n <= 2:
        p = prev_prime(1 << l)
        if blum:
            while p % 4 != 3:
                p = prev_prime(p)
        return p, n, (p - 1 if n == 2 else 1)
    # finfields.py:331-343: a Blum prime p = 1 (mod 2n) of about l bits and an n-th root of unity w (n rounded up to
    # a prime): host scalars, as in the reference
    if not blum:
        raise AssertionError('roots of unity of order n > 2 need a Blum prime (finfields.py:333)')
    if not is_prime(n):
        n = next_prime(n)
    p = 1 + 2 * n * (3 + 2 * ((1 << l - 3) // n))
    while not is_prime(p):
        p += 4 * n
    a = 2
    while (w := pow(a, (p - 1) // n, p)) == 1:
        a += 1
    return p, n, w


def find_irreducible(p, d):
    """finfields.py:502-505: lexicographically first irreducible polynomial of degree d."""
    if p != 2:
        raise NotImplementedError('only characteristic 2 extension fields are accelerated')
    return BinaryPolynomial.next_irreducible(p**d - 1)


# --------------------------------------------------------------------------------------------
# field elements (host scalars)
# --------------------------------------------------------------------------------------------
class FiniteFieldElement:
    """finfields.py:63-221.  Invariant: `value` is reduced w.r.t. the modulus."""

    __slots__ = 'value'
    modulus = None
    order = None
    characteristic = None
    ext_deg = None
    byte_length = None
    is_signed = None
    array = None
    _binary = False

    def __init__(self, value):
        self.value = self._reduce(value)

    # -- wire format: fixed-width little-endian (finfields.py:91-102) --
    @classmethod
    def to_bytes(cls, x):
        r = cls.byte_length
        return b''.join(int(v).to_bytes(r, 'little') for v in x)

    @classmethod
    def from_bytes(cls, data):
        r = cls.byte_length
        return [int.from_bytes(data[i:i + r], 'little') for i in range(0, len(data), r)]

    @classmethod
    def _coerce(cls, other):
        if isinstance(other, cls):
            return int(other.value)
        if isinstance(other, (int, np.integer)) and not isinstance(other, bool):
            return cls._reduce_int(int(other))
        if isinstance(other, BinaryPolynomial) and cls._binary:
            return cls._reduce_int(int(other))
        return None

    def __add__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._add(int(self.value), o))

    __radd__ = __add__

    def __sub__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._sub(int(self.value), o))

    def __rsub__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._sub(o, int(self.value)))

    def __mul__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._mul(int(self.value), o))

    __rmul__ = __mul__

    def __neg__(self):
        return type(self)(self._sub(0, int(self.value)))

    def __truediv__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._mul(int(self.value), self._inv(o)))

    def __rtruediv__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else type(self)(self._mul(o, self._inv(int(self.value))))

    def __pow__(self, e):
        if not isinstance(e, int):
            return NotImplemented
        base = int(self.value)
        if e < 0:
            base, e = self._inv(base), -e
        r = 1
        while e:
            if e & 1:
                r = self._mul(r, base)
            base = self._mul(base, base)
            e >>= 1
        return type(self)(r)

    def reciprocal(self):
        return type(self)(self._inv(int(self.value)))

    def __eq__(self, other):
        o = self._coerce(other)
        return NotImplemented if o is None else int(self.value) == o

    def __hash__(self):
        return hash((type(self).__name__, int(self.value)))

    def __bool__(self):
        return bool(int(self.value))

    def __repr__(self):
        return f'{int(self)}'


class PrimeFieldElement(FiniteFieldElement):
    """finfields.py:366-500."""

    __slots__ = ()

    @classmethod
    def _reduce_int(cls, v):
        return v % cls.modulus

    def _reduce(self, value):
        if isinstance(value, FiniteFieldElement):
            value = value.value
        if not isinstance(value, (int, np.integer)) or isinstance(value, bool):
            raise TypeError(f'int required, got {type(value).__name__}')      # finfields.py:379
        return int(value) % self.modulus

    @classmethod
    def _add(cls, a, b):
        return (a + b) % cls.modulus

    @classmethod
    def _sub(cls, a, b):
        return (a - b) % cls.modulus

    @classmethod
    def _mul(cls, a, b):
        return a * b % cls.modulus

    @classmethod
    def _inv(cls, a):
        if a % cls.modulus == 0:
            raise ZeroDivisionError('inverse does not exist')                 # gmpy.py:197-210
        return pow(a, -1, cls.modulus)

    def signed_(self):
        v = int(self.value)
        return v - self.modulus if v > self.modulus >> 1 else v              # finfields.py:1395-1398 rule

    def unsigned_(self):
        return int(self.value)

    def __int__(self):
        return self.signed_() if self.is_signed else self.unsigned_()


class BinaryFieldElement(FiniteFieldElement):
    """finfields.py:528-692 for characteristic 2; `value` is a BinaryPolynomial (gfpx.py:848)."""

    __slots__ = ()
    _binary = True

    @classmethod
    def _reduce_int(cls, v):
        return _clmod(abs(int(v)), int(cls.modulus))

    def _reduce(self, value):
        if isinstance(value, FiniteFieldElement):
            value = value.value
        if isinstance(value, float):
            raise TypeError('int or polynomial required')
        return BinaryPolynomial(_clmod(int(BinaryPolynomial(value)), int(self.modulus)))

    @classmethod
    def _add(cls, a, b):
        return a ^ b

    _sub = _add

    @classmethod
    def _mul(cls, a, b):
        return _clmod(_clmul(a, b), int(cls.modulus))

    @classmethod
    def _inv(cls, a):
        if a == 0:
            raise ZeroDivisionError('inverse does not exist')
        return _clinvert(a, int(cls.modulus))

    def __int__(self):
        return int(self.value)


# --------------------------------------------------------------------------------------------
# field factories
# --------------------------------------------------------------------------------------------
def GF(modulus):
    """Create a finite field for a prime modulus (int) or an irreducible binary polynomial
    (mpyc_amd.gfpx.BinaryPolynomial); also creates the GPU-backed array type (finfields.py:23-42)."""
    if isinstance(modulus, BinaryPolynomial):
        return _xGF(int(modulus))
    if isinstance(modulus, tuple):
        modulus = modulus[0]
    if isinstance(modulus, (int, np.integer)) and not isinstance(modulus, bool):
        return _pGF(int(modulus))
    raise TypeError('modulus must be a prime int or a BinaryPolynomial')


def _make_array(field):
    arr = type(f'Array{field.__name__}', (_MirrorFieldArray,), {'__slots__': ()})     # finfields.py:45-60
    arr.field = field
    field.array = arr
    return arr


def device_supports_prime(p: int) -> bool:
    """Primes the engine has a storage format / reduction for: every prime of up to 192 bits (three 64-bit limbs above
    128 bits: 2^k - c with c < 2^31 by folding, any other odd prime by Montgomery products; include/ffgpu.h
    ffgpu_ctx_create)."""
    return p.bit_length() <= 192


@functools.lru_cache(maxsize=None)
def _pGF(p):
    if not is_prime(p):
        raise ValueError('modulus is not a prime')                            # finfields.py:351
    if not device_supports_prime(p):
        raise NotImplementedError('primes above 192 bits are not supported by the device path')
    F = type(f'GF({p})', (PrimeFieldElement,), {'__slots__': ()})
    F.modulus, F.order, F.characteristic, F.ext_deg = p, p, p, 1
    F.byte_length = (p.bit_length() + 7) >> 3
    F.is_signed = True
    F.nth, F.root = (1, 1) if p == 2 else (2, p - 1)
    _make_array(F)
    return F


@functools.lru_cache(maxsize=None)
def _xGF(mod):
    if not BinaryPolynomial.is_irreducible(mod):
        raise ValueError('modulus is not irreducible')                        # finfields.py:514
    d = mod.bit_length() - 1
    if d > 128:
        raise NotImplementedError('GF(2^n) beyond n = 128 is not supported by the device path')
    F = type(f'GF(2^{d})', (BinaryFieldElement,), {'__slots__': ()})
    F.modulus, F.order, F.characteristic, F.ext_deg = BinaryPolynomial(mod), 2**d, 2, d
    F.byte_length = ((2**d).bit_length() + 7) >> 3                            # NB 2 for GF(2^8): finfields.py:524
    _make_array(F)
    return F



# --------------------------------------------------------------------------------------------
# scalar-op adapter: lets FieldArray / thresha work with this module's field classes AND with the
# reference's own (mpyc.finfields) classes when the array type is substituted into mpyc
# (INTEGRATION.md section 2): only `modulus` / `order` are read from the field class.
# --------------------------------------------------------------------------------------------
class _FieldOps:
    __slots__ = ('binary', 'modulus', 'order', 'poly_type')

    def __init__(self, field):
        mod = field.modulus
        self.binary = not isinstance(mod, (int, np.integer))
        if self.binary and getattr(type(mod), 'p', 2) != 2:
            raise NotImplementedError('extension fields of odd characteristic are not accelerated')
        self.modulus = int(mod)
        self.order = int(field.order)
        self.poly_type = type(mod) if self.binary else None

    def reduce_int(self, v):
        return _clmod(abs(int(v)), self.modulus) if self.binary else int(v) % self.modulus

    def add(self, a, b):
        return a ^ b if self.binary else (a + b) % self.modulus

    def sub(self, a, b):
        return a ^ b if self.binary else (a - b) % self.modulus

    def mul(self, a, b):
        return _clmod(_clmul(a, b), self.modulus) if self.binary else a * b % self.modulus

    def inv(self, a):
        if self.binary:
            if a == 0:
                raise ZeroDivisionError('inverse does not exist')
            return _clinvert(a, self.modulus)
        if a % self.modulus == 0:
            raise ZeroDivisionError('inverse does not exist')
        return pow(a, -1, self.modulus)

    def box(self, v):
        """canonical int -> what the reference stores in `.value` (int, or a polynomial object)"""
        return self.poly_type(v) if self.binary else v


_fops_cache = {}


def _fops(field) -> _FieldOps:
    ops = _fops_cache.get(field)
    if ops is None:
        ops = _fops_cache[field] = _FieldOps(field)
        _field_registry[_field_key(field, ops)] = field
    return ops


def _scalar_value(x):
    """field element (this module's or the reference's) -> canonical int, else None"""
    if isinstance(x, (np.ndarray, FieldArray)):
        return None
    v = getattr(x, 'value', None)
    if v is None:
        return None
    try:
        return int(v)
    except (TypeError, ValueError):
        return None


def _ctx_add(ctx, a, b, out=None):
    return ctx.add(a, b, out)


def _ctx_sub(ctx, a, b, out=None):
    return ctx.sub(a, b, out)


def _ctx_mul(ctx, a, b, out=None):
    return ctx.mul(a, b, out)


def _ctx_add_scalar(ctx, a, s, out=None):
    return ctx.add_scalar(a, s, out)


def _ctx_mul_scalar(ctx, a, s, out=None):
    return ctx.mul_scalar(a, s, out)


_ctx_cache = {}
_pending_products = []     # weakrefs to FieldArrays holding an unmaterialised product
lazy_products = True     # defer `a * b` so that a following np_random_split fuses it (see FieldArray._dev)


def _context(field, device: Optional[int] = None) -> FieldContext:
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    key = (field, device)
    ctx = _ctx_cache.get(key)
    if ctx is None:
        ops = _fops(field)
        ctx = FieldContext(ops.modulus, binary=ops.binary, device=device)
        _ctx_cache[key] = ctx
    return ctx


# --------------------------------------------------------------------------------------------
# arrays
# --------------------------------------------------------------------------------------------
class _Rec:
    """A deferred Lagrange recombination sum_j lam[j] * rows[j] (thresha.np_recombine with a single target):
    kept symbolic so that a following product + share generation can recombine in registers
    (ffgpu_gate_rng).  Materialised at most once."""

    __slots__ = ('rows', 'lam', '_val')

    def __init__(self, rows, lam):
        self.rows, self.lam, self._val = list(rows), [int(v) for v in lam], None

    @property
    def ctx(self):
        return self.rows[0].ctx

    def materialize(self) -> DevArray:
        if self._val is None:
            self._val = self.ctx.recombine(self.rows, self.lam)
        return self._val

    def reads(self, sid: int) -> bool:
        """does this deferred recombination read (or has it materialised into) the storage `sid`?"""
        return any(_storage_id(r) == sid for r in self.rows) or (self._val is not None and _storage_id(self._val) == sid)


def _storage_id(dev: DevArray) -> int:
    """Identity of the allocation behind a device array: views (slices, reshapes, rows of a share matrix) share it,
    so hazards between a deferred product and an in-place update are detected for every alias, not only for
    equal base pointers."""
    return dev.t.untyped_storage().data_ptr()


def _src_dev(src) -> DevArray:
    return src.materialize() if isinstance(src, _Rec) else src


_POISON = np.array(None, dtype=object)
_HV_OWN = frozenset(('_fa', '_fa_value', '_fa_make', '_real', '_is_lazy', '_exact', '_thunk', '_true', '_derive', '_dev_ok', '_cmp_on_device', '_operand', '_binop', '_shift_left',
                     '_bits_and', '_outer_src', '_outer_T', '_shift_src', '_lift', 'T', 'transpose', 'shape', 'ndim', 'size', 'dtype', 'reshape', '__class__', '__dict__', '__reduce__',
                     '__reduce_ex__', '__setitem__', '__array_function__', '__array_ufunc__', '__array_finalize__',
                     '__array_priority__', '__copy__', '__deepcopy__', '__len__'))


# A large public integer array that meets share arrays several times in a row -- np_sgn adds, multiplies and subtracts the
# bit matrix `c_bits` (3.2e6 bits for 10^5 32-bit values) with them, runtime.py:3663, 3671 -- was converted to limbs and uploaded
# every time (~4 ms each).  The last few uploads are kept, keyed by field, shape, dtype and a 64-bit digest of the CONTENT (the
# array is mutable: identity would not do); a hit hands back the device array that was made from the same values.
_PUBLIC_UPLOADS = collections.OrderedDict()
_PUBLIC_UPLOAD_MIN, _PUBLIC_UPLOAD_MAX, _PUBLIC_UPLOAD_KEEP = 1 << 16, 1 << 25, 4      # (at most 4 x 2^25 elements stay pinned)


def _uploaded_public(cls, arr: np.ndarray):
    if not _PUBLIC_UPLOAD_MIN <= arr.size <= _PUBLIC_UPLOAD_MAX or arr.dtype.kind not in 'iub' or _xxh64 is None:
        return cls(arr)
    flat = np.ascontiguousarray(arr)
    key = (cls, arr.shape, arr.dtype.str, _xxh64(flat.view(np.uint8).reshape(-1)).intdigest())
    kept = _PUBLIC_UPLOADS.get(key)
    if kept is not None:
        _PUBLIC_UPLOADS.move_to_end(key)
    else:
        kept = cls(flat)
        _PUBLIC_UPLOADS[key] = kept
        while len(_PUBLIC_UPLOADS) > _PUBLIC_UPLOAD_KEEP:
            _PUBLIC_UPLOADS.popitem(last=False)
    return kept.copy()                # (a device-side clone: the kept array itself never leaves this function -- a caller may
    #                                    update what it gets in place)


_UNIFORM_MASK_MIN = 1 << 16


class _UniformMask(np.ndarray):
    """A read-only boolean array whose elements are all equal: a zero-stride view of one np.bool_, so it costs nothing to
    make however large the shape.  It IS an ndarray (indexing with it, arithmetic on it etc. work as for any boolean array);
    what the runtime does with such masks -- np.count_nonzero, .all(), .any(), ~mask -- is answered without touching 10^7
    elements.  Only instances that still are the zero-stride view answer that way (a ufunc result of this type is ordinary)."""

    def __new__(cls, shape, value):
        return np.broadcast_to(np.bool_(bool(value)), tuple(shape)).view(cls)

    def _uniform(self):
        return self.dtype == np.bool_ and self.size > 0 and not any(self.strides) and self.base is not None

    def _value(self) -> bool:
        return bool(self.flat[0])

    def __invert__(self):
        if self._uniform():
            return _UniformMask(self.shape, not self._value())
        return np.invert(np.asarray(self))

    def all(self, *args, **kwargs):
        if self._uniform() and not args and not kwargs:
            return np.bool_(self._value())
        return np.asarray(self).all(*args, **kwargs)

    def any(self, *args, **kwargs):
        if self._uniform() and not args and not kwargs:
            return np.bool_(self._value())
        return np.asarray(self).any(*args, **kwargs)

    def sum(self, *args, **kwargs):
        if self._uniform() and not args and not kwargs:
            return np.int64(self.size if self._value() else 0)
        return np.asarray(self).sum(*args, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        if len(args) == 1 and not kwargs and args[0] is self and self._uniform():
            name = func.__name__
            if name == 'count_nonzero':
                return self.size if self._value() else 0
            if name in ('all', 'any'):
                return np.bool_(self._value())
        args = tuple(np.asarray(a) if isinstance(a, _UniformMask) else a for a in args)
        return func(*args, **kwargs)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if ufunc is np.invert and method == '__call__' and len(inputs) == 1 and not kwargs and self._uniform():
            return _UniformMask(self.shape, not self._value())
        inputs = tuple(np.asarray(a) if isinstance(a, _UniformMask) else a for a in inputs)
        if 'out' in kwargs:
            kwargs['out'] = tuple(np.asarray(a) if isinstance(a, _UniformMask) else a for a in kwargs['out'])
        return getattr(ufunc, method)(*inputs, **kwargs)


class HostView(np.ndarray):
    """What `FieldArray.value` returns: an np.ndarray (so `isinstance(v, np.ndarray)`, sectypes.py:1366,
    holds) that stands for the reference's object ndarray but keeps the data on the device for what the
    runtime does on its hot path --

        x = a.value; shape = x.shape; x = x.reshape(-1)          runtime.py:561-563, 643-645
        thresha.np_random_split(field, x, t, m)                  runtime.py:495, 662
        pickle.dumps(x) / points.append((pid + 1, x))            runtime.py:571-585

    -- and resolves to the real object ndarray of canonical ints (materialised once per array, read-only:
    FieldArray._host_value) for everything else: methods and attributes (__getattribute__), operators,
    indexing, NumPy functions and ufuncs (__array_function__ / __array_ufunc__).  Its OWN buffer is a
    zero-stride broadcast of None (no allocation): C-level code that bypasses every Python hook
    (np.asarray(v), ndarray.__setitem__ from it) sees None entries, which fail loudly in the next integer
    operation instead of yielding wrong numbers."""

    __array_priority__ = 0
    # defaults for views created by .view(HostView) without __new__ (the symbolic shift-outer objects)
    _fa_value = _fa_make = _thunk = _true = _outer_src = _shift_src = None
    _exact, _outer_T, _is_lazy = True, False, False

    lazy_min = None     # None: `.value` is lazy only when read by a registered reader (below) and
    #                     carries the REAL values in its buffer everywhere else, so that it behaves like the
    #                     reference's ndarray in every context, including C-level ones that no Python hook can
    #                     intercept (`b[mask] = x.value`, runtime.py:1285).  An int: also lazy from that size on.
    # Readers for which `.value` may stay on the device: CODE OBJECTS, registered explicitly
    # (register_lazy_reader).  Under install() these are the reference runtime's three hot-path coroutines
    # (runtime.py:643 _reshare, :561 output, :494 _distribute), where `.value` is only reshaped, handed to thresha
    # and pickled; they are looked up on the imported mpyc.runtime module itself, so a user function that happens to
    # carry one of those names -- in whatever file -- never gets the lazy view, and a renamed upstream method simply
    # is not registered (the view is then materialised: slower, never wrong).
    lazy_codes = set()
    lazy_codes_prime = set()      # readers that compute with integer operators on `.value`: device-resident for PRIME fields
    lazy_ints_min = int(os.environ.get('MPYC_AMD_LAZY_INTS_MIN', '4096'))   # ... from this many elements on: below it a
    #                     protocol's dozens of tiny launches cost more than boxing a few thousand integers (measured on
    #                     np_lpsolver -i5: 136-bit field, arrays of ~100 elements)
    _runtime_scanned = False
    RUNTIME_LAZY_READERS = ('_reshare', 'output', '_distribute')
    # runtime.py:838-873, 4391-4472, 4475-4484, 4187-4273, 3622-3690, 3581-3620, 1798-1822: every use of `.value` in these coroutines is a Python-level
    # operator / NumPy function handled below (audited for v0.11.2); binary fields take their own branches there
    # (np.vectorize(int) over BinaryPolynomial objects: C-level), so for them `.value` stays materialised
    RUNTIME_LAZY_READERS_PRIME = ('np_trunc', 'np_to_bits', 'np_from_bits', 'np_random_bits', 'np_sgn', '_np_is_zero', 'np_lsb')
    # Readers that are lazy on SOME of their lines only: {code: line numbers}.  runtime.py:1251-1294 np_reciprocal reads
    # `.value` four times; the one every call executes is `np.count_nonzero(ar.value) < n` (a NumPy function: counted on the
    # device), the other three sit on the ar == 0 branch and feed `b[mask] = ... .value` -- a C-level item assignment no
    # Python hook sees, which needs the real integers.  The lines are found in the function's SOURCE by pattern at
    # registration; no source or no match: not registered (materialised: slower, never wrong).
    lazy_lines = {}
    RUNTIME_LAZY_LINES = {'np_reciprocal': r'np\.count_nonzero\(\s*\w+\.value\s*\)'}

    @classmethod
    def register_lazy_reader(cls, func, prime_only=False, line_pattern=None):
        """Declare that `func` (a function / coroutine function, possibly wrapped by decorators that set
        __wrapped__) only reshapes `.value`, passes it to thresha / the array constructor, or pickles it --
        or, with prime_only, computes on it with the integer operators HostView implements on the device --
        or, with line_pattern (a regex), does so on the source lines that match and nowhere else."""
        import inspect
        code = getattr(inspect.unwrap(func), '__code__', None)
        if code is None:
            raise TypeError('register_lazy_reader needs a Python function')
        if line_pattern is not None:
            import re
            try:
                lines, first = inspect.getsourcelines(inspect.unwrap(func))
            except (OSError, TypeError):
                return False
            hits = frozenset(first + k for k, ln in enumerate(lines) if re.search(line_pattern, ln))
            if not hits:
                return False
            cls.lazy_lines[code] = hits
            return True
        (cls.lazy_codes_prime if prime_only else cls.lazy_codes).add(code)
        return True

    @classmethod
    def _scan_runtime(cls):
        """Once mpyc.runtime has been imported (importing it here would parse sys.argv and start a runtime), register
        its hot-path readers.  Called from FieldArray.value until the module shows up."""
        rt = sys.modules.get('mpyc.runtime')
        if rt is None or not hasattr(rt, 'Runtime'):
            return
        cls._runtime_scanned = True
        for name in cls.RUNTIME_LAZY_READERS:
            fn = getattr(rt.Runtime, name, None)
            if fn is not None:
                cls.register_lazy_reader(fn)
        for name, pattern in cls.RUNTIME_LAZY_LINES.items():
            fn = getattr(rt.Runtime, name, None)
            if fn is not None:
                cls.register_lazy_reader(fn, line_pattern=pattern)
        if os.environ.get('MPYC_AMD_LAZY_INTS', '1') != '0':
            for name in cls.RUNTIME_LAZY_READERS_PRIME:
                fn = getattr(rt.Runtime, name, None)
                if fn is not None:
                    cls.register_lazy_reader(fn, prime_only=True)

    def __new__(cls, fa, lazy=None, exact=True, thunk=None, shape=None):
        """fa: the device array -- or, with `shape`, a callable that makes it on first use (a derived view whose residues
        may never be needed, e.g. `bits << shifts` that is only summed)."""
        make = None
        if shape is not None:
            make, fa = fa, None
        if lazy is None:
            lazy = cls.lazy_min is not None and fa.size >= cls.lazy_min
        if lazy:
            obj = np.broadcast_to(_POISON, fa._shape if make is None else tuple(shape)).view(cls)
        else:
            obj = fa._host_value().view(cls)
        obj._fa_value = fa
        obj._fa_make = make
        obj._is_lazy = bool(lazy)
        obj._exact = bool(exact)        # True: the integers ARE the canonical residues in _fa
        obj._thunk = thunk              # else: () -> the true object ndarray (same NumPy expression on the parents)
        obj._true = None
        obj._outer_src = None
        obj._outer_T = False
        obj._shift_src = None
        return obj

    def __array_finalize__(self, obj):
        pass

    @property
    def _fa(self):
        fa = object.__getattribute__(self, '_fa_value')
        if fa is None:
            fa = object.__getattribute__(self, '_fa_make')()
            self._fa_value, self._fa_make = fa, None
        return fa

    @_fa.setter
    def _fa(self, fa):
        self._fa_value, self._fa_make = fa, None

    # ---- integer arithmetic that stays on the device (the runtime's protocols: np_trunc, np_to_bits, np_from_bits,
    #      np_random_bits -- runtime.py:838-873, 4391-4484, 4187-4273) --------------------------------------------
    # Those protocols compute on `.value` with INTEGER operators (<<, +, -, *, **, sum) and hand the result back to
    # `field.array(...)`, which reduces it: every such operator is a ring homomorphism onto GF(p), so the device can
    # compute the RESIDUES and the reduction at the end is already done.  A derived view therefore carries
    #     _fa      the residues (a device array: what field.array(view) takes, no host round trip),
    #     _exact   whether the residues ARE the integers (true for `.value` itself, for `% p`, `& mask`, ...),
    #     _thunk   how to compute the true integers on the host from the parents (the same NumPy expression),
    # and anything that is not a homomorphism (`&`, `% 2^l`, `>>`, comparisons, printing, iteration, any NumPy function
    # not handled below) either runs on the device when _exact holds or falls back to the true integers through
    # _thunk: always the reference's result, never an approximation.  Prime fields only.
    def _real(self) -> np.ndarray:
        thunk = object.__getattribute__(self, '_thunk')
        if thunk is None:
            return object.__getattribute__(self, '_fa')._host_value()
        true = object.__getattribute__(self, '_true')
        if true is None:
            true = thunk()
            self._true = true
        return true

    def __getattribute__(self, name):
        if name in _HV_OWN:
            return object.__getattribute__(self, name)
        return getattr(object.__getattribute__(self, '_real')(), name)

    def _dev_ok(self) -> bool:
        if not self._is_lazy or self._outer_src is not None:
            # (a symbolic np.right_shift.outer view: _fa is the UN-shifted source, only `& 1` and `.T` may use it --
            # both test _outer_src before they come here; everything else materialises)
            return False
        if self._fa_value is None:
            return True                       # deferred views are only ever created on the prime-field device path
        return not _fops(type(self._fa_value).field).binary

    def _derive(self, fa, exact, thunk):
        return HostView(fa, lazy=True, exact=exact, thunk=thunk)

    def _operand(self, other):
        """-> (device operand for FieldArray arithmetic, host operand for the thunk) or None"""
        if isinstance(other, HostView):
            if type(other._fa).field is not type(self._fa).field:
                return None
            return other._fa, other._real
        if isinstance(other, (int, np.integer)) and not isinstance(other, (bool, np.bool_)):
            return int(other), (lambda v=int(other): v)
        if isinstance(other, np.ndarray) and other.dtype.kind in 'iub':
            return _uploaded_public(type(self._fa), other), (lambda v=other: v)      # public integer arrays (shifts, masks, bits)
        return None

    def _binop(self, other, name, reflected=False):
        opd = self._operand(other) if self._dev_ok() else None
        if opd is None:
            a, b = self._real(), _hv_unwrap(other)
            return getattr(np, name)(b, a) if reflected else getattr(np, name)(a, b)
        dev, host = opd
        fa = self._fa
        if name == 'add':
            res = fa + dev
        elif name == 'subtract':
            res = (dev - fa) if reflected else (fa - dev)
        else:
            res = fa * dev
        mine = self._real
        fn = getattr(np, name)
        thunk = (lambda: fn(host(), mine())) if reflected else (lambda: fn(mine(), host()))
        return self._derive(res, False, thunk)

    def _shift_left(self, k):
        if self._dev_ok():
            p = _fops(type(self._fa).field).modulus
            if isinstance(k, (int, np.integer)) and int(k) >= 0:
                return self._derive(self._fa * pow(2, int(k), p), False, lambda: self._real() << int(k))
            if isinstance(k, np.ndarray) and k.dtype.kind in 'iu' and k.size <= (1 << 16) and (k >= 0).all():
                fa = self._fa
                pw = type(fa)(np.array([pow(2, int(v), p) for v in k.reshape(-1)], dtype=object).reshape(k.shape))
                shape = tuple(np.broadcast_shapes(self.shape, k.shape))
                out = HostView(lambda: fa * pw, lazy=True, exact=False, thunk=lambda: self._real() << k, shape=shape)
                if k.ndim == 1 and self.ndim >= 2 and self.shape[-1] == k.shape[0]:
                    out._shift_src = (fa, pw)          # sum(x << shifts, axis=-1) is ONE matrix-vector product
                return out
        return self._real() << _hv_unwrap(k)

    def _bits_and(self, mask):
        """x & mask for exact views: limb-wise AND on the device (mask = 1: the bit itself, returned as a real int8
        array -- what np.int8(... & 1) consumes, runtime.py:4443)."""
        src = self._outer_src
        if src is not None and isinstance(mask, (int, np.integer)) and int(mask) == 1:
            fa, shifts = src
            bits = fa._bit_matrix(shifts)
            return bits.T if self._outer_T else bits
        if self._dev_ok() and self._exact and isinstance(mask, (int, np.integer)) and int(mask) >= 0:
            mask = int(mask)
            res = self._fa._and_mask(mask)
            if mask == 1:
                return res._small_ints(np.int8)
            return self._derive(res, True, None)
        return self._real() & _hv_unwrap(mask)

    def transpose(self, *axes):
        if self._outer_src is not None and not axes and len(self.shape) == 2:
            out = np.broadcast_to(_POISON, self.shape[::-1]).view(HostView)
            out._fa, out._is_lazy, out._exact, out._true = self._fa, True, True, None
            mine = self._real
            out._thunk = lambda: mine().T
            out._outer_src, out._outer_T = self._outer_src, not self._outer_T
            return out
        if self._dev_ok() and self._outer_src is None:
            mine = self._real
            return self._derive(self._fa.transpose(*axes), self._exact, lambda: mine().transpose(*axes))
        return self._real().transpose(*axes)

    @property
    def T(self):
        return self.transpose()

    # shape-only / additive NumPy functions commute with reduction mod p: run them on the device arrays
    _LIFTED = frozenset(('vstack', 'hstack', 'concatenate', 'stack', 'cumsum', 'transpose', 'reshape', 'ravel', 'flip',
                         'roll', 'squeeze', 'expand_dims', 'swapaxes', 'moveaxis', 'tile', 'repeat', 'diff', 'where'))

    @staticmethod
    def _lift(func, args, kwargs):
        """func(*args) where some (nested) arguments are device views of ONE prime field and the rest public integer
        arrays: evaluate on FieldArrays; None if an argument does not fit"""
        cls = [None]

        def find(x):
            if isinstance(x, HostView):
                if not x._dev_ok() or x._outer_src is not None:
                    raise LookupError
                c = type(x._fa)
                if cls[0] is not None and cls[0].field is not c.field:
                    raise LookupError
                cls[0] = c
            elif isinstance(x, (list, tuple)):
                for y in x:
                    find(y)

        def dev(x):
            if isinstance(x, HostView):
                return x._fa
            if isinstance(x, np.ndarray):
                if x.dtype.kind == 'b':
                    return x                   # conditions / masks stay what they are (np.where)
                if x.dtype.kind not in 'iuO':
                    raise LookupError
                return cls[0](x)
            if isinstance(x, (list, tuple)):
                return type(x)(dev(y) for y in x)
            return x
        try:
            find(args)
            if cls[0] is None:
                return None
            dargs = dev(args)
            res = func(*dargs, **kwargs)
        except (LookupError, TypeError, ValueError, NotImplementedError):
            return None
        if not isinstance(res, FieldArray):
            return None
        return HostView(res, lazy=True, exact=False, thunk=lambda: func(*_hv_unwrap(args), **_hv_unwrap(kwargs)))

    # -- stays on the device --
    def reshape(self, *shape, **kw):
        if self._thunk is None:
            return HostView(self._fa.reshape(*shape, **kw), lazy=self._is_lazy or None)
        return self._derive(self._fa.reshape(*shape, **kw), self._exact, lambda: self._real().reshape(*shape, **kw))

    def __add__(self, other):
        return self._binop(other, 'add')

    __radd__ = __add__
    __iadd__ = __add__

    def __sub__(self, other):
        return self._binop(other, 'subtract')

    __isub__ = __sub__

    def __rsub__(self, other):
        return self._binop(other, 'subtract', reflected=True)

    def __mul__(self, other):
        return self._binop(other, 'multiply')

    __rmul__ = __mul__
    __imul__ = __mul__

    def __lshift__(self, k):
        return self._shift_left(k)

    __ilshift__ = __lshift__

    def __pow__(self, e):
        if self._dev_ok() and isinstance(e, (int, np.integer)) and 0 <= int(e) <= 64:
            return self._derive(self._fa ** int(e), False, lambda: self._real() ** int(e))
        return self._real() ** _hv_unwrap(e)

    def __mod__(self, m):
        if self._dev_ok() and isinstance(m, (int, np.integer)):
            m = int(m)
            if m == _fops(type(self._fa).field).modulus:
                return self._derive(self._fa, True, None)              # the canonical residues are x mod p
            if self._exact and m > 0 and m & (m - 1) == 0:
                return self._bits_and(m - 1)
        return self._real() % _hv_unwrap(m)

    __imod__ = __mod__

    def __and__(self, mask):
        return self._bits_and(mask)

    __rand__ = __and__
    __iand__ = __and__

    def _cmp_on_device(self, other) -> bool:
        # integers in [0, p) only: FieldArray comparison reduces its scalar, the reference compares integers (x == -1 and
        # x == p are False for every residue)
        return self._dev_ok() and self._exact and isinstance(other, (int, np.integer)) and \
            0 <= int(other) < type(self._fa).field.modulus

    def __ne__(self, other):
        if self._cmp_on_device(other):
            return self._fa != int(other)
        return self._real() != _hv_unwrap(other)

    def __eq__(self, other):
        if self._cmp_on_device(other):
            return self._fa == int(other)
        return self._real() == _hv_unwrap(other)

    def __getitem__(self, key):
        if self._dev_ok() and (isinstance(key, np.ndarray) or isinstance(key, (slice, tuple))):
            try:
                sub = self._fa[key]
            except (IndexError, TypeError, ValueError):
                return self._real()[_hv_unwrap(key)]
            if isinstance(sub, FieldArray):
                if self._thunk is None:
                    return HostView(sub, lazy=True)
                return self._derive(sub, self._exact, lambda: self._real()[key])
        return self._real()[_hv_unwrap(key)]

    def __setitem__(self, key, value):
        """`a.value[key] = v` writes through to the array, as it does in the reference where `.value` IS the
        storage (e.g. runtime.py:3975)."""
        fa = self._fa
        if self._thunk is not None:
            raise ValueError('assignment into a derived integer view')
        real = fa._host_value()
        real.flags.writeable = True
        try:
            real[key] = _hv_unwrap(value)
        finally:
            real.flags.writeable = False
        fa[key] = real[key]
        fa._cache = real

    def __reduce__(self):
        if self._thunk is not None:
            return self._real().__reduce__()           # derived integers: pickle the true values
        return self._fa.__reduce__()

    def __reduce_ex__(self, protocol):
        return self.__reduce__()

    def __copy__(self):
        return HostView(self._fa)

    def __deepcopy__(self, memo):
        return HostView(self._fa.copy())

    # -- everything else: the materialised ndarray --
    def __array_function__(self, func, types, args, kwargs):
        name = func.__name__
        if name == 'sum' and args and isinstance(args[0], HostView) and args[0]._dev_ok():
            v = args[0]
            axis = kwargs.get('axis', args[1] if len(args) > 1 else None)
            if isinstance(axis, (int, np.integer)) and v.ndim > 1 and set(kwargs) <= {'axis'} and len(args) <= 2:
                ax = int(axis) if int(axis) >= 0 else int(axis) + v.ndim
                if v._shift_src is not None and ax == v.ndim - 1:
                    # np.sum(bits << shifts, axis=-1) (np_trunc, np_to_bits, np_from_bits, np_sgn): the weighted row sums
                    # straight from the un-shifted array, one pass (x @ [2^k mod p])
                    src, pw = v._shift_src
                    rows = src.reshape(-1, src.shape[-1]) @ pw
                    return v._derive(rows.reshape(src.shape[:-1]), False, lambda: np.sum(v._real(), axis=ax))
                return v._derive(v._fa.sum(axis=ax), False, lambda: np.sum(v._real(), axis=ax))
        if name == 'count_nonzero' and len(args) == 1 and not kwargs and args[0]._dev_ok() and args[0]._exact:
            return int(np.count_nonzero(args[0]._fa != 0))
        if name in HostView._LIFTED and not any(isinstance(v, HostView) for v in kwargs.values()):
            res = HostView._lift(func, args, kwargs)
            if res is not None:
                return res
        return func(*_hv_unwrap(args), **_hv_unwrap(kwargs))

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        name = ufunc.__name__
        if not kwargs or set(kwargs) <= {'axis'}:
            a = inputs[0]
            if method == '__call__' and len(inputs) == 2:
                b = inputs[1]
                first = isinstance(a, HostView)
                me, other = (a, b) if first else (b, a)
                if name in ('add', 'subtract', 'multiply') and not kwargs:
                    return me._binop(other, name, reflected=not first)
                if name == 'left_shift' and first and not kwargs:
                    return me._shift_left(other)
                if name == 'bitwise_and' and not kwargs:
                    return me._bits_and(other)
                if name == 'remainder' and first and not kwargs:
                    return me.__mod__(other)
            if method == 'reduce' and name == 'add' and isinstance(a, HostView) and a._dev_ok() and a.ndim > 1:
                axis = kwargs.get('axis', 0)
                if isinstance(axis, (int, np.integer)):
                    return a._derive(a._fa.sum(axis=int(axis)), False, lambda: np.add.reduce(a._real(), axis=int(axis)))
            if method == 'outer' and name == 'right_shift' and len(inputs) == 2 and isinstance(a, HostView) and \
                    a._dev_ok() and a._exact and isinstance(inputs[1], np.ndarray) and inputs[1].dtype.kind in 'iu' and \
                    inputs[1].size <= 512 and (inputs[1] >= 0).all() and not kwargs:
                # np.right_shift.outer(c, shifts) (runtime.py:4443): kept symbolic; `& 1` extracts the bits on the device
                shifts = inputs[1]
                out = np.broadcast_to(_POISON, a.shape + shifts.shape).view(HostView)
                out._fa, out._is_lazy, out._exact, out._true = a._fa, True, True, None
                out._thunk = lambda: np.right_shift.outer(a._real(), shifts)
                out._outer_src, out._outer_T = (a._fa, shifts), False
                return out
        return getattr(ufunc, method)(*_hv_unwrap(inputs), **_hv_unwrap(kwargs))

    __hash__ = None


def _hv_unwrap(x):
    if isinstance(x, HostView):
        return x._real()
    if isinstance(x, (list, tuple)):
        return type(x)(_hv_unwrap(v) for v in x)
    if isinstance(x, dict):
        return {k: _hv_unwrap(v) for k, v in x.items()}
    return x


def _hv_delegate(name):
    def op(self, *args):
        return getattr(self._real(), name)(*_hv_unwrap(args))
    op.__name__ = name
    return op


for _n in ('add', 'sub', 'mul', 'matmul', 'truediv', 'floordiv', 'mod', 'divmod', 'pow', 'lshift', 'rshift',
           'and', 'or', 'xor'):
    for _m in (f'__{_n}__', f'__r{_n}__'):
        if _m not in HostView.__dict__:
            setattr(HostView, _m, _hv_delegate(_m))
    if f'__i{_n}__' not in HostView.__dict__:
        setattr(HostView, f'__i{_n}__', _hv_delegate(f'__{_n}__'))      # the snapshot is read-only: x op= y rebinds
for _n in ('neg', 'pos', 'abs', 'invert', 'lt', 'le', 'gt', 'ge', 'eq', 'ne', 'getitem', 'iter', 'contains', 'bool',
           'repr', 'str', 'int', 'float', 'index', 'array', 'format'):
    if f'__{_n}__' not in HostView.__dict__:
        setattr(HostView, f'__{_n}__', _hv_delegate(f'__{_n}__'))
del _n, _m


_field_registry = {}      # _field_key -> field class: where unpickled share rows find their array type


def _field_key(field, ops):
    """What identifies a field CLASS on the wire.  mpyc caches prime fields per (p, n, w) (finfields.py:347-363): GF(p)
    and GF((p, n, w)) with a root of unity are different classes over the same modulus, and an array of one is not an
    array of the other (np_recombine / field.array raise on the mix), so nth / root are part of the key."""
    return (ops.binary, ops.modulus, getattr(field, 'nth', None), getattr(field, 'root', None))


def _array_from_ipc(key, shape, n, desc):
    """Unpickle hook of the device-side wire for co-located parties (mpyc_amd/ipcwire.py): the message carried an
    interprocess handle of the peer's device buffer, not the limb bytes -- copy the row device-to-device."""
    field = _field_registry.get(tuple(key))
    if field is None:
        raise TypeError(f'no field with modulus {key[1]:#x} (nth, root = {key[2:]}) has been created in this process')
    ctx = _context(field)
    from . import ipcwire
    t = ipcwire.fetch(ctx, desc, reduce_n=n)                 # a peer's data: canonical before any kernel sees it
    return field.array._wrap(DevArray(ctx, t, n), tuple(shape))


def _array_from_wire(key, shape, data):
    """Unpickle hook: rebuild a device array from field.to_bytes-format limb bytes (see FieldArray.__reduce__)."""
    field = _field_registry.get(tuple(key))
    if field is None:
        raise TypeError(f'no field with modulus {key[1]:#x} (nth, root = {key[2:]}) has been created in this process')
    return field.array.from_wire(data, shape)


class FieldArray:
    """GPU-backed counterpart of finfields.FiniteFieldArray (finfields.py:695-1368).

    ctor (value, check=True, copy=False) as in the reference: `value` may be nested lists / ndarrays
    of ints (any integer dtype or object), another FieldArray, or an engine DevArray; with
    check=True inputs are reduced into canonical form (negative ints wrap Python-style)."""

    __slots__ = ()          # behaviour only: the storage slots live in the concrete leaf classes (_STORAGE_SLOTS),
    #                         so that install() can derive the array type from mpyc's own FiniteFieldArray as well
    field = None            # set per field by GF()
    __array_priority__ = 100

    def __init__(self, value, check=True, copy=False):
        F = type(self).field
        ctx = _context(F)
        self._cache = None
        if isinstance(value, HostView):
            value = value._fa                 # `.value` of an array handed back (runtime.py:508): still on the device
        if isinstance(value, FieldArray):
            if value.field is not F:
                raise TypeError('array over a different field')
            self._dev = value._dev.clone() if copy else value._dev
            self._shape = value._shape
            return
        if isinstance(value, DevArray):
            self._dev = ctx.reduce(value) if check else (value.clone() if copy else value)
            self._shape = (value.n,)
            return
        if isinstance(value, torch.Tensor):
            raise TypeError('wrap raw limb tensors in an engine.DevArray')
        # lists go through dtype=object so that big Python ints are never coerced to float64
        a = value if isinstance(value, np.ndarray) else np.array(value, dtype=object)
        if a.dtype.kind == 'f' and not check:
            # `Zp.array(np.empty(N), check=False)` (demos/np_lpsolver.py): an uninitialised placeholder that the
            # caller fills by item assignment; the reference keeps the float garbage, limbs start from zero
            a = np.zeros(a.shape, dtype=np.int64)
        if (a.dtype.kind == 'f' or a.dtype.kind == 'c') and a.size:               # np.array([]) is float64: allowed
            raise TypeError('float values are not field elements')            # tests/test_finfields.py:372-382
        shape = a.shape
        flat = a.reshape(-1)
        fast = None
        if flat.dtype == object and flat.size:
            # one C-level pass over the element TYPES instead of per-element isinstance tests (10^5..10^7 elements
            # per call on the runtime's hot path)
            kinds = set(map(type, flat.tolist()))
            if any(issubclass(k, (float, complex, np.floating, np.complexfloating)) for k in kinds):
                raise TypeError('float values are not field elements')
            if not all(issubclass(k, int) for k in kinds):       # field elements, numpy integers, polynomials, ...
                flat = np.array([int(v) if isinstance(v, (int, np.integer)) else int(getattr(v, 'value', v))
                                 for v in flat], dtype=object)
            if check and ctx.elem_bytes <= 8 and not _fops(F).binary:
                fast = self._limbs_one_word(flat, ctx.elem_bytes, _fops(F).modulus)
        elif flat.dtype.kind == 'i' and flat.size and check and ctx.elem_bytes <= 8 and not _fops(F).binary:
            fast = self._limbs_one_word(flat, ctx.elem_bytes, _fops(F).modulus)      # signed NumPy integers (weights, +-1)
        signed_wide = None
        if check and flat.size and fast is None and ctx.elem_bytes > 8 and not _fops(F).binary and flat.dtype.kind in 'iO':
            signed_wide = self._signed_words(flat)
        if not flat.size:
            self._dev = ctx.empty(0)
        elif signed_wide is not None:
            # multi-limb prime (p > 2^64) and every value fits a signed 64-bit word (weights, +-1, small constants):
            # upload the magnitudes as the low limb -- already canonical -- and negate on the device where the sign
            # was set; the alternative is a Python-level `%` per element (finfields.py:724 does exactly that)
            mag, neg = signed_wide
            dev = ctx.from_numpy(ints_to_np(mag, ctx.elem_bytes))
            if neg is not None:
                mask = torch.from_numpy(neg).to(dev.t.device)
                dev.t.copy_(torch.where(mask.unsqueeze(-1), ctx.neg(dev).t, dev.t))
            self._dev = dev
        elif fast is not None:
            self._dev = ctx.reduce(ctx.from_numpy(fast))
        elif not check:
            self._dev = ctx.from_numpy(ints_to_np(flat, ctx.elem_bytes))
        elif self._fits_limbs(flat, ctx.elem_bytes, F):
            # the usual case: non-negative values that fit the limb width -> upload as they are and let
            # the device do `value %= modulus` (ffgpu_reduce, finfields.py:724)
            self._dev = ctx.reduce(ctx.from_numpy(ints_to_np(flat, ctx.elem_bytes)))
        else:
            # negative or wider-than-limb Python ints cannot be expressed as limbs: canonicalise those on
            # the host while marshalling (the reference does the same `%` on the host for every input)
            self._dev = ctx.from_numpy(ints_to_np(self._canonical_host(flat, F), ctx.elem_bytes))
        self._shape = tuple(shape)

    @staticmethod
    def _signed_words(flat):
        """-> (|v| as uint64, boolean mask of the negative entries or None) if every value fits a signed 64-bit word,
        else None"""
        try:
            a64 = flat if flat.dtype.kind == 'i' else flat.astype(np.int64)
        except (OverflowError, TypeError):
            return None
        a64 = a64.astype(np.int64, copy=False)
        neg = a64 < 0
        if not neg.any():
            return a64.view(np.uint64) if a64.dtype == np.int64 else a64.astype(np.uint64), None
        if (a64 == np.iinfo(np.int64).min).any():
            return None
        return np.abs(a64).astype(np.uint64), neg

    @staticmethod
    def _limbs_one_word(flat, eb, p):
        """Object array of Python ints -> one-limb array congruent mod p, or None if some value needs the general
        path.  Signed 64-bit conversion inside NumPy (raises on overflow), negatives folded with a vectorised `%`
        (Python semantics: result in [0, p), finfields.py:724) -- the device reduces afterwards."""
        try:
            a64 = flat.astype(np.int64)
        except (OverflowError, TypeError):
            try:
                u64 = flat.astype(np.uint64)              # values in [2^63, 2^64)
            except (OverflowError, TypeError):
                return None
            return u64 if eb == 8 else None
        if p < (1 << 63):
            if eb == 4:
                return (a64 % np.int64(p)).astype(np.uint32)
            neg = a64 < 0
            if neg.any():
                a64 = np.where(neg, a64 % np.int64(p), a64)
            return a64.view(np.uint64)
        neg = a64 < 0                                      # 64-bit moduli: v in (-2^63, 0) -> p - |v|
        u = a64.view(np.uint64).copy()
        if neg.any():
            u[neg] = np.uint64(p) - (-a64[neg]).view(np.uint64)
        return u

    @staticmethod
    def _fits_limbs(flat, eb, F):
        if flat.dtype != object:
            if flat.dtype.kind == 'u':
                return flat.dtype.itemsize <= eb
            if flat.dtype.kind == 'i':
                return flat.dtype.itemsize <= eb and bool((flat >= 0).all())
            return False
        lo, hi = min(flat), max(flat)
        return lo >= 0 and hi < (1 << (8 * eb))

    @staticmethod
    def _canonical_host(flat, F):
        """`value %= modulus` (finfields.py:724) for inputs that arrive as host integers."""
        ops = _fops(F)
        if ops.binary:
            mod = ops.modulus
            if flat.dtype != object:
                flat = flat.astype(object)
            return np.array([_clmod(abs(int(v)), mod) for v in flat], dtype=object)
        p = ops.modulus
        if flat.dtype != object:
            if flat.dtype.kind == 'u' and flat.dtype.itemsize * 8 < p.bit_length():
                return flat                                                   # already < p
            flat = flat.astype(object)
        return flat % p

    # ---- deferred element-wise product ----------------------------------------------------
    # `a * b` on equal-shape arrays is recorded, not launched: if the next thing that happens to it is
    # share generation (the reference's np_multiply -> _reshare, runtime.py:1134-1138) the product is
    # fused into the split kernel and never written to HBM (ffgpu_mul_split).  Any other use
    # materialises it with the ordinary mul kernel.
    @property
    def _dev(self) -> DevArray:
        if self._devv is None:
            a, b = self._lazy
            A = _src_dev(a)
            self._devv = A if b is None else A.ctx.mul(A, A if b is a else _src_dev(b))
            self._lazy = None
        return self._devv

    @_dev.setter
    def _dev(self, v):
        self._devv = v
        self._lazy = None

    @staticmethod
    def _flush_products_reading(dev: DevArray):
        """An in-place update of `dev` is about to happen: materialise every pending product that still
        reads that buffer, so that deferred evaluation never observes the later mutation."""
        if not _pending_products:
            return
        ptr = _storage_id(dev)
        alive = []
        for ref in _pending_products:
            arr = ref()
            if arr is None or arr._devv is not None:
                continue
            lz = arr._lazy
            if any(src is not None and (src.reads(ptr) if isinstance(src, _Rec) else _storage_id(src) == ptr)
                   for src in lz):
                arr._dev            # noqa: B018  (property access materialises)
            else:
                alive.append(ref)
        _pending_products[:] = alive

    def _take_lazy_product(self):
        """(a, b) operands if this array is unmaterialised, else None.  Each operand is a DevArray or a
        deferred recombination (_Rec); b is None for a bare deferred recombination."""
        return self._lazy if self._devv is None else None

    def _source(self):
        """What a product should read for this array: the deferred recombination if that is all it is, else
        the device data (materialising a deferred product)."""
        if self._devv is None and self._lazy is not None and self._lazy[1] is None and isinstance(self._lazy[0], _Rec):
            return self._lazy[0]
        return self._dev

    @classmethod
    def _wrap_lazy_rec(cls, rows, lam, shape) -> 'FieldArray':
        return cls._wrap_lazy_product(_Rec(rows, lam), None, shape)

    # ---- representation --------------------------------------------------------------
    @property
    def ctx(self) -> FieldContext:
        return (self._lazy[0] if self._devv is None else self._devv).ctx        # DevArray and _Rec both have .ctx

    @property
    def value(self) -> 'HostView':
        """The reference's representation (finfields.py:703-709: object ndarray of canonical Python ints,
        BinaryPolynomial objects for GF(2^n)) as a lazy HostView: shape / reshape / pickling / being handed back to
        thresha or the array ctor stay on the device; any other use materialises the object ndarray once
        (read-only snapshot)."""
        if not HostView._runtime_scanned:
            HostView._scan_runtime()
        if HostView.lazy_codes:
            frame = sys._getframe(1)
            code = frame.f_code
            if code in HostView.lazy_codes:
                return HostView(self, lazy=True)
            if code in HostView.lazy_lines and frame.f_lineno in HostView.lazy_lines[code]:
                return HostView(self, lazy=True)
            if code in HostView.lazy_codes_prime and self.size >= HostView.lazy_ints_min and not _fops(type(self).field).binary:
                return HostView(self, lazy=True)
        return HostView(self)

    def _host_value(self) -> np.ndarray:
        if self._cache is None:
            ops = _fops(type(self).field)
            dev = self._dev
            v = np_to_objects(dev.to_numpy(), dev.ctx.elem_bytes)
            if ops.binary:
                boxed = np.empty(len(v), dtype=object)
                boxed[:] = [ops.box(x) for x in v.tolist()]
                v = boxed
            v = v.reshape(self._shape)
            v.flags.writeable = False
            self._cache = v
        return self._cache

    @value.setter
    def value(self, v):
        other = type(self)(v, check=False)
        self._dev, self._shape, self._cache = other._dev, other._shape, None

    @property
    def device_array(self) -> DevArray:
        return self._dev

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def size(self):
        return _prod(self._shape)          # (math.prod on Python ints: a tenth of np.prod's cost on the runtime's hot path)

    def __len__(self):
        if not self._shape:
            raise TypeError('len() of unsized object')
        return self._shape[0]

    @classmethod
    def _wrap(cls, dev: DevArray, shape) -> 'FieldArray':
        o = cls.__new__(cls)
        o._dev, o._shape, o._cache = dev, tuple(shape), None
        return o

    @classmethod
    def _wrap_lazy_product(cls, a: DevArray, b: DevArray, shape) -> 'FieldArray':
        o = cls.__new__(cls)
        o._devv, o._lazy, o._shape, o._cache = None, (a, b), tuple(shape), None
        if len(_pending_products) > 64:      # drop dead / materialised entries now and then
            _pending_products[:] = [r for r in _pending_products if r() is not None and r()._devv is None]
        _pending_products.append(weakref.ref(o))
        return o

    def copy(self, order='C'):
        return self._wrap(self._dev.clone(), self._shape)

    def reshape(self, *shape, order='C'):
        if order in ('F', 'f'):
            rs = self._reshape_shape(shape)
            return self.transpose().reshape(tuple(reversed(rs))).transpose()
        if self._devv is None and self._lazy is not None:
            # keep the product deferred through the "in-place flatten" the runtime does before sharing
            o = self._wrap_lazy_product(self._lazy[0], self._lazy[1], self._shape)
            o._shape = self._reshape_shape(shape)
            return o
        return self._wrap(self._dev, self._reshape_shape(shape))                # view: same device data

    def _reshape_shape(self, shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        n = self.size
        if -1 in shape:
            known = -_prod(shape)
            shape = tuple(n // known if s == -1 else s for s in shape)
        if _prod(shape) != n:
            raise ValueError(f'cannot reshape array of size {n} into shape {shape}')
        return tuple(shape)

    def flatten(self, order='C'):
        if order in ('F', 'f'):
            return self.transpose().flatten()
        return self._wrap(self._dev.clone(), (self.size,))

    def ravel(self, order='C'):
        return self.reshape(-1, order=order)

    def _limb_view(self):
        """limb tensor shaped like the array (+ trailing 2 for two-limb fields)."""
        t = self._dev.t
        return t.reshape(tuple(self._shape) + ((self.ctx.limbs,) if self.ctx.limbs else ()))

    def _index_key(self, key):
        """NumPy-style index -> torch index on the limb view (index arrays / masks move to the device)."""
        dev = self._dev.t.device

        def conv(k):
            if isinstance(k, np.ndarray):
                return torch.from_numpy(np.ascontiguousarray(k)).to(dev)
            if isinstance(k, (np.integer,)):
                return int(k)
            if isinstance(k, list):
                a = np.asarray(k)
                if a.dtype.kind in 'iub':          # (nested) lists of indices / booleans index like arrays
                    return torch.from_numpy(a if a.dtype.kind == 'b' else a.astype(np.int64)).to(dev)
            return k
        if isinstance(key, tuple):
            key = tuple(conv(k) for k in key)
            if self.ctx.limbs and Ellipsis not in key:
                key = key + (Ellipsis, slice(None))
            elif self.ctx.limbs:
                key = key + (slice(None),)
            return key
        key = conv(key)
        return (key, Ellipsis, slice(None)) if self.ctx.limbs and key is not Ellipsis else key

    def _positive_steps(self, key):
        """torch has no negative slice steps: a[::-1] and friends are rewritten as the ascending slice over the same
        elements plus a flip of the result along that axis.  -> (key, axes of the RESULT to flip); basic keys only."""
        keys = key if isinstance(key, tuple) else (key,)
        if not any(isinstance(k, slice) and k.step is not None and k.step < 0 for k in keys):
            return key, ()
        if any(not (isinstance(k, (int, np.integer, slice)) or k is None or k is Ellipsis) for k in keys):
            raise NotImplementedError('negative slice steps together with index arrays on GPU field arrays')
        n_real = sum(1 for k in keys if k is not None and k is not Ellipsis)
        out, flips, dim, axis = [], [], 0, 0
        for k in keys:
            if k is Ellipsis:
                skip = self.ndim - n_real
                dim += skip
                axis += skip
                out.append(k)
            elif k is None:
                axis += 1
                out.append(k)
            elif isinstance(k, slice):
                if k.step is not None and k.step < 0:
                    start, stop, step = k.indices(self._shape[dim])
                    cnt = max(0, (start - stop - step - 1) // (-step))
                    if cnt == 0:
                        out.append(slice(0, 0, 1))
                    else:
                        last = start + (cnt - 1) * step
                        out.append(slice(last, start + 1, -step))
                        flips.append(axis)
                else:
                    out.append(k)
                dim += 1
                axis += 1
            else:
                out.append(k)
                dim += 1
        return (tuple(out) if isinstance(key, tuple) else out[0]), tuple(flips)

    def __getitem__(self, key):
        key, flips = self._positive_steps(key)
        sub = self._limb_view()[self._index_key(key)]
        if flips:
            sub = sub.flip(flips)
        shape = sub.shape[:-1] if self.ctx.limbs else sub.shape
        if len(shape) == 0:
            flat = sub.reshape(1, self.ctx.limbs) if self.ctx.limbs else sub.reshape(1)
            return type(self).field(_fops(type(self).field).box(DevArray(self.ctx, flat, 1).to_ints()[0]))
        n = int(np.prod(shape, dtype=np.int64))
        flat = sub.reshape(n, self.ctx.limbs) if self.ctx.limbs else sub.reshape(n)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        return self._wrap(DevArray(self.ctx, flat, n), shape)

    def __setitem__(self, key, value):
        cls = type(self)
        if isinstance(value, HostView):
            value = value._fa
        if isinstance(value, (int, np.integer)) or _scalar_value(value) is not None:
            value = cls([int(value) if isinstance(value, (int, np.integer)) else _scalar_value(value)]).reshape(())
        elif not isinstance(value, FieldArray):
            value = cls(value)
        # shape check as the reference does it (finfields.py:1019-1027): object ndarrays would not complain
        target = np.broadcast_to(np.int8(0), self._shape)[key].shape
        try:
            ok = np.broadcast_shapes(value._shape, target) == tuple(target)
        except ValueError:
            ok = False
        if not ok:
            raise ValueError(f'could not broadcast input array from shape {value._shape} into shape {tuple(target)}')
        src = value._limb_view()
        self._flush_products_reading(self._dev)
        key2, flips = self._positive_steps(key)
        if flips:                                   # a[::-1] = v  ==  a[ascending slice] = v flipped (after broadcasting)
            nl = 1 if self.ctx.limbs else 0
            src = src.broadcast_to(tuple(target) + ((self.ctx.limbs,) if nl else ())).flip(flips)
        self._limb_view()[self._index_key(key2)] = src
        self._cache = None

    def __contains__(self, value):
        """NumPy's `in`: any element equal, after broadcasting (finfields.py:995-1005)."""
        if isinstance(value, HostView):
            value = value._fa
        return bool(np.any(self == value))

    @property
    def flat(self):
        field = type(self).field
        for a in self._host_value().flat:
            yield field(a)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # ---- operand handling (finfields.py:1045-1054 _coerce) ---------------------------------
    def _operand(self, other):
        """-> ('array', FieldArray) | ('scalar', int) | None"""
        cls, F = type(self), type(self).field
        ops = _fops(F)
        if isinstance(other, FieldArray):
            if other.field is not F:
                raise TypeError('arrays over different fields')
            return ('array', other) if other.size != 1 or other.ndim > self.ndim else ('scalar', other._dev.to_ints()[0])
        if isinstance(other, (int, np.integer, np.bool_)):          # bool is an int (the reference's _mix_types)
            return 'scalar', ops.reduce_int(int(other))
        if isinstance(other, float):
            raise TypeError('float operand')
        if isinstance(other, (np.ndarray, list, tuple)):
            arr = cls(other)                         # raises TypeError for float dtypes
            return ('array', arr) if arr.size != 1 else ('scalar', arr._dev.to_ints()[0])
        if hasattr(other, 'modulus') and hasattr(other, 'value'):      # a field element
            if type(other) is not F:
                raise TypeError('element of a different field')
            return 'scalar', int(other.value)
        if ops.binary and ops.poly_type is not None and isinstance(other, (ops.poly_type, BinaryPolynomial)):
            return 'scalar', ops.reduce_int(int(other))
        return None

    def _broadcast(self, other: 'FieldArray'):
        if self._shape == other._shape:
            return self._dev, other._dev, self._shape
        shape = tuple(np.broadcast_shapes(self._shape, other._shape))
        out = []
        for x in (self, other):
            if x._shape == shape:
                out.append(x._dev)
                continue
            t = x._limb_view()
            if x.ctx.limbs:
                t = t.expand(*shape, x.ctx.limbs).contiguous().view(-1, x.ctx.limbs)
            else:
                t = t.expand(*shape).contiguous().view(-1)
            out.append(DevArray(x.ctx, t, t.shape[0]))
        return out[0], out[1], shape

    def _binop(self, other, arr_op, scalar_op, reflected_scalar_op=None, inplace=False):
        opd = self._operand(other)
        if opd is None:
            return NotImplemented
        ctx = self.ctx
        kind, o = opd
        if inplace:
            self._flush_products_reading(self._dev)
        if kind == 'scalar':
            fn = reflected_scalar_op or scalar_op
            res = fn(ctx, self._dev, o, self._dev if inplace else None)
            shape = self._shape
        else:
            a, b, shape = self._broadcast(o)
            if inplace and shape != self._shape:
                raise ValueError('non-broadcastable output operand')
            res = arr_op(ctx, a, b, self._dev if inplace else None)
        if inplace:
            self._cache = None
            return self
        return self._wrap(res, shape)

    # ---- arithmetic (finfields.py:1056-1124, 1189-1197) --------------------------------------
    def __add__(self, other):
        return self._binop(other, _ctx_add, _ctx_add_scalar)

    __radd__ = __add__

    def __iadd__(self, other):
        return self._binop(other, _ctx_add, _ctx_add_scalar, inplace=True)

    def __sub__(self, other):
        def sub_scalar(ctx, a, s, out):
            return ctx.add_scalar(a, _fops(type(self).field).sub(0, s), out)
        return self._binop(other, _ctx_sub, sub_scalar)

    def __rsub__(self, other):
        opd = self._operand(other)
        if opd is None:
            return NotImplemented
        kind, o = opd
        if kind == 'scalar':
            return self._wrap(self.ctx.rsub_scalar(self._dev, o), self._shape)
        return o.__sub__(self)

    def __isub__(self, other):
        def sub_scalar(ctx, a, s, out):
            return ctx.add_scalar(a, _fops(type(self).field).sub(0, s), out)
        return self._binop(other, _ctx_sub, sub_scalar, inplace=True)

    def __mul__(self, other):
        if isinstance(other, FieldArray) and other.field is type(self).field and other._shape == self._shape \
                and self.size > 1 and lazy_products:
            a = self._source()
            return self._wrap_lazy_product(a, a if other is self else other._source(), self._shape)
        return self._binop(other, _ctx_mul, _ctx_mul_scalar)

    __rmul__ = __mul__

    def __imul__(self, other):
        return self._binop(other, _ctx_mul, _ctx_mul_scalar, inplace=True)

    def __neg__(self):
        return self._wrap(self.ctx.neg(self._dev), self._shape)

    def __pos__(self):
        return self.copy()

    def __pow__(self, e):
        if not isinstance(e, (int, np.integer)) or isinstance(e, bool):
            return NotImplemented
        e = int(e)
        if e < 0:
            return self.reciprocal() ** (-e)
        order1 = _fops(type(self).field).order - 1
        if e >> 192:
            e = e % order1 if e % order1 or not e else order1      # a^(q-1) = 1 for a != 0, keep 0^e = 0
        return self._wrap(self.ctx.pow(self._dev, e), self._shape)    # one kernel: finfields.py:1159-1187

    def reciprocal(self):
        """Element-wise inverse, batched on the device (finfields.py:1278-1281, :1416-1422); raises
        ZeroDivisionError if any element is zero, as the reference does (gmpy.py:197-210)."""
        return self._wrap(self.ctx.inv(self._dev), self._shape)

    def sqrt(self, INV=False):
        """Modular (inverse) square root (finfields.py:1283-1290, :1424-1458 / :1550-1563): for
        p = 3 mod 4 a^((p+1)/4) (resp. a^((3p-5)/4)); for GF(2^n) a^(q/2) (resp. q/2 - 1)."""
        ops = _fops(type(self).field)
        if INV and self.size and not self._no_element_equals(0) and bool(self._zero_mask().any()):
            raise ZeroDivisionError('no inverse sqrt of 0')
        if ops.binary:
            e = (ops.order >> 1) - (1 if INV else 0)
        elif ops.modulus == 2:
            return self.copy()
        elif ops.modulus & 3 == 3:
            e = (ops.modulus * 3 - 5) >> 2 if INV else (ops.modulus + 1) >> 2
        else:
            # p = 1 mod 4: Cipolla-Lehmer on the device, then the reciprocal for INV (finfields.py:447-470)
            r = self._wrap(self.ctx.sqrt_cl(self._dev), self._shape)
            return r.reciprocal() if INV else r
        if e == 0:
            return self ** 0
        return self._wrap(self.ctx.pow(self._dev, e), self._shape)

    def is_sqr(self):
        """Quadratic residuosity (finfields.py:1292-1295, :1460-1470): a^((p-1)/2) != p-1."""
        ops = _fops(type(self).field)
        if ops.binary or ops.modulus == 2:
            return np.full(self._shape, True, dtype=bool)
        leg = self._wrap(self.ctx.pow(self._dev, (ops.modulus - 1) >> 1), self._shape)
        return leg != (ops.modulus - 1)

    def __truediv__(self, other):
        opd = self._operand(other)
        if opd is None:
            return NotImplemented
        kind, o = opd
        if kind == 'scalar':
            return self * _fops(type(self).field).inv(o)
        return self * o.reciprocal()

    def __rtruediv__(self, other):
        return self.reciprocal() * other

    def __lshift__(self, k):
        if not isinstance(k, (int, np.integer)):
            return NotImplemented
        ops = _fops(type(self).field)                                          # finfields.py:1227-1234
        return self * (ops.reduce_int(1 << int(k)) if ops.binary else pow(2, int(k), ops.modulus))

    def __rshift__(self, k):
        if not isinstance(k, (int, np.integer)):
            return NotImplemented
        ops = _fops(type(self).field)
        if ops.binary:
            return self * ops.inv(ops.reduce_int(1 << int(k)))
        return self * ops.inv(pow(2, int(k), ops.modulus))                     # :1250-1258

    # ---- small-matrix products (finfields.py:1126-1157) --------------------------------------
    def __rmatmul__(self, other):
        """A @ self for a small public/host matrix A (w, k) and self (k, n): the shape of the
        Vandermonde and Lagrange products in thresha (one pass over HBM, unreduced accumulation)."""
        if self.ndim != 2:
            return (other if isinstance(other, FieldArray) else type(self)(other)).__matmul__(self)
        F = type(self).field
        A = np.asarray(other.value if isinstance(other, FieldArray) else other, dtype=object)
        vec = A.ndim == 1
        A2 = A.reshape(1, -1) if vec else A
        if A2.ndim != 2 or A2.shape[1] != self._shape[0]:
            raise ValueError('matmul: shape mismatch')
        if A2.size > 4096 or isinstance(other, FieldArray):
            return (other if isinstance(other, FieldArray) else type(self)(other)).__matmul__(self)
        ops = _fops(F)
        lam = [ops.reduce_int(int(v) if isinstance(v, (int, np.integer)) else int(getattr(v, 'value', v)))
               for v in A2.reshape(-1)]
        k, n = self._shape
        rows = [self[j]._dev for j in range(k)]
        w = A2.shape[0]
        out = self.ctx.recombine(rows, lam, w=w)
        if w == 1:
            res = self._wrap(out, (n,))
            return res if vec else res.reshape(1, n)
        return _matrix_to_array(type(self), out)

    def __matmul__(self, other):
        """Matrix product over the field on the device (finfields.py:1126-1135): 1-D / 2-D operands
        with NumPy's matmul shape rules; the right operand may be anything the ctor accepts."""
        cls = type(self)
        if isinstance(other, HostView):
            other = other._fa
        if not isinstance(other, FieldArray):
            if not isinstance(other, (np.ndarray, list, tuple)):
                return NotImplemented                   # e.g. a secure array: its __rmatmul__ takes over
            other = cls(other)
        elif other.field is not cls.field:
            raise TypeError('arrays over different fields')
        if (self.ndim == 2 and other.ndim >= 3 and other._shape[-1] == 1 and other._shape[-2] == self._shape[1]
                and max(self._shape) <= 16):
            # small public matrix applied along the last axis of a batch: `A @ x[..., np.newaxis]`
            # (demos/np_aes.py:40) -- one streaming pass, the matrix travels as kernel arguments
            r, g = self._shape
            A = [int(v) for v in self._dev.to_ints()]
            out = self.ctx.group_matvec(other._dev, [A[i * g:(i + 1) * g] for i in range(r)])
            return self._wrap(out, other._shape[:-2] + (r, 1))
        if self.ndim == 0 or other.ndim == 0:
            raise ValueError('matmul: Input operand does not have enough dimensions')
        if self.ndim > 2 or other.ndim > 2:
            # stacks of matrices (NumPy's matmul broadcasting over the leading dimensions): one product per matrix
            A = self.reshape(1, -1) if self.ndim == 1 else self
            B = other.reshape(-1, 1) if other.ndim == 1 else other
            if A._shape[-1] != B._shape[-2]:
                raise ValueError(f'matmul: shapes {self._shape} and {other._shape} not aligned')
            batch = tuple(np.broadcast_shapes(A._shape[:-2], B._shape[:-2]))
            lb = (self.ctx.limbs,) if self.ctx.limbs else ()
            At = A._limb_view().expand(*batch, *A._shape[-2:], *lb).reshape(-1, *A._shape[-2:], *lb)
            Bt = B._limb_view().expand(*batch, *B._shape[-2:], *lb).reshape(-1, *B._shape[-2:], *lb)
            outs = [self._from_limb_view(At[i]) @ self._from_limb_view(Bt[i]) for i in range(At.shape[0])]
            mshape = (A._shape[-2], B._shape[-1])
            if not outs:
                res = cls(np.zeros(batch + mshape, dtype=object))
            else:
                res = _np_stack(outs, 0).reshape(batch + mshape)
            if self.ndim == 1:
                res = res.reshape(res._shape[:-2] + res._shape[-1:])
            elif other.ndim == 1:
                res = res.reshape(res._shape[:-1])
            return res
        M, K = (1, self._shape[0]) if self.ndim == 1 else self._shape
        K2, N = (other._shape[0], 1) if other.ndim == 1 else other._shape
        if K != K2:
            raise ValueError(f'matmul: shapes {self._shape} and {other._shape} not aligned')
        if self.ndim == 1 and other.ndim == 1:
            # inner product: two-stage reduction kernel (the local part of runtime.in_prod)
            return cls.field(self.ctx.dot(self._dev, other._dev).to_ints()[0])
        out = self.ctx.matmul(self._dev, other._dev, M, K, N)
        shape = (N,) if self.ndim == 1 else (M,) if other.ndim == 1 else (M, N)
        return self._wrap(out, shape)

    # ---- comparisons (finfields.py:1031-1043) --------------------------------------------------
    def _zero_mask(self):
        t = self._dev.t
        z = (t == 0)
        return z.all(dim=-1) if self.ctx.limbs else z

    def _no_element_equals(self, o: int) -> bool:
        """True if NO element equals the canonical scalar o, decided on the device by the lowest limb alone (a necessary
        condition for equality: one strided compare + one reduction instead of a compare over all limbs, a reduction over
        the limb axis, a mask-sized copy to the host).  False = unknown (some lowest limb matches): the caller compares in full."""
        t = self._dev.t
        if not self.ctx.limbs:
            if self.ctx.elem_bytes == 1 or t.dim() != 1:
                return False
            lim = 1 << (8 * self.ctx.elem_bytes)
            lo = o % lim
            return not bool((t == (lo - lim if lo >= lim >> 1 else lo)).any())
        bits = self._limb_bits()
        lo = o & ((1 << bits) - 1)
        if lo >= 1 << (bits - 1):
            lo -= 1 << bits                                  # the limb tensors are signed
        return not bool((t[..., 0] == lo).any())

    def __eq__(self, other):
        opd = self._operand(other)
        if opd is None:
            return NotImplemented
        kind, o = opd
        if kind == 'scalar':
            # large array against a scalar (np_random_bits: `_r2.value != 0` over f x n opened squares, runtime.py:4257):
            # usually no element matches -- then the answer is a uniform mask and nothing mask-sized is built or copied
            if self.size >= _UNIFORM_MASK_MIN and self._no_element_equals(o):
                return _UniformMask(self._shape, False)
            o = type(self)([o])
        a, b, shape = self._broadcast(o)
        eq = (a.t == b.t)
        if self.ctx.limbs:
            eq = eq.all(dim=-1)
        return eq.cpu().numpy().reshape(shape)

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else ~r

    __hash__ = None

    # ---- bit-level views of CANONICAL residues (for HostView's exact integer operations) --------------------
    def _limb_bits(self):
        return 32 if self.ctx.elem_bytes in (4, 12) else (8 if self.ctx.elem_bytes == 1 else 64)

    def _and_mask(self, mask: int) -> 'FieldArray':
        """element-wise x & mask on the limbs (x canonical, so is the result)"""
        W = self._limb_bits()
        t = self._dev.t
        signed = lambda v: v - (1 << W) if v >> (W - 1) else v
        if self.ctx.limbs:
            out = torch.empty_like(t)
            for q in range(self.ctx.limbs):
                out[..., q] = t[..., q] & signed((mask >> (W * q)) & ((1 << W) - 1))
        else:
            out = t & signed(mask & ((1 << W) - 1)) if W > 8 else t & (mask & 0xff)
        return self._wrap(DevArray(self.ctx, out, self._dev.n), self._shape)

    def _small_ints(self, dtype) -> np.ndarray:
        """the values as a NumPy integer array of `dtype` (caller knows they fit): low limb only"""
        t = self._dev.t
        low = t[..., 0] if self.ctx.limbs else t
        return low.cpu().numpy().astype(dtype).reshape(self._shape)

    def _bit_matrix(self, shifts: np.ndarray) -> np.ndarray:
        """(x >> shifts[j]) & 1 for every element and shift: int8 array of shape self.shape + shifts.shape"""
        W = self._limb_bits()
        t = self._dev.t
        ks = shifts.reshape(-1).astype(np.int64)
        nl = self.ctx.limbs or 1
        if ks.size == 0:
            return np.zeros(self._shape + tuple(shifts.shape), dtype=np.int8)
        tl = t if self.ctx.limbs else t.unsqueeze(-1)                       # (n, limbs)
        q = torch.from_numpy(np.minimum(ks // W, nl - 1)).to(t.device)       # limb of every shift ...
        b = torch.from_numpy(ks % W).to(device=t.device, dtype=tl.dtype)     # ... and the bit inside it
        live = torch.from_numpy((ks // W < nl)).to(t.device)                 # shifts beyond the element width give 0
        out = ((tl.index_select(-1, q) >> b) & 1).to(torch.int8) * live.to(torch.int8)    # three launches for any number of shifts
        return out.cpu().numpy().reshape(self._shape + tuple(shifts.shape))

    # ---- integer views (finfields.py:1375-1406) --------------------------------------------------
    def unsigned_(self):
        return np.array(self._dev.to_ints(), dtype=object).reshape(self._shape)

    def signed_(self):
        p = _fops(type(self).field).modulus
        return np.array([v - p if v > p >> 1 else v for v in self._dev.to_ints()], dtype=object).reshape(self._shape)

    @classmethod
    def intarray(cls, a):
        if _fops(cls.field).binary:
            return a.unsigned_()
        return a.signed_() if cls.field.is_signed else a.unsigned_()

    def sum(self, axis=None, keepdims=False, initial=None, **kw):
        """Sum of all elements as a field element / along axes (np.sum on a field array, finfields.py:1332-1337)."""
        if kw.get('where') is not None or kw.get('out') is not None:
            raise NotImplementedError('sum: where / out')
        if initial is not None:
            r = self.sum(axis=axis, keepdims=keepdims)
            return r + initial
        if isinstance(axis, (tuple, list)):
            axes = sorted((a if a >= 0 else a + self.ndim) for a in axis)
            r = self
            for a in reversed(axes):
                r = r.sum(axis=a)
            if keepdims:
                shape = tuple(1 if d in axes else s for d, s in enumerate(self._shape))
                r = (r if isinstance(r, FieldArray) else type(self)([r])).reshape(shape)
            return r
        if keepdims:
            r = self.sum(axis=axis)
            if axis is None:
                return type(self)([r]).reshape((1,) * self.ndim)
            axis = axis if axis >= 0 else axis + self.ndim
            r = r if isinstance(r, FieldArray) else type(self)([r])
            return r.reshape(self._shape[:axis] + (1,) + self._shape[axis + 1:])
        if axis is None or self.ndim <= 1:
            F = type(self).field
            if self.size == 0:
                return F(_fops(F).box(0))
            return F(_fops(F).box(self.ctx.sum(self._dev).to_ints()[0]))
        # reduce one axis: move it last ON THE DEVICE (a permuted copy of the limb tensor), then per row of k
        # elements: k <= 16 -> one thread per row (ffgpu_group_matvec with a row of ones); many rows -> matrix x
        # ones(k) (the HBM-bound matvec kernels); few long rows -> one two-stage reduction (ffgpu_sum) per row
        axis = axis if axis >= 0 else axis + self.ndim
        if axis == self.ndim - 1:
            moved = self
        else:
            perm = [d for d in range(self.ndim) if d != axis] + [axis]
            t = self._limb_view().permute(*(perm + ([self.ndim] if self.ctx.limbs else [])))
            moved = self._from_limb_view(t)
        k = moved.shape[-1]
        rows = moved.size // k if k else 0
        cls, ctx = type(self), self.ctx
        if k == 0:
            return cls(np.zeros(moved.shape[:-1], dtype=object))
        if rows == 0:
            return cls(np.zeros(moved.shape[:-1], dtype=object))
        flat = moved._dev
        if k <= 16:
            out = ctx.group_matvec(flat, [[1] * k])
        elif rows >= 64:
            ones = ctx.from_ints([1] * k)
            out = ctx.matmul(flat, ones, rows, k, 1)
        else:
            out = ctx.empty(rows)
            lv = flat.t.reshape(rows, k, ctx.limbs) if ctx.limbs else flat.t.reshape(rows, k)
            ov = out.t.reshape(rows, ctx.limbs) if ctx.limbs else out.t.reshape(rows)
            for r in range(rows):
                ov[r:r + 1].copy_(ctx.sum(DevArray(ctx, lv[r].reshape(-1, ctx.limbs) if ctx.limbs else lv[r], k)).t)
        return cls._wrap(out, moved.shape[:-1])

    def prod(self, axis=None, **kw):
        """Product of all elements / along one axis (finfields.py:1339-1349): log2(k) halving passes of the
        element-wise product kernel."""
        cls, ctx = type(self), self.ctx
        if kw.get('initial') is not None or kw.get('keepdims'):
            raise NotImplementedError('prod: initial / keepdims')
        if axis is not None and self.ndim > 1:
            axis = axis if axis >= 0 else axis + self.ndim
            perm = [axis] + [d for d in range(self.ndim) if d != axis]        # reduced axis FIRST: halves are contiguous
            t = self._limb_view().permute(*(perm + ([self.ndim] if ctx.limbs else [])))
            moved = self._from_limb_view(t)
            k, rest = moved._shape[0], moved._shape[1:]
            inner = int(np.prod(rest, dtype=np.int64))
            if k == 0:
                return cls(np.ones(rest, dtype=object))
            cur = moved._dev
        else:
            if self.size == 0:
                return cls.field(1)
            k, rest, inner, cur = self.size, (), 1, self._dev
        while k > 1:
            h = k // 2
            lo = DevArray(ctx, cur.t[:h * inner], h * inner)
            hi = DevArray(ctx, cur.t[h * inner:2 * h * inner], h * inner)
            nxt = ctx.empty((h + (k & 1)) * inner)
            ctx.mul(lo, hi, out=DevArray(ctx, nxt.t[:h * inner], h * inner))
            if k & 1:
                nxt.t[h * inner:].copy_(cur.t[2 * h * inner:k * inner])
            cur, k = nxt, h + (k & 1)
        if rest == () and (axis is None or self.ndim <= 1):
            return cls.field(_fops(cls.field).box(DevArray(ctx, cur.t[:1], 1).to_ints()[0]))
        return cls._wrap(DevArray(ctx, cur.t[:inner], inner), rest)

    def trace(self, offset=0, axis1=0, axis2=1, **kw):
        d = _np_diagonal(self, offset, axis1, axis2)
        return d.sum() if d.ndim == 1 else d.sum(axis=-1)

    # ---- linear algebra (finfields.py:872-978) --------------------------------------------------
    @classmethod
    def _eye(cls, n):
        ctx = _context(cls.field)
        e = ctx.empty(n * n)
        e.t.zero_()
        v = e.t.view(n, n, ctx.limbs)[..., 0] if ctx.limbs else e.t.view(n, n)
        v.fill_diagonal_(1)
        return cls._wrap(e, (n, n))

    @staticmethod
    def gauss_solve(A, B):
        """np.linalg.solve: Gauss-Jordan on (A | B) on the device (finfields.py:872-908)."""
        cls = type(A)
        n = A.shape[0] if A.ndim else 0
        if A.shape != (n, n):
            raise np.linalg.LinAlgError('array must be square')
        if not isinstance(B, FieldArray):
            B = cls(B)
        if B.ndim != 2 or B.shape[0] != n:
            raise ValueError('right-hand side must be a 2-D array with as many rows as A')
        if n == 0:
            return B.copy()
        aug = A._from_limb_view(torch.cat([A._limb_view(), B._limb_view()], dim=1))
        _, sing = A.ctx.gauss(aug._dev, n, n + B.shape[1], 1)
        if int(sing[0]):
            raise ZeroDivisionError('no inverse exists')
        return A._from_limb_view(aug._limb_view()[:, n:])

    @staticmethod
    def gauss_inv(A):
        """np.linalg.inv (finfields.py:910-916)."""
        return FieldArray.gauss_solve(A, type(A)._eye(len(A)))

    @staticmethod
    def gauss_det(a):
        """np.linalg.det over the last two dimensions (finfields.py:918-955): product of the pivots under
        the reference's pivot rule (it does not flip the sign on row swaps; neither does this)."""
        cls = type(a)
        if a.ndim < 2 or a.shape[-2] != a.shape[-1]:
            return np.linalg.det(np.empty(a.shape))          # NumPy's own error message
        n = a.shape[-1]
        batch = a.size // (n * n) if n else int(np.prod(a.shape[:-2], dtype=np.int64))
        if n == 0:
            d = cls(np.ones(a.shape[:-2], dtype=object))
            return cls.field(1) if a.ndim == 2 else d
        if batch == 0:
            return cls(np.empty(a.shape[:-2], dtype=object))
        d, _ = a.ctx.gauss(a.copy()._dev, n, n, batch, det=True)
        if a.ndim == 2:
            return cls.field(d.to_ints()[0])
        return cls._wrap(d, a.shape[:-2])

    @staticmethod
    def matrix_pow(A, n):
        """np.linalg.matrix_power, negative n through the inverse (finfields.py:957-978)."""
        cls = type(A)
        if n < 0:
            A, n = FieldArray.gauss_inv(A), -n
        C, D = cls._eye(len(A)), A
        for i in range(n.bit_length()):
            if (n >> i) & 1:
                C = C @ D
            if i + 1 < n.bit_length():
                D = D @ D
        return C

    # ---- NumPy protocol (finfields.py:728-819): data movement on the device limb tensors, arithmetic
    #      through the kernels; anything not listed raises instead of silently computing on the host ----
    # NB no __array__: like the reference's class, np.asarray(a) goes through the sequence protocol and yields an
    # object ndarray of field ELEMENTS (what np.vectorize / np.testing rely on); the raw values are `a.value`.

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        name = ufunc.__name__
        if method == 'at':
            # in-place ufunc.at(a, indices[, b]) (finfields.py:759-761): gather, apply, scatter on the device
            a, idx = inputs[0], inputs[1]
            sub = a[idx]
            res = getattr(np, name)(sub, *inputs[2:])
            a[idx] = res
            return None
        if method == 'reduce' and name in ('add', 'multiply'):
            a, axis = inputs[0], (inputs[1] if len(inputs) > 1 else kwargs.get('axis', 0))
            return a.sum(axis=axis) if name == 'add' else a.prod(axis=axis)
        if method == 'accumulate' and name in ('add', 'multiply'):
            a, axis = inputs[0], (inputs[1] if len(inputs) > 1 else kwargs.get('axis', 0))
            return _np_scan(a, axis, name == 'multiply')
        if method == 'outer' and name in ('add', 'subtract', 'multiply'):
            cls, a, b = _pair(inputs[0], inputs[1])
            return getattr(np, name)(a.reshape(tuple(a.shape) + (1,) * b.ndim), b)
        if method != '__call__' or kwargs.get('out') is not None:
            return NotImplemented
        a = inputs[0]
        b = inputs[1] if len(inputs) > 1 else None
        first = isinstance(a, FieldArray)
        binary = {'add': ('__add__', '__radd__'), 'subtract': ('__sub__', '__rsub__'),
                  'multiply': ('__mul__', '__rmul__'), 'divide': ('__truediv__', '__rtruediv__'),
                  'true_divide': ('__truediv__', '__rtruediv__'), 'matmul': ('__matmul__', '__rmatmul__'),
                  'equal': ('__eq__', '__eq__'), 'not_equal': ('__ne__', '__ne__'),
                  'left_shift': ('__lshift__', None), 'right_shift': ('__rshift__', None),
                  'power': ('__pow__', None)}
        if name in binary:
            fwd, rev = binary[name]
            if first:
                return getattr(a, fwd)(b)
            if rev is None:
                return NotImplemented
            return getattr(b, rev)(a)
        unary = {'negative': '__neg__', 'positive': '__pos__', 'reciprocal': 'reciprocal', 'sqrt': 'sqrt'}
        if name in unary:
            return getattr(a, unary[name])()
        if name == 'square':                          # the reference: value ** 2, reduced (finfields.py:755-764)
            return a * a
        if name in ('absolute', 'fabs'):              # of the canonical (non-negative) representatives: the array itself
            return a.copy()
        return NotImplemented

    def __array_function__(self, func, types, args, kwargs):
        # `self` is a field ELEMENT when the reference's element class redirects here (finfields.py:83-85)
        cls = type(self) if isinstance(self, FieldArray) else type(self).array
        name = func.__name__
        args = _lift_elements(cls, args)
        kwargs = {k: _lift_elements(cls, v) for k, v in kwargs.items()}
        impl = _ARRAY_FUNCTIONS.get(name)
        if impl is None and name in _MOVEMENT_FUNCTIONS:
            return _np_movement(func, args, kwargs)
        if impl is None:
            raise NotImplementedError(f'numpy.{name} is not supported on GPU field arrays')
        return impl(*args, **kwargs)

    # ---- in-place variants and reflected stubs of the reference (finfields.py:1148-1157, 1170-1271) ----
    def _assign(self, res: 'FieldArray'):
        if res._shape != self._shape:
            raise ValueError('non-broadcastable output operand')
        self._flush_products_reading(self._dev)
        self._dev.t.copy_(res._dev.t.reshape(self._dev.t.shape))       # views of this array see the update
        self._cache = None
        return self

    def __itruediv__(self, other):
        r = self.__truediv__(other)
        return r if r is NotImplemented else self._assign(r)

    def __ilshift__(self, other):
        r = self.__lshift__(other)
        return r if r is NotImplemented else self._assign(r)

    def __irshift__(self, other):
        r = self.__rshift__(other)
        return r if r is NotImplemented else self._assign(r)

    def __ipow__(self, other):
        r = self.__pow__(other)
        return r if r is NotImplemented else self._assign(r)

    def __imatmul__(self, other):
        r = self.__matmul__(other)
        if r is NotImplemented:
            return r
        if isinstance(r, FieldArray) and r._shape == self._shape:
            return self._assign(r)
        raise ValueError('in-place matmul needs a result of the same shape')

    def __rpow__(self, other):
        return NotImplemented

    def __rlshift__(self, other):
        return NotImplemented

    def __rrshift__(self, other):
        return NotImplemented

    def __int__(self):
        """(signed) integer value of a size-1 array (finfields.py:1385-1392)."""
        return self.intarray(self).__int__()

    def __abs__(self):
        return abs(self.signed_())                                     # finfields.py:1394-1396

    # ---- class-level primitives on RAW values (finfields.py:1408-1470, 1550-1563): the runtime calls e.g.
    #      field.array._sqrt(r2, INV=True) on `.value` arrays (runtime.py:4265); computed on the device, returned
    #      as the object ndarray (or int) the reference returns ----
    @classmethod
    def _raw(cls, a):
        if isinstance(a, HostView):
            return a._fa, ('lazy' if a._is_lazy else True)   # (a derived view's residues are the field elements it stands for)
        if isinstance(a, FieldArray):
            return a, True
        if isinstance(a, np.ndarray):
            return cls(a, check=False), True
        return cls([int(a)], check=False), False

    @staticmethod
    def _unraw(r, is_array):
        if is_array == 'lazy':
            return HostView(r, lazy=True)                # the argument was a device-resident view: so is the result
        if is_array:
            return np.array(r._host_value())             # writable copy, as the reference returns a fresh array
        return r._host_value().reshape(-1)[0]

    @classmethod
    def _pow(cls, a, b):
        A, arr = cls._raw(a)
        if isinstance(b, np.ndarray) or isinstance(b, (list, tuple)):
            b = np.asarray(b)
            shape = np.broadcast_shapes(A._shape, b.shape)
            vals = np.broadcast_to(b, shape).reshape(-1)
            Ab = A._from_limb_view(A._limb_view().expand(*shape, *((A.ctx.limbs,) if A.ctx.limbs else ())))
            out = Ab.copy()
            for e in set(int(v) for v in vals):       # one kernel per distinct public exponent
                mask = np.array([int(v) == e for v in vals]).reshape(shape)
                out[mask] = Ab[mask] ** e
            return cls._unraw(out, True)
        return cls._unraw(A ** int(b), arr)

    @classmethod
    def _reciprocal(cls, a):
        A, arr = cls._raw(a)
        return cls._unraw(A.reciprocal(), arr)

    @classmethod
    def _sqrt(cls, a, INV=False):
        A, arr = cls._raw(a)
        return cls._unraw(A.sqrt(INV=INV), arr)

    @classmethod
    def _is_sqr(cls, a):
        A, arr = cls._raw(a)
        r = A.is_sqr()
        return r if arr else bool(r.reshape(-1)[0])

    # ---- ndarray-style methods of the reference class (finfields.py:1308-1368) on the device ----
    def compress(self, condition, axis=None):
        return _np_movement(np.compress, (condition, self), {'axis': axis})

    def nonzero(self):
        return np.nonzero(_np_nonzero_mask(self))

    def take(self, indices, axis=None, **kw):
        return _np_take(self, indices, axis)

    def repeat(self, repeats, axis=None):
        return _np_repeat(self, repeats, axis)

    def diagonal(self, offset=0, axis1=0, axis2=1):
        return _np_diagonal(self, offset, axis1, axis2)

    def swapaxes(self, axis1, axis2):
        return _ARRAY_FUNCTIONS['swapaxes'](self, axis1, axis2)

    @staticmethod
    def diag(a, k=0):
        return _np_movement(np.diag, (a, k), {})

    @property
    def T(self):
        return self.transpose()

    def transpose(self, *axes):
        if len(axes) == 1 and (axes[0] is None or isinstance(axes[0], (tuple, list))):
            axes = tuple(axes[0] or ())
        if not axes:
            axes = tuple(reversed(range(self.ndim)))
        axes = tuple(a if a >= 0 else a + self.ndim for a in axes)
        t = self._limb_view()
        perm = tuple(axes) + ((self.ndim,) if self.ctx.limbs else ())
        return self._from_limb_view(t.permute(*perm))

    def _from_limb_view(self, t):
        """limb tensor (array shape [+ trailing 2]) -> FieldArray with contiguous device data"""
        lb = self.ctx.limbs
        shape = tuple(t.shape[:-1]) if lb else tuple(t.shape)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        flat = t.contiguous().view(n, lb) if lb else t.contiguous().view(n)
        return self._wrap(DevArray(self.ctx, flat, n), shape)

    def tolist(self):
        """(nested) list of field elements (finfields.py:1320-1321)."""
        return np.vectorize(type(self).field, otypes='O')(self._host_value()).tolist()

    def __reduce__(self):
        """pickle.dumps(row) -- how the runtime marshals arrays (runtime.py:484,571,655) -- ships the
        field.to_bytes-format limb bytes from the pinned staging buffer instead of a graph of PyLongs; the receiving
        party rebuilds a device array (no Python integers on either side)."""
        F = type(self).field
        key = _field_key(F, _fops(F))
        from . import ipcwire
        if ipcwire.ENABLED:
            ctx = self.ctx
            ipcwire.ensure_runtime_hooks()
            if ipcwire.want_descriptor(ctx, self.size * ctx.elem_bytes):
                dev = self._dev
                if dev.t.is_cuda and dev.t.is_contiguous():
                    return _array_from_ipc, (key, self._shape, self.size, ipcwire.export(ctx, dev.t))
            ipcwire.stats['inline'] += 1
        return _array_from_wire, (key, self._shape, self.to_wire())

    # ---- wire format (finfields.py:91-102) straight from device limbs ------------------------------
    def to_wire(self) -> bytes:
        """field.to_bytes(self.value) without boxing Python ints when byte_length == limb width."""
        F, eb = type(self).field, self.ctx.elem_bytes
        raw = self.ctx.download_bytes(self._dev.t)          # pinned staging buffer: PCIe speed
        if F.byte_length == eb:
            return raw.tobytes()
        n = self.size
        b = raw.reshape(n, eb)
        if F.byte_length < eb:
            return np.ascontiguousarray(b[:, :F.byte_length]).tobytes()
        pad = np.zeros((n, F.byte_length - eb), dtype=np.uint8)                # e.g. GF(2^8): 2-byte wire elements
        return np.concatenate([b, pad], axis=1).tobytes()

    @classmethod
    def from_wire(cls, data: bytes, shape=None, check=True) -> 'FieldArray':
        """field.from_bytes (finfields.py:97-102) straight into device limbs.  The length must be a whole number of
        byte_length-sized elements (and match `shape`); with check=True (default: the bytes come from a peer) the
        values are reduced on the device like `field.array(...)` does (finfields.py:724), so that kernels never
        see a non-canonical element."""
        F = cls.field
        ctx = _context(F)
        eb, r = ctx.elem_bytes, F.byte_length
        if len(data) % r:
            raise ValueError(f'{len(data)} bytes is not a whole number of {r}-byte field elements')
        n = len(data) // r
        if shape is not None and int(np.prod(shape, dtype=np.int64)) != n:
            raise ValueError(f'{n} elements on the wire do not fill shape {tuple(shape)}')
        if r == eb and n:
            # the usual case (64-, 96-, 128-, 192-bit fields): the wire bytes ARE the limb layout -- one copy into the
            # pinned staging buffer, one transfer, no NumPy intermediates
            from .engine import _torch_dtype
            t = ctx.upload_bytes(data, _torch_dtype(eb), (n, ctx.limbs) if ctx.limbs else (n,))
            dev = DevArray(ctx, t, n)
            if check:
                dev = ctx.reduce(dev, out=dev)
            return cls._wrap(dev, shape if shape is not None else (n,))
        b = np.frombuffer(data, dtype=np.uint8).reshape(n, r)
        if r < eb:
            b = np.concatenate([b, np.zeros((n, eb - r), dtype=np.uint8)], axis=1)
        elif r > eb:
            if check and n and b[:, eb:].any():
                raise ValueError('field element on the wire exceeds the element width')
            b = b[:, :eb]
        raw = np.ascontiguousarray(b).view({1: np.uint8, 4: np.uint32, 8: np.uint64, 12: np.uint32, 16: np.uint64, 24: np.uint64}[eb])
        raw = raw.reshape(n, ctx.limbs) if ctx.limbs else raw.reshape(n)
        dev = ctx.from_numpy(raw)
        if check and n:
            dev = ctx.reduce(dev, out=dev)
        return cls._wrap(dev, shape if shape is not None else (n,))

    def __repr__(self):
        return f'{self.intarray(self)}'


_STORAGE_SLOTS = ('_devv', '_shape', '_cache', '_lazy', '__weakref__')


class _MirrorFieldArray(FieldArray):
    """Concrete base of the array types this module's GF() creates."""
    __slots__ = _STORAGE_SLOTS


def _matrix_to_array(cls, mtx: DevMatrix) -> FieldArray:
    """(rows, n) DevMatrix with padded pitch -> contiguous (rows, n) FieldArray."""
    ctx = mtx.ctx
    if ctx.limbs:
        t = mtx.t[:, :mtx.n, :].contiguous().view(-1, ctx.limbs)
    else:
        t = mtx.t[:, :mtx.n].contiguous().view(-1)
    return cls._wrap(DevArray(ctx, t, mtx.rows * mtx.n), (mtx.rows, mtx.n))


# ---- numpy functions routed through __array_function__ ----------------------------------------------
def _np_concatenate(arrays, axis=0):
    arrays = list(arrays)
    cls = type(next(a for a in arrays if isinstance(a, FieldArray)))
    arrays = [_as_arr(cls, a) for a in arrays]
    if axis is None:
        arrays, axis = [a.reshape(-1) for a in arrays], 0
    t = torch.cat([a._limb_view() for a in arrays], dim=axis if axis >= 0 else axis + arrays[0].ndim)
    return arrays[0]._from_limb_view(t)


def _np_stack(arrays, axis=0):
    arrays = list(arrays)
    cls = type(next(a for a in arrays if isinstance(a, FieldArray)))
    arrays = [_as_arr(cls, a) for a in arrays]
    nd = arrays[0].ndim + 1
    t = torch.stack([a._limb_view() for a in arrays], dim=axis if axis >= 0 else axis + nd)
    return arrays[0]._from_limb_view(t)


def _np_roll(a, shift, axis=None):
    if axis is None:
        return _np_roll(a.reshape(-1), shift, 0).reshape(a.shape)
    return a._from_limb_view(torch.roll(a._limb_view(), shift, dims=axis if axis >= 0 else axis + a.ndim))


def _np_flip(a, axis=None):
    dims = tuple(range(a.ndim)) if axis is None else ((axis if axis >= 0 else axis + a.ndim),)
    return a._from_limb_view(torch.flip(a._limb_view(), dims=dims))



# ---- generic data-movement functions ------------------------------------------------------------------
# The reference lets NumPy run any function on the object array and re-wraps the result
# (finfields.py:766-819).  For functions that only MOVE elements (no arithmetic) the same generality comes
# from running the NumPy function on int64 INDEX arrays on the host (index 0 = the zero element, so tril /
# diag / pad work) and gathering on the device with the resulting index array.
_MOVEMENT_FUNCTIONS = frozenset((
    'reshape', 'ravel', 'transpose', 'swapaxes', 'moveaxis', 'rollaxis', 'squeeze', 'expand_dims',
    'atleast_1d', 'atleast_2d', 'atleast_3d', 'concatenate', 'stack', 'vstack', 'hstack', 'dstack',
    'column_stack', 'block', 'split', 'array_split', 'vsplit', 'hsplit', 'dsplit', 'tile', 'repeat', 'flip',
    'fliplr', 'flipud', 'roll', 'rot90', 'take', 'take_along_axis', 'diag', 'diagonal', 'diagflat', 'tril',
    'triu', 'delete', 'append', 'broadcast_to', 'compress', 'pad', 'where', 'copy', 'resize', 'select', 'choose'))


def _np_movement(func, args, kwargs, keep=()):
    name = func.__name__
    if name == 'pad' and ('constant_values' in kwargs or 'end_values' in kwargs or len(args) > 3):
        raise NotImplementedError('np.pad on GPU field arrays: only zero / element-copy padding')
    pool, state = [], {'offset': 1, 'cls': None}

    def find_cls(x):
        if isinstance(x, FieldArray):
            return type(x)
        if isinstance(x, (list, tuple)):
            for y in x:
                c = find_cls(y)
                if c is not None:
                    return c
        return None

    cls = find_cls(args) or find_cls(tuple(kwargs.values()))

    def index_of(x):
        idx = np.arange(state['offset'], state['offset'] + x.size, dtype=np.int64).reshape(x.shape)
        pool.append(x)
        state['offset'] += x.size
        return idx

    def sub(x, inside=False):
        if isinstance(x, FieldArray):
            if x.field is not cls.field:
                raise TypeError('arrays over different fields')
            return index_of(x)
        if isinstance(x, (list, tuple)) and find_cls(x) is not None:
            return type(x)(sub(y, True) for y in x)
        if inside and not isinstance(x, FieldArray):
            return index_of(_as_arr(cls, x))              # plain ints / lists next to field arrays
        return x

    args = list(args)
    if name == 'append' and len(args) > 1 and not isinstance(args[1], FieldArray):
        args[1] = cls(args[1])
    if name == 'where' and len(args) == 3:
        args[1:] = [a if isinstance(a, FieldArray) else cls(a) for a in args[1:]]
    iargs = [a if i in keep else sub(a) for i, a in enumerate(args)]
    ikw = {k: sub(v) for k, v in kwargs.items()}
    res = func(*iargs, **ikw)
    ctx = pool[0].ctx
    eb = ctx.elem_bytes
    lb = ctx.limbs
    flats = [p._dev.t.reshape(-1, lb) if lb else p._dev.t.reshape(-1) for p in pool]
    zero = torch.zeros((1, lb) if lb else (1,), dtype=flats[0].dtype, device=flats[0].device)
    table = torch.cat([zero] + flats)

    def back(r):
        if isinstance(r, np.ndarray):
            idx = torch.from_numpy(np.array(r, dtype=np.int64).reshape(-1)).to(table.device)
            t = table.index_select(0, idx)
            return cls._wrap(DevArray(ctx, t, idx.shape[0]), tuple(r.shape))
        if isinstance(r, (list, tuple)):
            return type(r)(back(y) for y in r)
        if isinstance(r, (int, np.integer)):
            return cls.field(DevArray(ctx, table[int(r):int(r) + 1], 1).to_ints()[0])
        return r

    return back(res)


def _np_outer(a, b):
    cls = type(a) if isinstance(a, FieldArray) else type(b)
    a = a if isinstance(a, FieldArray) else cls(a)
    b = b if isinstance(b, FieldArray) else cls(b)
    return a.reshape(-1, 1) * b.reshape(1, -1)


# ---- ring arithmetic that NumPy composes from +, -, * on object arrays in the reference (its generic __array_function__
#      path, finfields.py:766-819); here the same compositions over the device operators ---------------------------------
def _np_diff(a, n=1, axis=-1, prepend=np._NoValue, append=np._NoValue):
    if n < 0:
        raise ValueError(f'order must be non-negative but got {n}')
    cls = type(a)
    if a.ndim == 0:
        raise ValueError('diff requires input that is at least one dimensional')
    axis = axis if axis >= 0 else axis + a.ndim
    parts = []
    for extra in (prepend, None, append):
        if extra is None:
            parts.append(a)
        elif extra is not np._NoValue:
            e = _as_arr(cls, extra)
            if e.ndim == 0:
                shp = list(a.shape)
                shp[axis] = 1
                e = _np_movement(np.broadcast_to, (e, tuple(shp)), {})
            parts.append(e)
    if len(parts) > 1:
        a = _np_concatenate(parts, axis)
    hi = [slice(None)] * a.ndim
    lo = [slice(None)] * a.ndim
    hi[axis], lo[axis] = slice(1, None), slice(None, -1)
    for _ in range(n):
        a = a[tuple(hi)] - a[tuple(lo)]
    return a


def _np_ediff1d(ary, to_end=None, to_begin=None):
    cls = type(ary)
    d = _np_diff(ary.reshape(-1))
    parts = ([_as_arr(cls, to_begin).reshape(-1)] if to_begin is not None else []) + [d] + \
            ([_as_arr(cls, to_end).reshape(-1)] if to_end is not None else [])
    return d if len(parts) == 1 else _np_concatenate(parts, 0)


def _np_cross(a, b, axisa=-1, axisb=-1, axisc=-1, axis=None):
    cls, a, b = _pair(a, b)
    if axis is not None:
        axisa = axisb = axisc = axis
    a = _ARRAY_FUNCTIONS['moveaxis'](a, axisa, -1) if a.ndim > 1 else a
    b = _ARRAY_FUNCTIONS['moveaxis'](b, axisb, -1) if b.ndim > 1 else b
    if a.shape[-1] != 3 or b.shape[-1] != 3:
        raise NotImplementedError('np.cross on GPU field arrays: vectors of dimension 3')
    a0, a1, a2 = a[..., 0:1], a[..., 1:2], a[..., 2:3]           # (slices keep the axis: 1-D inputs stay arrays)
    b0, b1, b2 = b[..., 0:1], b[..., 1:2], b[..., 2:3]
    c = _np_concatenate([a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0], -1)
    return _ARRAY_FUNCTIONS['moveaxis'](c, -1, axisc) if c.ndim > 1 and axisc not in (-1, c.ndim - 1) else c

def _np_polyval(p, x):
    """Horner, as numpy.polyval: p[0] is the leading coefficient."""
    cls = type(p) if isinstance(p, FieldArray) else type(x)
    p, x = _as_arr(cls, p), _as_arr(cls, x)
    if p.ndim != 1:
        raise ValueError('polyval: 1-D coefficient array required')
    y = x * 0
    for k in range(len(p)):
        y = y * x + p[k]
    return y


def _np_polyadd(a1, a2, sign=1):
    cls, a1, a2 = _pair(a1, a2)
    a1, a2 = a1.reshape(-1), a2.reshape(-1)
    d = len(a1) - len(a2)
    if d > 0:
        a2 = _np_concatenate([cls(np.zeros(d, dtype=object)), a2], 0)
    elif d < 0:
        a1 = _np_concatenate([cls(np.zeros(-d, dtype=object)), a1], 0)
    return a1 + a2 if sign > 0 else a1 - a2


def _np_polymul(a1, a2):
    cls, a1, a2 = _pair(a1, a2)
    return _np_convolve(a1.reshape(-1), a2.reshape(-1))


def _np_multi_dot(arrays, *, out=None):
    arrays = list(arrays)
    if len(arrays) < 2:
        raise ValueError('Expecting at least two arrays.')
    r = arrays[0]
    for m_ in arrays[1:]:                # (left to right: the result is the same in a field; no cost-based ordering)
        r = _np_dot(r, m_)
    return r


def _np_array_equiv(a1, a2):
    cls, a1, a2 = _pair(a1, a2)
    try:
        return bool((a1 == a2).all())
    except ValueError:
        return False


def _np_convolve(a, v, mode='full'):
    """np.convolve over the field (runtime.np_convolve's local part, runtime.py:2580): Toeplitz gather of
    the longer operand (index 0 = zero element) times the shorter one through the product kernel."""
    cls = type(a) if isinstance(a, FieldArray) else type(v)
    a = a if isinstance(a, FieldArray) else cls(a)
    v = v if isinstance(v, FieldArray) else cls(v)
    if a.ndim != 1 or v.ndim != 1 or a.size == 0 or v.size == 0:
        raise ValueError('convolve: 1-D non-empty arrays required')
    if a.size < v.size:
        a, v = v, a
    na, nv = a.size, v.size
    ctx, eb = a.ctx, a.ctx.elem_bytes
    lb = ctx.limbs
    flat = a._dev.t.reshape(-1, lb) if lb else a._dev.t.reshape(-1)
    dev = flat.device
    k = torch.arange(na + nv - 1, device=dev).unsqueeze(1) - torch.arange(nv, device=dev).unsqueeze(0)    # T[k][j] = a[k - j]
    idx = torch.where((k >= 0) & (k < na), k + 1, torch.zeros_like(k))                                    # 0 = the zero element
    zero = torch.zeros((1, lb) if lb else (1,), dtype=flat.dtype, device=dev)
    t = torch.cat([zero, flat]).index_select(0, idx.reshape(-1))
    T = cls._wrap(DevArray(ctx, t, idx.numel()), tuple(idx.shape))
    full = T @ v
    if mode == 'full':
        return full
    if mode == 'same':
        lo = (nv - 1) // 2
        return full[lo:lo + na]
    if mode == 'valid':
        return full[nv - 1:na]
    raise ValueError("mode must be 'full', 'same' or 'valid'")


def _np_nonzero_mask(a):
    return (~a._zero_mask()).cpu().numpy().reshape(a.shape)



# ---- data movement with a torch equivalent: done on the device limb tensors (the generic index-gather path
#      above costs a host index array of 8 bytes per element) -------------------------------------------------
def _limb_dims(a, t_fn):
    """apply t_fn to the limb view, which has one trailing limb axis for multi-limb fields"""
    return a._from_limb_view(t_fn(a._limb_view()))


def _np_where(cond, x=None, y=None):
    if x is None or y is None:
        raise NotImplementedError('np.where(condition) alone: use np.nonzero')
    cls = type(x) if isinstance(x, FieldArray) else type(y)
    x = x if isinstance(x, FieldArray) else cls(x)
    y = y if isinstance(y, FieldArray) else cls(y)
    if isinstance(cond, FieldArray):
        cond = _np_nonzero_mask(cond)
    shape = np.broadcast_shapes(np.shape(cond), x.shape, y.shape)
    c = torch.from_numpy(np.array(np.broadcast_to(np.asarray(cond, dtype=bool), shape))).to(x._dev.t.device)   # a copy: writable
    lb = x.ctx.limbs
    tx = x._limb_view().expand(*shape, *((lb,) if lb else ()))
    ty = y._limb_view().expand(*shape, *((lb,) if lb else ()))
    return x._from_limb_view(torch.where(c.unsqueeze(-1) if lb else c, tx, ty))


def _np_tile(a, reps):
    reps = (reps,) if isinstance(reps, (int, np.integer)) else tuple(reps)
    nd = max(a.ndim, len(reps))
    reps = (1,) * (nd - len(reps)) + reps
    t = a._limb_view().reshape((1,) * (nd - a.ndim) + tuple(a.shape) + ((a.ctx.limbs,) if a.ctx.limbs else ()))
    return a._from_limb_view(t.repeat(*reps, *((1,) if a.ctx.limbs else ())))


def _np_repeat(a, repeats, axis=None):
    if not isinstance(repeats, (int, np.integer)):
        return _np_movement(np.repeat, (a, repeats), {'axis': axis})
    if axis is None:
        a, axis = a.reshape(-1), 0
    axis = axis if axis >= 0 else axis + a.ndim
    return _limb_dims(a, lambda t: torch.repeat_interleave(t, int(repeats), dim=axis))


def _np_take(a, indices, axis=None):
    idx = torch.from_numpy(np.ascontiguousarray(np.asarray(indices, dtype=np.int64))).to(a._dev.t.device)
    if axis is None:
        src, axis, out_shape = a.reshape(-1), 0, tuple(idx.shape)
    else:
        axis = axis if axis >= 0 else axis + a.ndim
        src, out_shape = a, None
    n_ax = src.shape[axis]
    flat_idx = torch.where(idx < 0, idx + n_ax, idx).reshape(-1)
    r = _limb_dims(src, lambda t: torch.index_select(t, axis, flat_idx))
    if out_shape is not None:
        return r.reshape(out_shape) if out_shape else r.reshape(())
    return r.reshape(src.shape[:axis] + tuple(idx.shape) + src.shape[axis + 1:])


def _np_diagonal(a, offset=0, axis1=0, axis2=1):
    return _limb_dims(a, lambda t: torch.diagonal(t, offset, axis1, axis2).movedim(-2, -1) if a.ctx.limbs
                      else torch.diagonal(t, offset, axis1, axis2))


def _np_tri(a, k, upper):
    if a.ndim < 2:
        return _np_movement(np.triu if upper else np.tril, (a, k), {})
    r, c = a.shape[-2], a.shape[-1]
    mask = (torch.triu if upper else torch.tril)(torch.ones(r, c, dtype=torch.bool, device=a._dev.t.device), diagonal=k)
    lb = a.ctx.limbs
    t = a._limb_view()
    return a._from_limb_view(torch.where(mask.unsqueeze(-1) if lb else mask, t, torch.zeros_like(t)))


def _np_axes(a, fn):
    return _limb_dims(a, fn)

def _lift_elements(cls, x):
    """field elements among the arguments of a NumPy function -> 0-d arrays (the reference unwraps them to raw
    values, finfields.py:778-797)."""
    if isinstance(x, cls.field):
        return cls([int(x.value)], check=False).reshape(())
    if isinstance(x, HostView):
        return x._fa
    if isinstance(x, (list, tuple)) and not isinstance(x, FieldArray):
        y = [_lift_elements(cls, v) for v in x]
        if any(a is not b for a, b in zip(x, y)):
            return type(x)(y) if type(x) in (list, tuple) else y
    return x


def _as_arr(cls, x):
    if isinstance(x, FieldArray):
        return x
    if isinstance(x, HostView):
        return x._fa
    if isinstance(x, np.ndarray) and x.dtype.kind == 'f':
        # a float ndarray combined with field arrays (tests/test_runtime.py:475-478): the reference lets NumPy mix
        # the float objects into the share array (meaningless values, only the metadata of the secure array is
        # used afterwards); limbs cannot hold floats, so they are truncated
        x = x.astype(np.int64)
    return cls(x)


def _pair(a, b):
    cls = type(a) if isinstance(a, FieldArray) else type(b)
    return cls, _as_arr(cls, a), _as_arr(cls, b)


def _scalar_out(r):
    """0-d result -> field element, as NumPy returns scalars"""
    if isinstance(r, FieldArray) and r.ndim == 0:
        return r[()]
    return r


def _np_dot(a, b):
    cls, a, b = _pair(a, b)
    if a.ndim == 0 or b.ndim == 0:
        return a * b
    if a.ndim <= 2 and b.ndim <= 2:
        return a @ b
    return _np_tensordot(a, b, axes=((a.ndim - 1,), (b.ndim - 2,)))


def _np_vdot(a, b):
    cls, a, b = _pair(a, b)
    return a.reshape(-1) @ b.reshape(-1)


def _np_tensordot(a, b, axes=2):
    cls, a, b = _pair(a, b)
    if isinstance(axes, (int, np.integer)):
        ax_a, ax_b = list(range(a.ndim - axes, a.ndim)), list(range(axes))
    else:
        ax_a, ax_b = axes
        ax_a = [ax_a] if isinstance(ax_a, (int, np.integer)) else list(ax_a)
        ax_b = [ax_b] if isinstance(ax_b, (int, np.integer)) else list(ax_b)
    ax_a = [x if x >= 0 else x + a.ndim for x in ax_a]
    ax_b = [x if x >= 0 else x + b.ndim for x in ax_b]
    if [a.shape[x] for x in ax_a] != [b.shape[x] for x in ax_b]:
        raise ValueError('shape-mismatch for sum')
    free_a = [d for d in range(a.ndim) if d not in ax_a]
    free_b = [d for d in range(b.ndim) if d not in ax_b]
    K = int(np.prod([a.shape[x] for x in ax_a], dtype=np.int64))
    M = int(np.prod([a.shape[x] for x in free_a], dtype=np.int64))
    N = int(np.prod([b.shape[x] for x in free_b], dtype=np.int64))
    A2 = a.transpose(*(free_a + ax_a)).reshape(M, K)
    B2 = b.transpose(*(ax_b + free_b)).reshape(K, N)
    out_shape = tuple(a.shape[x] for x in free_a) + tuple(b.shape[x] for x in free_b)
    if K == 0:
        return cls(np.zeros(out_shape, dtype=object))
    return (A2 @ B2).reshape(out_shape)


def _np_inner(a, b):
    cls, a, b = _pair(a, b)
    if a.ndim == 0 or b.ndim == 0:
        return a * b
    return _scalar_out(_np_tensordot(a, b, axes=((a.ndim - 1,), (b.ndim - 1,))))


def _np_kron(a, b):
    cls, a, b = _pair(a, b)
    nd = max(a.ndim, b.ndim, 1)
    sa = (1,) * (nd - a.ndim) + tuple(a.shape)
    sb = (1,) * (nd - b.ndim) + tuple(b.shape)
    A = a.reshape(tuple(x for s_ in sa for x in (s_, 1)))
    B = b.reshape(tuple(x for s_ in sb for x in (1, s_)))
    return (A * B).reshape(tuple(x * y for x, y in zip(sa, sb)))


def _np_vander(x, N=None, increasing=False):
    """np.vander on a 1-D field array: one exponentiation kernel per column (runtime.py:4931)."""
    if x.ndim != 1:
        raise ValueError('x must be a one-dimensional array or sequence.')
    N = len(x) if N is None else N
    if N == 0:
        return type(x)(np.empty((len(x), 0), dtype=object))
    cols = [x ** e for e in range(N)]
    if not increasing:
        cols.reverse()
    return _np_stack(cols, axis=1)


def _np_insert(arr, obj, values, axis=None):
    cls = type(arr) if isinstance(arr, FieldArray) else type(values)
    return _np_movement(np.insert, (_as_arr(cls, arr), obj, _as_arr(cls, values)), {'axis': axis}, keep=(1,))


def _np_trim_zeros(filt, trim='fb', **kw):
    nz = np.flatnonzero(_np_nonzero_mask(filt))
    if nz.size == 0:
        return filt[:0]
    lo = int(nz[0]) if 'f' in trim.lower() else 0
    hi = int(nz[-1]) + 1 if 'b' in trim.lower() else len(filt)
    return filt[lo:hi]


def _np_scan(a, axis, mul: bool, include_initial=False):
    """np.cumsum / np.cumprod along one axis: Hillis-Steele scan, ceil(log2 k) passes of the add / mul kernel."""
    if axis is None:
        a, axis = a.reshape(-1), 0
    axis = axis if axis >= 0 else axis + a.ndim
    if include_initial:                                   # np.cumulative_sum/prod: leading 0 / 1 along the axis
        init = type(a)(np.full(a.shape[:axis] + (1,) + a.shape[axis + 1:], 1 if mul else 0, dtype=object))
        return _np_concatenate([init, _np_scan(a, axis, mul)], axis)
    perm = [axis] + [d for d in range(a.ndim) if d != axis]
    inv = [perm.index(d) for d in range(a.ndim)]
    moved = a.transpose(*perm) if a.ndim > 1 else a.copy()
    ctx = a.ctx
    k = moved.shape[0] if moved.ndim else 1
    inner = moved.size // k if k else 0
    cur = moved._dev if _storage_id(moved._dev) != _storage_id(a._dev) else moved._dev.clone()
    step = 1
    while step < k and inner:
        nxt = cur.clone()
        m = (k - step) * inner
        lo = DevArray(ctx, cur.t[:m], m)
        hi = DevArray(ctx, cur.t[step * inner:k * inner], m)
        (ctx.mul if mul else ctx.add)(lo, hi, out=DevArray(ctx, nxt.t[step * inner:k * inner], m))
        cur, step = nxt, step * 2
    res = a._wrap(cur, moved.shape)
    return res.transpose(*inv) if a.ndim > 1 else res


_ARRAY_FUNCTIONS = {
    'shape': lambda a: a.shape, 'ndim': lambda a: a.ndim, 'size': lambda a: a.size,
    'reshape': lambda a, *shape, order='C', **kw: a.reshape(*shape if shape else (kw.get('newshape', kw.get('shape')),), order=order or 'C'),
    'ravel': lambda a: a.ravel(), 'copy': lambda a: a.copy(),
    'transpose': lambda a, axes=None: a.transpose(*([axes] if axes is not None else [])),
    'concatenate': _np_concatenate, 'stack': _np_stack,
    'vstack': lambda arrays: _np_concatenate([a.reshape(1, -1) if a.ndim == 1 else a for a in arrays], 0),
    'hstack': lambda arrays: _np_concatenate(list(arrays), 0 if list(arrays)[0].ndim == 1 else 1),
    'roll': _np_roll, 'flip': _np_flip,
    'where': _np_where, 'tile': _np_tile, 'repeat': _np_repeat, 'take': _np_take, 'diagonal': _np_diagonal,
    'tril': lambda a, k=0: _np_tri(a, k, False), 'triu': lambda a, k=0: _np_tri(a, k, True),
    'swapaxes': lambda a, a1, a2: _np_axes(a, lambda t: t.transpose(a1 if a1 >= 0 else a1 + a.ndim, a2 if a2 >= 0 else a2 + a.ndim)),
    'moveaxis': lambda a, src, dst: _np_axes(a, lambda t: t.movedim(src if src >= 0 else src + a.ndim, dst if dst >= 0 else dst + a.ndim))
    if isinstance(src, (int, np.integer)) else _np_movement(np.moveaxis, (a, src, dst), {}),
    'expand_dims': lambda a, axis: a.reshape(np.expand_dims(np.empty(a.shape, dtype=np.int8), axis).shape),
    'squeeze': lambda a, axis=None: a.reshape(np.squeeze(np.empty(a.shape, dtype=np.int8), axis).shape),
    'sum': lambda a, axis=None, **kw: a.sum(axis, **{k: v for k, v in kw.items() if k in ('keepdims', 'initial') and v is not np._NoValue}),
    'dot': _np_dot, 'matmul': lambda a, b: _pair(a, b)[1] @ _pair(a, b)[2], 'vdot': _np_vdot, 'inner': _np_inner,
    'tensordot': _np_tensordot, 'kron': _np_kron, 'vander': _np_vander, 'insert': _np_insert,
    'trim_zeros': _np_trim_zeros,
    'cumsum': lambda a, axis=None, **kw: _np_scan(a, axis, False),
    'cumprod': lambda a, axis=None, **kw: _np_scan(a, axis, True),
    'cumulative_sum': lambda a, axis=None, include_initial=False, **kw: _np_scan(a, axis, False, include_initial),
    'cumulative_prod': lambda a, axis=None, include_initial=False, **kw: _np_scan(a, axis, True, include_initial),
    'outer': _np_outer, 'convolve': _np_convolve, 'prod': lambda a, axis=None, **kw: a.prod(axis),
    'trace': lambda a, offset=0, axis1=0, axis2=1, **kw: a.trace(offset, axis1, axis2),
    'nonzero': lambda a: np.nonzero(_np_nonzero_mask(a)), 'flatnonzero': lambda a: np.flatnonzero(_np_nonzero_mask(a)),
    'count_nonzero': lambda a, **kw: int(np.count_nonzero(_np_nonzero_mask(a))),
    'any': lambda a, **kw: bool(_np_nonzero_mask(a).any()), 'all': lambda a, **kw: bool(_np_nonzero_mask(a).all()),
    'array_equal': lambda a, b: a.shape == np.shape(b) and bool((a == b).all()),
    'solve': FieldArray.gauss_solve, 'inv': FieldArray.gauss_inv, 'det': FieldArray.gauss_det,
    'matrix_power': FieldArray.matrix_pow,
    'negative': lambda a: -a, 'add': lambda a, b: a + b, 'subtract': lambda a, b: a - b, 'multiply': lambda a, b: a * b,
    'diff': _np_diff, 'ediff1d': _np_ediff1d, 'cross': _np_cross, 'polyval': _np_polyval, 'polyadd': _np_polyadd,
    'polysub': lambda a1, a2: _np_polyadd(a1, a2, -1), 'polymul': _np_polymul, 'multi_dot': _np_multi_dot,
    'array_equiv': _np_array_equiv, 'iscomplexobj': lambda a: False, 'isrealobj': lambda a: True,
    'square': lambda a: a * a,
}
