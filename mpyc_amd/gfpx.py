"""Host-side mirror of the part of mpyc.gfpx the field path needs: polynomials over GF(2)
represented as non-negative integers (bit i = coefficient of X^i), reference
mpyc/gfpx.py:848-1121 (class BinaryPolynomial).

Only SCALAR values live here (moduli, single field elements, x-coordinates, Lagrange
coefficients); arrays of GF(2^n) elements live on the GPU (finfields.FieldArray).
"""
from __future__ import annotations

import re
import functools


def _clmul(a: int, b: int) -> int:
    """gfpx.py:988-1003 (_mul): carry-less product."""
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return r


def _cldivmod(a: int, b: int):
    """gfpx.py:1047-1066 (_divmod)."""
    if b == 0:
        raise ZeroDivisionError('division by zero polynomial')
    q = 0
    db = b.bit_length()
    while a.bit_length() >= db:
        s = a.bit_length() - db
        q ^= 1 << s
        a ^= b << s
    return q, a


def _clmod(a: int, b: int) -> int:
    return _cldivmod(a, b)[1]


def _clgcd(a: int, b: int) -> int:
    while b:
        a, b = b, _clmod(a, b)
    return a


def _clinvert(a: int, f: int) -> int:
    """gfpx.py:1084-1096 (_invert): inverse of a modulo f by extended Euclid."""
    if f == 0:
        raise ZeroDivisionError('division by zero polynomial')
    s, s1 = 1, 0
    a = _clmod(a, f)
    b = f
    while b:
        q, r = _cldivmod(a, b)
        a, b = b, r
        s, s1 = s1, s ^ _clmul(q, s1)
    if a != 1:
        raise ZeroDivisionError('inverse does not exist')
    return _clmod(s, f)


def _is_irreducible(a: int) -> bool:
    """gfpx.py:1098-1111 (Ben-Or style test: gcd(X^(2^i) - X, a) == 1 for i <= deg/2)."""
    if a <= 1:
        return False
    b = 2
    for _ in range((a.bit_length() - 1) // 2):
        b = _clmod(_clmul(b, b), a)
        if _clgcd(b ^ 2, a) != 1:
            return False
    return True


@functools.total_ordering
class BinaryPolynomial:
    """Polynomial over GF(2); `value` is its bit pattern (gfpx.py:848-884)."""

    __slots__ = 'value'
    p = 2

    def __init__(self, value=0, check=True):
        if isinstance(value, BinaryPolynomial):
            value = value.value
        elif isinstance(value, (list, tuple)):
            v = 0
            for c in reversed(value):
                v = (v << 1) | (int(c) & 1)
            value = v
        elif isinstance(value, str):
            value = self._from_terms(value)
        else:
            value = abs(int(value))           # gfpx.py:879-880 _from_int
        self.value = value

    _TERM = re.compile(r'(?P<const>[01])|x(?:\^(?P<exp>[0-9a-fA-FxXoObB_]+))?')

    @classmethod
    def _from_terms(cls, text):
        """'x^8+x^4+x^3+x+1' -> 0x11b: each '+'-separated term toggles one exponent bit (coefficients live in GF(2),
        so repeated terms cancel); the constants 0 / 1 and a bare 'x' are allowed."""
        bits = 0
        for term in re.sub(r'\s+', '', text).split('+'):
            mt = cls._TERM.fullmatch(term)
            if mt is None:
                raise ValueError('ill formatted polynomial')
            if mt.group('const') is not None:
                bits ^= int(mt.group('const'))
            else:
                bits ^= 1 << (int(mt.group('exp'), 0) if mt.group('exp') else 1)
        return bits

    def __int__(self):
        return self.value

    __index__ = __int__

    def degree(self):
        return self.value.bit_length() - 1

    def __hash__(self):
        return hash(('BinaryPolynomial', self.value))

    def __eq__(self, other):
        if isinstance(other, BinaryPolynomial):
            return self.value == other.value
        if isinstance(other, int):
            return self.value == abs(other)
        return NotImplemented

    def __lt__(self, other):
        return self.value < int(other)

    def __bool__(self):
        return bool(self.value)

    def _c(self, other):
        if isinstance(other, BinaryPolynomial):
            return other.value
        if isinstance(other, int):
            return abs(other)
        return None

    def __add__(self, other):
        o = self._c(other)
        return NotImplemented if o is None else BinaryPolynomial(self.value ^ o)

    __radd__ = __sub__ = __rsub__ = __add__

    def __neg__(self):
        return self

    def __mul__(self, other):
        o = self._c(other)
        return NotImplemented if o is None else BinaryPolynomial(_clmul(self.value, o))

    __rmul__ = __mul__

    def __mod__(self, other):
        o = self._c(other)
        return NotImplemented if o is None else BinaryPolynomial(_clmod(self.value, o))

    def __rmod__(self, other):
        # int % BinaryPolynomial: how the reference's array ctor reduces ints (finfields.py:724)
        o = self._c(other)
        return NotImplemented if o is None else BinaryPolynomial(_clmod(o, self.value))

    def __floordiv__(self, other):
        o = self._c(other)
        return NotImplemented if o is None else BinaryPolynomial(_cldivmod(self.value, o)[0])

    def __divmod__(self, other):
        q, r = _cldivmod(self.value, self._c(other))
        return BinaryPolynomial(q), BinaryPolynomial(r)

    def __pow__(self, e):
        r, b = 1, self.value
        while e:
            if e & 1:
                r = _clmul(r, b)
            b = _clmul(b, b)
            e >>= 1
        return BinaryPolynomial(r)

    def __lshift__(self, n):
        return BinaryPolynomial(self.value << n)

    def __rshift__(self, n):
        return BinaryPolynomial(self.value >> n)

    def to_bytes(self, length, byteorder):
        return self.value.to_bytes(length, byteorder)

    def __repr__(self):
        a = self.value
        if a == 0:
            return '0'
        terms = []
        for i in range(a.bit_length() - 1, -1, -1):
            if (a >> i) & 1:
                terms.append('1' if i == 0 else 'x' if i == 1 else f'x^{i}')
        return '+'.join(terms)

    # class-level helpers with the reference's names
    @staticmethod
    def mod(a, b):
        return BinaryPolynomial(_clmod(int(a), int(b)))

    @staticmethod
    def invert(a, b):
        return BinaryPolynomial(_clinvert(int(a), int(b)))

    @staticmethod
    def powmod(a, n, b):
        a, b = int(a), int(b)
        if n < 0:
            a, n = _clinvert(a, b), -n
        r = 1
        a = _clmod(a, b)
        while n:
            if n & 1:
                r = _clmod(_clmul(r, a), b)
            a = _clmod(_clmul(a, a), b)
            n >>= 1
        return BinaryPolynomial(r)

    @staticmethod
    def is_irreducible(a):
        return _is_irreducible(int(a))

    @staticmethod
    def next_irreducible(a):
        """gfpx.py:1113-1121: smallest irreducible polynomial greater than a."""
        a = int(a)
        if a <= 1:
            a = 2
        else:
            a += 1 + a % 2
            while not _is_irreducible(a):
                a += 2
        return BinaryPolynomial(a)


def GFpX(p):
    """mpyc.gfpx.GFpX(2) -> the polynomial type over GF(2).  Odd characteristic polynomials
    (extension fields GF(p^d), d > 1) are outside the accelerated path."""
    if p != 2:
        raise NotImplementedError('only GF(2)[X] is mirrored; GF(p^d), d > 1, p odd is out of scope')
    return BinaryPolynomial
