"""oracle/nporacle.py -- the reference's vectorised CPU path restated with NumPy
dtype=object arrays, i.e. computed the way the reference computes it (a C loop over
PyObject* calling int.__mul__/__mod__).  TEST INFRASTRUCTURE ONLY; used by bench.py's
cpu_baseline leg to time "the reference's own CPU path" shape of work on the GPU box
(the reference package itself cannot travel there), and cross-checked against
pyoracle in tests/test_oracle_golden.py.  Prime fields only.

    mul        finfields.py:1105-1112  cls(self.value * other) ; :724 value %= modulus
    split      thresha.py:61-63        (V @ concatenate((s, C))) % p,  V = vander(1..m)
    recombine  thresha.py:128-129      vector @ field.array(shares)  (+ % p in the ctor)
"""
import numpy as np


def mul(p, a, b):
    c = a * b
    c %= p
    return c


def split(p, s, C, t, m):
    n = len(s)
    V = np.vander(np.array(list(range(1, m + 1)), dtype=object), N=t + 1, increasing=True)
    return (V @ np.concatenate((s.reshape(1, n), C.reshape(t, n)))) % p


def recombine(p, rows, lam):
    shares = np.array(rows, dtype=object)
    shares %= p
    sums = np.array([lam], dtype=object) @ shares
    sums %= p
    return sums[0]
