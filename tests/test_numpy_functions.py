"""NumPy functions on field arrays that the reference serves through its GENERIC __array_function__ / __array_ufunc__ path
(finfields.py:728-819: the function runs on the object ndarrays of Python integers and the result is reduced): ring
arithmetic composed from +, -, * -- np.diff, ediff1d, cross, polyval / polyadd / polysub / polymul, linalg.multi_dot,
square, array_equiv -- and negative slice steps.  The mirror composes them from its device operators; the expected values
here are NumPy's own results on object arrays reduced modulo p, i.e. what the reference returns (checked against the
reference itself when a checkout is importable).  CPU: host logic on tests/cpuctx.py; `-m gpu`: the kernels."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = next((r for r in ('/root/reference', os.path.join(ROOT, '_refstage')) if os.path.isdir(os.path.join(r, 'mpyc'))), None)
PRIMES = (2**61 - 1, 2**64 - 189, 2**128 - 173, 101)


def vals(a):
    return [int(v) for v in np.asarray(a.value).reshape(-1)]


def body(gff, p):
    F = gff.GF(p)
    rng = np.random.default_rng(p % 1000)

    def rnd(*shape):
        return np.array([int.from_bytes(rng.bytes(17), 'little') % p for _ in range(int(np.prod(shape)))], dtype=object).reshape(shape)
    A, B, V, W, T3, U3 = rnd(4, 5), rnd(5, 3), rnd(7), rnd(4), rnd(6, 3), rnd(6, 3)

    def same(got, want):
        want = np.asarray(want, dtype=object) % p
        assert isinstance(got, F.array), type(got)
        assert tuple(got.shape) == want.shape, (got.shape, want.shape)
        assert vals(got) == [int(v) for v in want.reshape(-1)]
    fa, fb, fv, fw, ft, fu = (F.array(x) for x in (A, B, V, W, T3, U3))

    def cross3(a, b):                  # (np.cross itself trips over object arrays of SMALL Python ints: an int64 temporary)
        a, b = np.broadcast_arrays(a, b)
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                         a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)
    same(np.diff(fv), np.diff(V))
    same(np.diff(fv, n=3), np.diff(V, n=3))
    same(np.diff(fa, axis=0), np.diff(A, axis=0))
    same(np.diff(fa, axis=-1, n=2), np.diff(A, axis=-1, n=2))
    same(np.diff(fv, prepend=F.array([5]), append=F.array([7, 9])), np.diff(V, prepend=np.array([5], dtype=object), append=np.array([7, 9], dtype=object)))
    same(np.ediff1d(fa), np.ediff1d(A))
    same(np.ediff1d(fv, to_begin=F.array([3]), to_end=F.array([4, 5])), np.concatenate([[3], np.diff(V), [4, 5]]))
    same(np.cross(ft[0], fu[0]), cross3(T3[0], U3[0]))
    same(np.cross(ft, fu), cross3(T3, U3))
    same(np.cross(ft, fu[1]), cross3(T3, U3[1]))
    same(np.polyval(fw, fv), np.polyval(W, V))
    same(np.polyadd(fv, fw), np.polyadd(V, W))
    same(np.polysub(fw, fv), np.polysub(W, V))
    same(np.polymul(fv, fw), np.polymul(V, W))
    same(np.linalg.multi_dot([fa, fb, fb.T]), A @ B @ B.T)
    same(np.square(fv), V * V)
    same(np.absolute(fv), V)
    assert np.array_equiv(fv, fv) and np.array_equiv(fa, fa[0]) == bool((A == A[0]).all()) and not np.array_equiv(fv, fw)
    assert np.iscomplexobj(fv) is False and np.isrealobj(fv) is True
    # negative slice steps (views in NumPy; the reference slices its object ndarray)
    for key in (slice(None, None, -1), slice(None, None, -2), slice(5, 1, -1), slice(1, 5, -1), slice(-2, None, -3), slice(None, 2, -1)):
        same(fv[key], V[key])
    for key in ((slice(None), slice(None, None, -1)), (slice(None, None, -1), slice(1, None)), (Ellipsis, slice(None, None, -2)),
                (None, slice(None, None, -1)), (2, slice(None, None, -1)), (slice(None, None, -1), 1),
                (slice(None, None, -1), None, slice(3, 0, -1))):
        same(fa[key], A[key])
    x, X = F.array(V.copy()), V.copy()
    x[::-1] = fv[:7]
    X[::-1] = V[:7]
    same(x, X)
    y, Y = F.array(A.copy()), A.copy()
    y[::-2, ::-1] = fa[:2]
    Y[::-2, ::-1] = A[:2]
    same(y, Y)
    y[1, ::-1] = 7
    Y[1, ::-1] = 7
    same(y, Y)
    with pytest.raises(NotImplementedError):
        fa[[0, 1], ::-1]


@pytest.mark.parametrize('p', PRIMES)
def test_numpy_ring_functions_host_logic(monkeypatch, p):
    from cpuctx import use_cpu_contexts
    import mpyc_amd.finfields as gff
    use_cpu_contexts(monkeypatch)
    monkeypatch.setattr(gff, '_ctx_cache', {})
    gff._pGF.cache_clear()
    body(gff, p)
    gff._pGF.cache_clear()


@pytest.mark.skipif(REF is None, reason='no importable mpyc checkout')
def test_expected_values_are_the_references(monkeypatch):
    """the expectation used above (NumPy on object arrays, reduced) IS the reference's behaviour"""
    monkeypatch.syspath_prepend(REF)
    from mpyc import finfields as rff
    p = 2**61 - 1
    F = rff.GF(p)
    # (entries large enough that NumPy's own np.cross keeps object temporaries: with small Python ints it builds an int64
    # temporary and fails inside the reference as well)
    V = np.array([2**60 + 3, p - 1, 2**59 + 77, 2**60, 2**58 + 5, 2**57 + 6, 2**56 + 9], dtype=object)
    W = np.array([p - 2, 2**58 + 4, 10**18, 2**57 + 1], dtype=object)
    fv, fw = F.array(V), F.array(W)
    for got, want in ((np.diff(fv), np.diff(V)), (np.ediff1d(fv), np.ediff1d(V)), (np.cross(fv[:3], fw[:3]), np.cross(V[:3], W[:3])),
                      (np.polyval(fw, fv), np.polyval(W, V)), (np.polymul(fv, fw), np.polymul(V, W)), (np.polyadd(fv, fw), np.polyadd(V, W)),
                      (np.square(fv), V * V), (fv[::-2], V[::-2])):
        assert [int(v) for v in got.value.reshape(-1)] == [int(v) % p for v in np.asarray(want, dtype=object).reshape(-1)]


@pytest.mark.gpu
@pytest.mark.parametrize('p', PRIMES)
def test_numpy_ring_functions_on_gpu(p):
    import torch
    assert torch.cuda.is_available()
    import mpyc_amd.finfields as gff
    body(gff, p)
