"""ctypes wrapper around oracle/liboracle.so (fforacle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ADD, SUB, MUL, NEG, REDUCE = range(5)


def build():
    subprocess.run(['make', '-C', _HERE, '-s'], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_field_sizeof.restype = ctypes.c_size_t
        L.orc_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def limbs(x: int, n: int = 3):
    return (ctypes.c_uint64 * n)(*[(x >> (64 * i)) & (2**64 - 1) for i in range(n)])


def elem_bytes(modulus: int, binary: bool) -> int:
    """Storage width used on the device (include/ffgpu.h conventions)."""
    if binary:
        n = modulus.bit_length() - 1
        return 1 if n <= 8 else 4 if n <= 32 else 8 if n <= 64 else 16
    b = modulus.bit_length()
    if 64 < b <= 96 and (1 << b) - modulus < (1 << 31):
        return 12                      # p = 2^k - c, k <= 96: three 32-bit limbs (include/ffgpu.h)
    return 4 if b <= 32 else 8 if b <= 64 else 16 if b <= 128 else 24     # 24: three 64-bit limbs (2^k - c, k <= 192)


class CField:
    def __init__(self, modulus: int, binary: bool = False, eb: int = None):
        L = lib()
        self.modulus, self.binary = int(modulus), bool(binary)
        self.eb = eb or elem_bytes(self.modulus, self.binary)
        self._buf = ctypes.create_string_buffer(L.orc_field_sizeof())
        rc = L.orc_field_init(self._buf, int(self.binary), limbs(self.modulus), 3, self.eb)
        if rc:
            raise ValueError('bad modulus for oracle')

    def _p(self, a):
        return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None

    def ew(self, op, a: np.ndarray, b: np.ndarray = None) -> np.ndarray:
        a = np.ascontiguousarray(a)
        out = np.empty_like(a)
        n = a.nbytes // self.eb
        if b is not None:
            b = np.ascontiguousarray(b)
        lib().orc_ew(self._buf, op, self._p(a), self._p(b), self._p(out), ctypes.c_size_t(n))
        return out

    def split(self, s: np.ndarray, coef: np.ndarray, t: int, m: int) -> np.ndarray:
        """s: (n,) elements, coef: (t, n) elements -> (m, n).  Arrays are raw byte-compatible
        numpy arrays (uint8/uint32/uint64, (..., 2) uint64 for 16-byte and (..., 3) uint32 for 12-byte elements)."""
        s = np.ascontiguousarray(s)
        n = s.nbytes // self.eb
        coef = np.ascontiguousarray(coef)
        out = np.empty((m,) + s.shape, dtype=s.dtype)
        lib().orc_split(self._buf, self._p(s), self._p(coef) if t else None, ctypes.c_size_t(n), t, m,
                        self._p(out), ctypes.c_size_t(n), ctypes.c_size_t(n))
        return out

    def recombine(self, rows, lam, w: int = 1) -> np.ndarray:
        """rows: list of k arrays; lam: list of w*k canonical ints (row-major (w,k))."""
        rows = [np.ascontiguousarray(r) for r in rows]
        k = len(rows)
        n = rows[0].nbytes // self.eb
        ptrs = (ctypes.c_void_p * k)(*[r.ctypes.data for r in rows])
        lam_l = (ctypes.c_uint64 * (2 * w * k))()
        for i, v in enumerate(lam):
            lam_l[2 * i] = int(v) & (2**64 - 1)
            lam_l[2 * i + 1] = int(v) >> 64
        out = np.empty((w,) + rows[0].shape, dtype=rows[0].dtype)
        lib().orc_recombine(self._buf, ptrs, lam_l, k, w, self._p(out), ctypes.c_size_t(n), ctypes.c_size_t(n))
        return out[0] if w == 1 else out


def chacha_block(key32: bytes, w12_15, rounds: int = 20):
    """RFC 8439 section 2.3 block function on explicit state words 12..15."""
    key = (ctypes.c_uint32 * 8).from_buffer_copy(key32)
    w = (ctypes.c_uint32 * 4)(*w12_15)
    out = (ctypes.c_uint32 * 16)()
    lib().orc_chacha_block(key, w, rounds, out)
    return list(out)


def rng_coeffs(cf: 'CField', key32: bytes, nonce: int, rounds: int, t: int, n: int) -> np.ndarray:
    """(t, n) coefficient matrix exactly as the device CSPRNG draws it (mpyc_amd/csrc/rng.hpp)."""
    eb = cf.eb
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64, 12: np.uint32, 16: np.uint64}[eb]
    shape = (t, n, 2) if eb == 16 else (t, n, 3) if eb == 12 else (t, n)
    out = np.zeros(shape, dtype=dt)
    lib().orc_rng_coeffs(cf._buf, key32, ctypes.c_uint64(nonce), rounds, t, out.ctypes.data_as(ctypes.c_void_p),
                         ctypes.c_size_t(n), ctypes.c_size_t(n))
    return out


def prss_chacha(cf: 'CField', keys40, d: int, l: int, mask_bits: int, rounds: int, weights, n: int, out: np.ndarray = None,
                accumulate: bool = False) -> np.ndarray:
    """PRSS combination over ChaCha streams exactly as ffgpu_prss_chacha computes it (fields of up to 128 bits)."""
    eb = cf.eb
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64, 12: np.uint32, 16: np.uint64}[eb]
    shape = (n, 2) if eb == 16 else (n, 3) if eb == 12 else (n,)
    if out is None:
        out = np.zeros(shape, dtype=dt)
    ks = len(keys40)
    kb = b''.join(bytes(k) for k in keys40)
    w = (ctypes.c_uint64 * (2 * ks * d))()
    for i, v in enumerate(weights):
        w[2 * i] = int(v) & (2**64 - 1)
        w[2 * i + 1] = int(v) >> 64
    rc = lib().orc_prss_chacha(cf._buf, kb, ks, d, l, mask_bits, rounds, w, int(accumulate), out.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_size_t(n))
    if rc:
        raise ValueError('orc_prss_chacha: unsupported parameters')
    return out


def matmul(cf: 'CField', A: np.ndarray, B: np.ndarray, M: int, K: int, N: int) -> np.ndarray:
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    shape = (M * N, 2) if cf.eb == 16 else (M * N, 3) if cf.eb == 12 else (M * N,)
    C = np.zeros(shape, dtype=A.dtype)
    lib().orc_matmul(cf._buf, A.ctypes.data_as(ctypes.c_void_p), B.ctypes.data_as(ctypes.c_void_p),
                     C.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(M), ctypes.c_size_t(K), ctypes.c_size_t(N))
    return C


def sbox(x: np.ndarray, rows8, b: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty_like(x)
    r = (ctypes.c_uint8 * 8)(*rows8)
    lib().orc_sbox(x.ctypes.data_as(ctypes.c_void_p), r, ctypes.c_uint8(b), out.ctypes.data_as(ctypes.c_void_p),
                   ctypes.c_size_t(x.size))
    return out


def set_threads(t: int):
    lib().orc_set_threads(int(t))


def max_threads() -> int:
    return lib().orc_max_threads()
