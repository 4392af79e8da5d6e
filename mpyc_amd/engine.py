"""Device-side engine: a field context plus device-resident limb arrays.

This is the layer directly above the C ABI (include/ffgpu.h).  Arrays live in HBM
as fixed-width little-endian limbs (DevArray); Python integers exist only at the
API edges (from_ints / to_ints), because boxing 10^7 PyLongs costs more than every
kernel here put together.  torch is used for device memory (caching allocator,
streams, interop with torch.distributed); all arithmetic goes through libffgpu.so.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _ffi

_MASK64 = (1 << 64) - 1


def _np_dtype(eb: int):
    return {1: np.uint8, 4: np.uint32, 8: np.uint64, 12: np.uint32, 16: np.uint64, 24: np.uint64}[eb]


def _torch_dtype(eb: int):
    return {1: torch.uint8, 4: torch.int32, 8: torch.int64, 12: torch.int32, 16: torch.int64, 24: torch.int64}[eb]


_MASK64_OBJ = (1 << 64) - 1


def limbs_of(eb: int) -> int:
    """Trailing limb dimension of a device tensor: 0 (scalar dtype), 2 (16 bytes = 2 x int64), 3 (12 bytes = 3 x int32,
    24 bytes = 3 x int64)."""
    return {16: 2, 12: 3, 24: 3}.get(eb, 0)


def ints_to_np(vals: Iterable[int], eb: int) -> np.ndarray:
    """Canonical Python ints -> limb array ((n,) for eb<=8, (n,2) uint64 for eb=16, (n,3) uint32 for eb=12)."""
    if isinstance(vals, np.ndarray) and vals.dtype != object:
        vals = vals.reshape(-1)
        if eb in (16, 24):
            out = np.zeros((vals.size, eb // 8), dtype=np.uint64)
            out[:, 0] = vals.astype(np.uint64)
            return out
        if eb == 12:
            v64 = vals.astype(np.uint64)
            out = np.zeros((vals.size, 3), dtype=np.uint32)
            out[:, 0] = (v64 & np.uint64(0xffffffff)).astype(np.uint32)
            out[:, 1] = (v64 >> np.uint64(32)).astype(np.uint32)
            return out
        return vals.astype(_np_dtype(eb))
    vals = list(vals) if not isinstance(vals, (list, np.ndarray)) else vals
    n = len(vals)
    if n == 0:
        return np.zeros((0, limbs_of(eb)) if limbs_of(eb) else 0, dtype=_np_dtype(eb))
    obj = vals if isinstance(vals, np.ndarray) else np.array(vals, dtype=object)
    if eb <= 8:
        return obj.astype(np.uint64).astype(_np_dtype(eb))
    # two / three limbs: split inside NumPy's object loops (3 big-int operations per element instead of a Python-level
    # to_bytes + join per element)
    if eb == 24:
        try:
            out = np.empty((n, 3), dtype=np.uint64)
            out[:, 0] = (obj & _MASK64_OBJ).astype(np.uint64)
            out[:, 1] = ((obj >> 64) & _MASK64_OBJ).astype(np.uint64)
            out[:, 2] = (obj >> 128).astype(np.uint64)
            return out
        except (TypeError, OverflowError):
            buf = b''.join(int(v).to_bytes(24, 'little') for v in obj)
            return np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).copy()
    try:
        lo64 = (obj & _MASK64_OBJ).astype(np.uint64)
        hi64 = (obj >> 64).astype(np.uint64)
    except (TypeError, OverflowError):       # elements that are not plain ints (numpy integers in an object array, ...)
        ints = [int(v) for v in obj]
        buf = b''.join(v.to_bytes(16, 'little') for v in ints)
        both = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2)
        lo64, hi64 = both[:, 0], both[:, 1]
    if eb == 16:
        out = np.empty((n, 2), dtype=np.uint64)
        out[:, 0], out[:, 1] = lo64, hi64
        return out
    out = np.empty((n, 3), dtype=np.uint32)
    out[:, 0] = (lo64 & np.uint64(0xffffffff)).astype(np.uint32)
    out[:, 1] = (lo64 >> np.uint64(32)).astype(np.uint32)
    out[:, 2] = hi64.astype(np.uint32)
    return out


def np_to_objects(arr: np.ndarray, eb: int) -> np.ndarray:
    """Limb array -> 1-D object ndarray of Python ints, converted inside NumPy's C loops (this is the price of the
    reference's representation: ~30 ns per element for one limb, three object operations per element for two)."""
    if eb == 24:
        a = arr.reshape(-1, 3)
        if not len(a):
            return np.empty(0, dtype=object)
        return (a[:, 2].astype(object) << 128) | (a[:, 1].astype(object) << 64) | a[:, 0].astype(object)
    if eb == 16:
        a = arr.reshape(-1, 2)
        if not len(a):
            return np.empty(0, dtype=object)
        return (a[:, 1].astype(object) << 64) | a[:, 0].astype(object)
    if eb == 12:
        a = arr.view(np.uint32).reshape(-1, 3)
        if not len(a):
            return np.empty(0, dtype=object)
        lo = a[:, 0].astype(np.uint64) | (a[:, 1].astype(np.uint64) << np.uint64(32))
        return (a[:, 2].astype(object) << 64) | lo.astype(object)
    return arr.reshape(-1).astype(object)


def np_to_ints(arr: np.ndarray, eb: int) -> List[int]:
    return np_to_objects(arr, eb).tolist()


class DevArray:
    """n field elements resident in HBM (row of limbs).  Thin wrapper over a torch tensor."""

    __slots__ = ('ctx', 't', 'n')

    def __init__(self, ctx: 'FieldContext', t: torch.Tensor, n: int):
        self.ctx, self.t, self.n = ctx, t, n

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()

    def to_numpy(self) -> np.ndarray:
        a = self.t.cpu().numpy()
        eb = self.ctx.elem_bytes
        if eb in (4, 12):
            a = a.view(np.uint32)
        elif eb >= 8:
            a = a.view(np.uint64)
        return a

    def to_ints(self) -> List[int]:
        return np_to_ints(self.to_numpy(), self.ctx.elem_bytes)

    def clone(self) -> 'DevArray':
        return DevArray(self.ctx, self.t.clone(), self.n)

    def __len__(self):
        return self.n


class DevMatrix:
    """(rows, n) elements with 256-byte aligned rows (so every row takes the 16 B/lane path)."""

    __slots__ = ('ctx', 't', 'rows', 'n', 'stride')

    def __init__(self, ctx, t, rows, n, stride):
        self.ctx, self.t, self.rows, self.n, self.stride = ctx, t, rows, n, stride

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()

    def row(self, i: int) -> DevArray:
        eb = self.ctx.elem_bytes
        if limbs_of(eb):
            return DevArray(self.ctx, self.t[i, :self.n, :], self.n)
        return DevArray(self.ctx, self.t[i, :self.n], self.n)

    def to_numpy(self) -> np.ndarray:
        a = self.t.cpu().numpy()
        eb = self.ctx.elem_bytes
        if eb in (4, 12):
            a = a.view(np.uint32)
        elif eb >= 8:
            a = a.view(np.uint64)
        return a[:, :self.n]

    def to_ints(self) -> List[List[int]]:
        a = self.to_numpy()
        return [np_to_ints(a[i], self.ctx.elem_bytes) for i in range(self.rows)]


class RngState:
    """Key / nonce / rounds of the device CSPRNG in device memory (FieldContext.rng_state)."""

    __slots__ = ('ctx', 't', 'pending')

    def __init__(self, ctx, t):
        self.ctx, self.t, self.pending = ctx, t, 0

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()

    def take_offset(self) -> int:
        """Deferred advance (ffgpu_rng_state_advance): the next launch of a sequence draws from nonce + offset and
        leaves the device nonce alone; commit() advances it once by the number of launches."""
        off = self.pending
        self.pending += 1
        return off

    def commit(self):
        """One nonce update for all deferred launches since the last commit (stream-ordered, capturable).  Every
        engine call that advances the nonce itself commits first, so offsets never collide with it."""
        if self.pending:
            by, self.pending = self.pending, 0
            _ffi.check(self.ctx._L.ffgpu_rng_state_advance(self.ctx._h, self.ptr, by, self.ctx._stream()), 'rng_state_advance')

    def nonce(self) -> int:
        w = self.t.cpu().numpy().view(np.uint32)
        return int(w[8]) | (int(w[9]) << 32)


class CapturedLaunches:
    """A fixed sequence of engine calls captured into a HIP graph (through torch.cuda.CUDAGraph, which
    captures every launch issued on the capture stream -- including libffgpu's, since the engine always
    launches on torch's current stream).  For launch-bound work: a gate on a few thousand elements is
    three kernels of ~3 us each, dominated by per-launch host cost when issued one by one.
    The captured kernels read and write the SAME device buffers on every replay: refresh inputs with
    tensor.copy_() into those buffers, then replay()."""

    def __init__(self, fn, warmup: int = 2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = fn()          # arrays allocated during capture live in the graph's pool

    def replay(self):
        self.graph.replay()


class FieldContext:
    """One finite field on one GPU.  modulus: prime p, or (binary=True) the bit pattern of the
    irreducible polynomial.  Mirrors what finfields.GF(modulus) fixes for an array type
    (finfields.py:23-60)."""

    def __init__(self, modulus: int, binary: bool = False, device: Optional[int] = None):
        L = _ffi.lib()
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.modulus = int(modulus)
        self.binary = bool(binary)
        self.device = int(device)
        h = ctypes.c_void_p()
        nl = 3
        rc = L.ffgpu_ctx_create(_ffi.BINARY if binary else _ffi.PRIME, _ffi.limbs(self.modulus, nl), nl,
                                self.device, ctypes.byref(h))
        _ffi.check(rc, 'ctx_create')
        self._h = h
        self.elem_bytes = L.ffgpu_ctx_elem_bytes(h)
        self.limbs = limbs_of(self.elem_bytes)        # trailing limb dimension of device tensors (0, 2 or 3)
        self.scalar_limbs = int(L.ffgpu_ctx_scalar_limbs(h))   # limbs per host scalar in the C ABI (2; 3 for 24-byte fields)
        self.reduction = _ffi.RED_NAMES.get(L.ffgpu_ctx_reduction(h), '?')
        if binary:
            self.order = 1 << (self.modulus.bit_length() - 1)
        else:
            self.order = self.modulus
        self._L = L

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._L.ffgpu_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- memory ---------------------------------------------------------
    @property
    def torch_device(self):
        return torch.device('cuda', self.device)

    def pci_bus_id(self) -> str:
        """PCI bus id of this context's GPU (multi-process runs report it per rank)."""
        buf = ctypes.create_string_buffer(32)
        _ffi.check(self._L.ffgpu_device_pci_bus_id(self.device, buf, 32), "device_pci_bus_id")
        return buf.value.decode()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, n: int) -> DevArray:
        eb = self.elem_bytes
        shape = (n, limbs_of(eb)) if limbs_of(eb) else (n,)
        return DevArray(self, torch.empty(shape, dtype=_torch_dtype(eb), device=self.torch_device), n)

    def empty_matrix(self, rows: int, n: int) -> DevMatrix:
        eb = self.elem_bytes
        per = 64 if eb == 12 else 32 if eb == 24 else 256 // eb      # elements per 256-byte-aligned pitch unit (768 B for 12- and 24-byte elements)
        stride = max(per, (n + per - 1) // per * per)
        if (stride * eb) % 16384 == 0:
            # rows a multiple of 16 KiB apart alias onto the same HBM channels when m rows are
            # written at once (measured -12 % at a 64 MiB pitch): skew the pitch by 17 x 256 B
            stride += 17 * per
        shape = (rows, stride, limbs_of(eb)) if limbs_of(eb) else (rows, stride)
        return DevMatrix(self, torch.empty(shape, dtype=_torch_dtype(eb), device=self.torch_device), rows, n,
                         stride)

    def from_numpy(self, a: np.ndarray) -> DevArray:
        """a: limb array as produced by ints_to_np (already canonical)."""
        eb = self.elem_bytes
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        n = a.shape[0]
        if eb in (4, 12):
            t = torch.from_numpy(a.view(np.int32))
        elif eb >= 8:
            t = torch.from_numpy(a.view(np.int64))
        else:
            t = torch.from_numpy(a)
        return DevArray(self, t.to(self.torch_device), n)

    def from_ints(self, vals: Sequence[int]) -> DevArray:
        """Canonical ints (0 <= v < order) -> device."""
        return self.from_numpy(ints_to_np(vals, self.elem_bytes))

    def matrix_from_numpy(self, a: np.ndarray) -> DevMatrix:
        rows, n = a.shape[0], a.shape[1]
        mtx = self.empty_matrix(rows, n)
        for i in range(rows):
            mtx.row(i).t.copy_(self.from_numpy(a[i]).t)
        return mtx

    # ---- operand checks ---------------------------------------------------
    # The kernels take raw pointers and ONE element count: every operand (and a caller-supplied output) must have
    # exactly that many elements and belong to this field, otherwise a launch would read or write past an
    # allocation.  Share rows arrive from peers (from_wire derives their length from the message size), so a short
    # or ragged row must be refused here -- the reference's field.array(shares) raises on the same input.
    def _same(self, n: int, *arrays, what: str = 'operand'):
        for a in arrays:
            if a is None:
                continue
            if a.ctx is not self and (a.ctx.modulus != self.modulus or a.ctx.binary != self.binary
                                      or a.ctx.device != self.device):
                raise ValueError(f'{what} belongs to a different field context')
            if a.n != n:
                raise ValueError(f'{what} has {a.n} elements, expected {n}')

    def _out_matrix(self, out: Optional[DevMatrix], rows: int, n: int) -> DevMatrix:
        if out is None:
            return self.empty_matrix(rows, n)
        if out.rows < rows or out.n != n:
            raise ValueError(f'output matrix is ({out.rows}, {out.n}), need ({rows}, {n})')
        return out

    # ---- element-wise -----------------------------------------------------
    def _ew2(self, fn, a: DevArray, b: DevArray, out: Optional[DevArray]) -> DevArray:
        self._same(a.n, b, out)
        out = out or self.empty(a.n)
        _ffi.check(fn(self._h, a.ptr, b.ptr, out.ptr, a.n, self._stream()), fn.__name__)
        return out

    def add(self, a, b, out=None):
        return self._ew2(self._L.ffgpu_add, a, b, out)

    def sub(self, a, b, out=None):
        return self._ew2(self._L.ffgpu_sub, a, b, out)

    def mul(self, a, b, out=None):
        return self._ew2(self._L.ffgpu_mul, a, b, out)

    def neg(self, a, out=None):
        self._same(a.n, out, what='output')
        out = out or self.empty(a.n)
        _ffi.check(self._L.ffgpu_neg(self._h, a.ptr, out.ptr, a.n, self._stream()), 'neg')
        return out

    def reduce(self, raw: DevArray, out=None):
        self._same(raw.n, out, what='output')
        out = out or self.empty(raw.n)
        _ffi.check(self._L.ffgpu_reduce(self._h, raw.ptr, out.ptr, raw.n, self._stream()), 'reduce')
        return out

    def _ews(self, fn, a, scalar: int, out):
        self._same(a.n, out, what='output')
        out = out or self.empty(a.n)
        _ffi.check(fn(self._h, a.ptr, _ffi.limbs(scalar, self.scalar_limbs), out.ptr, a.n, self._stream()), fn.__name__)
        return out

    def add_scalar(self, a, s: int, out=None):
        return self._ews(self._L.ffgpu_add_scalar, a, s, out)

    def mul_scalar(self, a, s: int, out=None):
        return self._ews(self._L.ffgpu_mul_scalar, a, s, out)

    def rsub_scalar(self, a, s: int, out=None):
        return self._ews(self._L.ffgpu_rsub_scalar, a, s, out)

    def muladd(self, a, b, c, out=None):
        self._same(a.n, b, c, out)
        out = out or self.empty(a.n)
        _ffi.check(self._L.ffgpu_muladd(self._h, a.ptr, b.ptr, c.ptr, out.ptr, a.n, self._stream()), 'muladd')
        return out

    def pow(self, a, e: int, out=None):
        """a ** e for a public exponent 0 <= e < 2^128, one kernel (finfields.py:1159-1187)."""
        if e < 0 or e >> 192:
            raise ValueError('exponent out of range')
        self._same(a.n, out, what='output')
        out = out or self.empty(a.n)
        _ffi.check(self._L.ffgpu_pow(self._h, a.ptr, _ffi.limbs(e, 3), 3, out.ptr, a.n, self._stream()), 'pow')
        return out

    def inv(self, a, out=None, check_zero: bool = True):
        """Element-wise inverse (batched, one kernel).  Raises ZeroDivisionError like the reference
        if any element is zero (costs one device->host flag read); check_zero=False skips that."""
        self._same(a.n, out, what='output')
        out = out or self.empty(a.n)
        flag = torch.zeros(1, dtype=torch.int32, device=self.torch_device) if check_zero else None
        _ffi.check(self._L.ffgpu_inv(self._h, a.ptr, out.ptr, a.n, flag.data_ptr() if check_zero else None,
                                     self._stream()), 'inv')
        if check_zero and int(flag.item()):
            raise ZeroDivisionError('inverse of 0 does not exist')
        return out

    def beaver_combine(self, z, x, y, d, e, add_de: bool, out=None):
        """z + d*y + e*x (+ d*e): local step of Beaver multiplication (not a reference function; see ffgpu.h)."""
        self._same(z.n, x, y, d, e, out)
        out = out or self.empty(z.n)
        _ffi.check(self._L.ffgpu_beaver_combine(self._h, z.ptr, x.ptr, y.ptr, d.ptr, e.ptr, int(bool(add_de)), out.ptr,
                                                z.n, self._stream()), 'beaver_combine')
        return out

    # ---- sharing ----------------------------------------------------------
    def split(self, secrets: DevArray, coeffs: Optional[DevMatrix], t: int, m: int,
              out: Optional[DevMatrix] = None, mul_by: Optional[DevArray] = None) -> DevMatrix:
        """np_random_split with the coefficient matrix supplied (thresha.py:47-64).
        mul_by: fuse the local product secrets*mul_by (runtime.py:1134-1138)."""
        n = secrets.n
        if t and (coeffs is None or coeffs.rows < t or coeffs.n != n):
            raise ValueError('coefficient matrix must be (t, n)')
        self._same(n, mul_by)
        out = self._out_matrix(out, m, n)
        cptr = coeffs.ptr if t else None
        cstride = coeffs.stride if t else 0
        if mul_by is None:
            rc = self._L.ffgpu_split(self._h, secrets.ptr, cptr, cstride, t, m, out.ptr, out.stride, n,
                                     self._stream())
        else:
            rc = self._L.ffgpu_mul_split(self._h, secrets.ptr, mul_by.ptr, cptr, cstride, t, m, out.ptr,
                                         out.stride, n, self._stream())
        _ffi.check(rc, 'split')
        return out

    def rng_state(self, key: Optional[bytes] = None, nonce: int = 0, rounds: int = 20) -> 'RngState':
        """Device-resident generator state (key from the host CSPRNG by default).  Pass it as `state=` to
        split_rng: the kernels then read key/nonce from device memory and the nonce advances on the device
        after every call, so the calls can be captured in a HIP graph and every replay draws fresh
        coefficients."""
        import secrets as _secrets
        key = key if key is not None else _secrets.token_bytes(32)
        if len(key) != 32:
            raise ValueError('key must be 32 bytes')
        t = torch.zeros(int(self._L.ffgpu_rng_state_bytes()), dtype=torch.uint8, device=self.torch_device)
        _ffi.check(self._L.ffgpu_rng_state_init(self._h, t.data_ptr(), key, nonce, rounds, self._stream()),
                   'rng_state_init')
        return RngState(self, t)

    def rng_coeffs(self, key: bytes, nonce: int, t: int, n: int, rounds: int = 20,
                   out: Optional[DevMatrix] = None) -> DevMatrix:
        """Materialise the (t, n) coefficient matrix the fused split_rng kernel draws for
        (key, nonce, rounds) -- device CSPRNG, mpyc_amd/csrc/rng.hpp."""
        if len(key) != 32:
            raise ValueError('key must be 32 bytes')
        out = out or self.empty_matrix(t, n)
        _ffi.check(self._L.ffgpu_rng_coeffs(self._h, key, nonce, rounds, t, out.ptr, out.stride, n,
                                            self._stream()), 'rng_coeffs')
        return out

    def split_rng(self, secrets: DevArray, t: int, m: int, key: Optional[bytes] = None, nonce: int = 0,
                  rounds: int = 20, out: Optional[DevMatrix] = None,
                  mul_by: Optional[DevArray] = None, state: Optional['RngState'] = None) -> DevMatrix:
        """np_random_split with coefficients drawn on the device (production mode): the t*n
        random coefficients never touch HBM.  key: 32 bytes; default = fresh from the host CSPRNG
        (the reference draws from `secrets` too, thresha.py:58)."""
        import secrets as _secrets
        n = secrets.n
        self._same(n, mul_by)
        out = self._out_matrix(out, m, n)
        if state is not None:
            state.commit()
            _ffi.check(self._L.ffgpu_split_rng_state(self._h, secrets.ptr, mul_by.ptr if mul_by is not None else None,
                                                     state.ptr, t, m, out.ptr, out.stride, n, self._stream()),
                       'split_rng_state')
            return out
        if key is None:
            key = _secrets.token_bytes(32)
        if len(key) != 32:
            raise ValueError('key must be 32 bytes')
        if mul_by is None:
            rc = self._L.ffgpu_split_rng(self._h, secrets.ptr, key, nonce, rounds, t, m, out.ptr, out.stride, n,
                                         self._stream())
        else:
            rc = self._L.ffgpu_mul_split_rng(self._h, secrets.ptr, mul_by.ptr, key, nonce, rounds, t, m, out.ptr,
                                             out.stride, n, self._stream())
        _ffi.check(rc, 'split_rng')
        return out

    def gate(self, rows_a: Sequence[DevArray], lam_a: Sequence[int], rows_b: Optional[Sequence[DevArray]],
             lam_b: Optional[Sequence[int]], t: int, m: int, key: Optional[bytes] = None, nonce: int = 0,
             rounds: int = 20, state: Optional['RngState'] = None, out: Optional[DevMatrix] = None) -> DevMatrix:
        """Fused chain gate (ffgpu_gate_rng): shares of A*B with A = sum lam_a[j]*rows_a[j], B likewise
        (rows_b None: B = A); recombination, product and share generation in one pass."""
        import secrets as _secrets
        n = rows_a[0].n
        ka, pa, la = self._rec_args(rows_a, lam_a, 1)
        if rows_b:
            kb, pb, lb = self._rec_args(rows_b, lam_b, 1)
        else:
            kb, pb, lb = 0, None, None
        self._same(n, *(rows_b or ()), what='share row')
        out = self._out_matrix(out, m, n)
        if state is None and key is None:
            key = _secrets.token_bytes(32)
        if state is not None:
            state.commit()
        _ffi.check(self._L.ffgpu_gate_rng(self._h, pa, la, ka, pb, lb, kb, key, nonce, rounds,
                                          state.ptr if state is not None else None, t, m, out.ptr, out.stride, n,
                                          self._stream()), 'gate_rng')
        return out

    def gate_batch(self, rows_a: Sequence[DevArray], lam_a: Sequence[int], stride_a: int,
                   rows_b: Optional[Sequence[DevArray]], lam_b: Optional[Sequence[int]], stride_b: int,
                   t: int, m: int, nbatch: int, out: DevMatrix, out_rows_per_party: int,
                   key: Optional[bytes] = None, nonce: int = 0, rounds: int = 20,
                   state: Optional['RngState'] = None) -> DevMatrix:
        """`nbatch` chain gates in one launch (ffgpu_gate_rng_batch): rows_a / rows_b are the operand rows of gate 0;
        gate y reads them `y * stride_a` (`y * stride_b`) elements further on.  `out` is a DevMatrix of
        m * out_rows_per_party rows used as [recipient][sender]: gate y writes the share row for party i+1 into row
        i * out_rows_per_party + y.  The caller owns the layout (mpyc_amd/protocols.py); operand rows of gate 0
        and the output are length-checked here."""
        import secrets as _secrets
        n = rows_a[0].n
        if nbatch < 1 or out_rows_per_party < nbatch or out.rows < m * out_rows_per_party or out.n != n:
            raise ValueError('output block does not hold m x nbatch share rows of n elements')
        ka, pa, la = self._rec_args(rows_a, lam_a, 1)
        if rows_b:
            kb, pb, lb = self._rec_args(rows_b, lam_b, 1)
            self._same(n, *rows_b, what='share row')
        else:
            kb, pb, lb = 0, None, None
        if state is None and key is None:
            key = _secrets.token_bytes(32)
        defer = 0
        if state is not None:
            nonce, defer = state.take_offset(), 1          # deferred advance: state.commit() follows the sequence
        _ffi.check(self._L.ffgpu_gate_rng_batch(self._h, pa, la, ka, stride_a, pb, lb, kb, stride_b, key, nonce, rounds,
                                                state.ptr if state is not None else None, defer, t, m, out.ptr,
                                                out.stride * out_rows_per_party, out.stride, n, nbatch,
                                                self._stream()), 'gate_rng_batch')
        return out

    def _rec_args(self, rows: Sequence[DevArray], lambdas: Sequence[int], w: int):
        k = len(rows)
        if not k:
            raise ValueError('no share rows')
        if len(lambdas) != w * k:
            raise ValueError('need w*k lambda values')
        self._same(rows[0].n, *rows, what='share row')
        ptrs = (ctypes.c_void_p * k)(*[r.ptr for r in rows])
        lam = self._scalars(lambdas)
        return k, ptrs, lam

    def recombine(self, rows: Sequence[DevArray], lambdas: Sequence[int], w: int = 1, out=None):
        """out[r] = sum_j lambdas[r*k+j] * rows[j]  (thresha.py:119-132)."""
        k, ptrs, lam = self._rec_args(rows, lambdas, w)
        n = rows[0].n
        if w == 1:
            self._same(n, out, what='output')
            out = out or self.empty(n)
            stride = n
        else:
            out = self._out_matrix(out, w, n)
            stride = out.stride
        _ffi.check(self._L.ffgpu_recombine(self._h, ptrs, lam, k, w, out.ptr, stride, n, self._stream()),
                   'recombine')
        return out

    def recombine_plan(self, rows: Sequence[DevArray], lambdas: Sequence[int], out, w: int = 1):
        """Pre-marshal a recombination (row pointers + Lagrange vector) for repeated launches:
        the returned callable only issues the kernel (hot loops, benchmarks)."""
        k, ptrs, lam = self._rec_args(rows, lambdas, w)
        n = rows[0].n
        if w == 1:
            self._same(n, out, what='output')
        else:
            self._out_matrix(out, w, n)
        stride = n if w == 1 else out.stride
        fn, h, optr = self._L.ffgpu_recombine, self._h, out.ptr
        keep = (rows, out)

        def launch():
            _ffi.check(fn(h, ptrs, lam, k, w, optr, stride, n, self._stream()), 'recombine')
            return keep[1]
        return launch

    def matmul(self, A: DevArray, B: DevArray, M: int, K: int, N: int, out: Optional[DevArray] = None) -> DevArray:
        """C (M, N) = A (M, K) @ B (K, N): contiguous row-major operands (finfields.py:1126-1135)."""
        if A.n != M * K or B.n != K * N:
            raise ValueError('matmul: operand sizes do not match the shapes')
        if out is not None:
            self._same(M * N, out, what='matmul output')
            eb = self.elem_bytes
            for x in (A, B):                       # C is written while other tiles still read A and B
                if out.ptr < x.ptr + x.n * eb and x.ptr < out.ptr + out.n * eb:
                    raise ValueError('matmul: the output overlaps an operand')
        out = out or self.empty(M * N)
        _ffi.check(self._L.ffgpu_matmul(self._h, A.ptr, K, B.ptr, N, out.ptr, N, M, K, N, self._stream()), 'matmul')
        return out

    def sqrt_cl(self, a: DevArray, out: Optional[DevArray] = None) -> DevArray:
        """Square roots for p = 1 mod 4 (Cipolla-Lehmer, finfields.py:447-470)."""
        out = out or self.empty(a.n)
        _ffi.check(self._L.ffgpu_sqrt_cl(self._h, a.ptr, out.ptr, a.n, self._stream()), 'sqrt_cl')
        return out

    def gauss(self, a: DevArray, n: int, ncols: int, batch: int = 1, det: bool = False):
        """Gaussian elimination in place on `batch` row-major (n, ncols) matrices (finfields.py:872-955).
        det=False: (A | B) -> (. | A^-1 B).  det=True: returns the reference's determinant per matrix.
        Returns (det DevArray or None, singular flags as an int32 tensor on the device)."""
        if a.n != batch * n * ncols:
            raise ValueError('array size does not match (batch, n, ncols)')
        sing = torch.empty(max(batch, 1), dtype=torch.int32, device=self.torch_device)
        d = self.empty(batch) if det else None
        _ffi.check(self._L.ffgpu_gauss(self._h, a.ptr, n, ncols, batch, 1 if det else 0, d.ptr if det else None,
                                       sing.data_ptr(), self._stream()), 'gauss')
        return d, sing

    def group_matvec(self, x: DevArray, matrix: Sequence[Sequence[int]], bias: Optional[Sequence[int]] = None,
                     out: Optional[DevArray] = None) -> DevArray:
        """out[i*r+a] = bias[a] + sum_c matrix[a][c] * x[i*g+c] over groups of g = len(matrix[0]) consecutive
        elements (public r x g matrix; np_aes affine layer, np_from_bits)."""
        r, g = len(matrix), len(matrix[0])
        if x.n % g:
            raise ValueError('array length is not a multiple of the group size')
        ng = x.n // g
        out = out or self.empty(ng * r)
        m = self._scalars([v for row in matrix for v in row])
        b = self._scalars(bias) if bias is not None else None
        _ffi.check(self._L.ffgpu_group_matvec(self._h, m, b, r, g, x.ptr, out.ptr, ng, self._stream()), 'group_matvec')
        return out

    def _workspace(self):
        """partial-sum workspace of dot / sum: one per stream (launches on different streams must not share it)"""
        wss = self.__dict__.setdefault('_ws', {})
        key = self._stream()
        ws = wss.get(key)
        if ws is None:
            ws = wss[key] = torch.empty(1024 * 16, dtype=torch.uint8, device=self.torch_device)
        return ws

    def dot(self, a: DevArray, b: DevArray) -> DevArray:
        """sum_i a[i]*b[i] as a 1-element device array (local part of runtime.in_prod)."""
        if a.n != b.n:
            raise ValueError('length mismatch')
        out = self.empty(1)
        _ffi.check(self._L.ffgpu_dot(self._h, a.ptr, b.ptr, out.ptr, self._workspace().data_ptr(), a.n,
                                     self._stream()), 'dot')
        return out

    def sum(self, a: DevArray) -> DevArray:
        out = self.empty(1)
        _ffi.check(self._L.ffgpu_sum(self._h, a.ptr, out.ptr, self._workspace().data_ptr(), a.n, self._stream()), 'sum')
        return out

    def _stage(self, nbytes: int) -> torch.Tensor:
        """grow-only pinned host staging buffer of this context (uint8)"""
        stages = self.__dict__.setdefault('_pinned_stage', {})
        key = self._stream()                              # one staging buffer per stream
        stage = stages.get(key)
        if stage is None or stage.numel() < nbytes:
            stage = stages[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return stage

    def download_bytes(self, t: torch.Tensor) -> np.ndarray:
        """Device tensor -> uint8 numpy VIEW of the pinned staging buffer (valid until the next staged
        transfer): a pinned copy runs at PCIe speed, a pageable one at a third of it."""
        flat = t.contiguous().view(torch.uint8).reshape(-1)
        stage = self._stage(flat.numel())
        stage[:flat.numel()].copy_(flat, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return stage[:flat.numel()].numpy()

    def upload_bytes(self, data, dtype: torch.dtype, shape) -> torch.Tensor:
        """Host bytes (any buffer: a message off the wire) -> device tensor of `dtype` / `shape`: ONE host copy into
        the pinned staging buffer, then an asynchronous copy at PCIe speed (a pageable source costs an extra pass
        and transfers at a fraction of that rate)."""
        src = np.frombuffer(data, dtype=np.uint8)
        nbytes = src.size
        stage = self._stage(nbytes)
        stage[:nbytes].numpy()[:] = src
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.torch_device)
        dev.copy_(stage[:nbytes], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()        # the staging buffer is reused by the next transfer
        return dev.view(dtype).reshape(shape)

    def shake128_streams(self, msgs: Sequence[bytes], out_len: int, threads: int = 0) -> List[torch.Tensor]:
        """SHAKE128(msg).digest(out_len) for every msg, expanded in parallel on host threads into pinned
        buffers (libffgpu's ffgpu_shake128_expand) and uploaded: the XOF streams of a PRSS call
        (thresha.py:255, one per subset key).  Returns uint8 device tensors."""
        msgs = [bytes(mg) for mg in msgs]
        k = len(msgs)
        if k == 0 or out_len == 0:
            return [torch.empty(0, dtype=torch.uint8, device=self.torch_device) for _ in msgs]
        # one grow-only pinned staging buffer per context (allocating pinned memory per call costs more than
        # the expansion itself), rows padded to 256 bytes
        pitch = (out_len + 255) // 256 * 256
        need = k * pitch
        stage = self._stage(need)
        keep = [ctypes.create_string_buffer(mg, max(len(mg), 1)) for mg in msgs]
        mp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in keep])
        ml = (ctypes.c_size_t * k)(*[len(mg) for mg in msgs])
        op = (ctypes.c_void_p * k)(*[stage.data_ptr() + j * pitch for j in range(k)])
        _ffi.check(self._L.ffgpu_shake128_expand(mp, ml, k, out_len, op, threads), 'shake128_expand')
        dev = stage[:need].to(self.torch_device, non_blocking=True)
        torch.cuda.current_stream().synchronize()          # the staging buffer is reused by the next call
        devs = [dev[j * pitch:j * pitch + out_len] for j in range(k)]
        return devs

    PRSS_SLICE_BYTES = 8 << 20        # per stream and slice: 10 ms of one host thread, 160 MB per half for 20 keys

    def prss_streamed(self, msgs: Sequence[bytes], d: int, l: int, weights: Sequence[int], n: int, mask_bits: int,
                      out: DevArray, accumulate: bool, slice_bytes: int = 0, threads: int = 0) -> DevArray:
        """The PRSS combination of `prss_combine` with the XOF streams SHAKE128(msg) produced on the fly
        (ffgpu_shake128_open / _squeeze): every stream is squeezed a slice (~slice_bytes) at a time on host threads into
        one half of a pinned staging buffer; the slice is uploaded and combined into its range of `out` on the device
        while the host squeezes the next slice into the other half.  Pinned memory: 2 * len(msgs) * slice_bytes."""
        msgs = [bytes(mg) for mg in msgs]
        k = len(msgs)
        slice_bytes = slice_bytes or self.PRSS_SLICE_BYTES
        per = d * l                                                  # XOF bytes per output element and stream
        step = max(256, slice_bytes // per // 256 * 256)             # elements per slice (aligned ranges of `out`)
        pitch = (step * per + 255) // 256 * 256
        stage = self._stage(2 * k * pitch)
        keep = [ctypes.create_string_buffer(mg, max(len(mg), 1)) for mg in msgs]
        mp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in keep])
        ml = (ctypes.c_size_t * k)(*[len(mg) for mg in msgs])
        handle = ctypes.c_void_p()
        _ffi.check(self._L.ffgpu_shake128_open(mp, ml, k, ctypes.byref(handle)), 'shake128_open')
        free = [None, None]                                          # event: the half has been uploaded
        try:
            for c, h0 in enumerate(range(0, n, step)):
                cnt = min(step, n - h0)
                half = c & 1
                base = stage.data_ptr() + half * k * pitch
                if free[half] is not None:
                    free[half].synchronize()
                op = (ctypes.c_void_p * k)(*[base + j * pitch for j in range(k)])
                _ffi.check(self._L.ffgpu_shake128_squeeze(handle, op, cnt * per, threads), 'shake128_squeeze')
                dev = torch.empty(k * pitch, dtype=torch.uint8, device=self.torch_device)
                dev.copy_(stage[half * k * pitch:(half + 1) * k * pitch], non_blocking=True)
                free[half] = torch.cuda.Event()
                free[half].record()
                self.prss_combine([dev[j * pitch:j * pitch + cnt * per] for j in range(k)], d, l, weights, cnt,
                                  mask_bits=mask_bits, out=DevArray(self, out.t[h0:h0 + cnt], cnt), accumulate=accumulate)
        finally:
            self._L.ffgpu_shake128_close(handle)
            for ev in free:
                if ev is not None:
                    ev.synchronize()                                 # the staging buffer is reused by the next call
        return out

    def prss_combine(self, streams: Sequence, d: int, l: int, weights: Sequence[int], n: int,
                     mask_bits: int = 0, out: Optional[DevArray] = None, accumulate: bool = False) -> DevArray:
        """out[h] (+)= sum_s sum_j draw_s[h*d+j] * weights[s*d+j]; streams are the raw XOF outputs, n*d*l
        bytes each: host bytes (uploaded as they are) or uint8 device tensors from shake128_streams
        (thresha.py:163-173, 201-217)."""
        ks = len(streams)
        if len(weights) != ks * d:
            raise ValueError('need ks*d weights')
        out = out or self.empty(n)
        devs = []
        for sbytes in streams:
            if isinstance(sbytes, torch.Tensor):
                if sbytes.numel() < n * d * l:
                    raise ValueError('XOF stream too short')
                devs.append(sbytes)
                continue
            if len(sbytes) < n * d * l:
                raise ValueError('XOF stream too short')
            a = np.frombuffer(sbytes, dtype=np.uint8, count=n * d * l)
            devs.append(torch.from_numpy(a.copy()).to(self.torch_device))
        ptrs = (ctypes.c_void_p * ks)(*[t.data_ptr() for t in devs])
        w = self._scalars(weights)
        _ffi.check(self._L.ffgpu_prss_combine(self._h, ptrs, ks, d, l, mask_bits, w, int(accumulate), out.ptr, n,
                                              self._stream()), 'prss_combine')
        return out

    def prss_chacha(self, keys40: Sequence[bytes], d: int, l: int, weights: Sequence[int], n: int, mask_bits: int = 0,
                    rounds: int = 20, out: Optional[DevArray] = None, accumulate: bool = False) -> DevArray:
        """The same combination with the draws expanded ON THE DEVICE from one ChaCha stream per subset key (production
        mode, ffgpu_prss_chacha): keys40[s] = 32-byte key + 8-byte nonce of stream s; a draw = l keystream bytes reduced
        by the reference's rule (thresha.py:234-266).  At most 32 streams and 64 weights per call."""
        ks = len(keys40)
        if len(weights) != ks * d or any(len(k_) != 40 for k_ in keys40):
            raise ValueError('need ks 40-byte stream keys and ks*d weights')
        out = out or self.empty(n)
        w = self._scalars(weights)
        kb = ctypes.create_string_buffer(b''.join(bytes(k_) for k_ in keys40), 40 * ks)
        _ffi.check(self._L.ffgpu_prss_chacha(self._h, ctypes.cast(kb, ctypes.POINTER(ctypes.c_uint8)), ks, d, l, mask_bits,
                                             rounds, w, int(accumulate), out.ptr, n, self._stream()), 'prss_chacha')
        return out

    def bit_affine(self, bits: DevArray, matrix: Sequence[Sequence[int]], bias: Optional[Sequence[int]] = None,
                   from_bits: bool = False, out: Optional[DevArray] = None) -> DevArray:
        """GF(2^n<=8): y = M bits + bias per group of 8 bit shares (np_aes.py:40-41); from_bits: also
        recompose sum_r 2^r y_r in the same pass (np_aes.py:42)."""
        if bits.n % 8:
            raise ValueError('bit array length must be a multiple of 8')
        ng = bits.n // 8
        out = out or self.empty(ng if from_bits else bits.n)
        m = (ctypes.c_uint64 * 128)()
        for i, v in enumerate(v for row in matrix for v in row):
            m[2 * i] = int(v)
        b = None
        if bias is not None:
            b = (ctypes.c_uint64 * 16)()
            for i, v in enumerate(bias):
                b[2 * i] = int(v)
        _ffi.check(self._L.ffgpu_gf256_bit_affine(self._h, m, b, 1 if from_bits else 0, bits.ptr, out.ptr, ng,
                                                  self._stream()), 'bit_affine')
        return out

    def _scalars(self, vals: Sequence[int]):
        """Host scalars in the C ABI's layout: scalar_limbs little-endian 64-bit limbs each (ffgpu_ctx_scalar_limbs)."""
        sl = self.scalar_limbs
        w = (ctypes.c_uint64 * (sl * max(1, len(vals))))()
        for i, v in enumerate(vals):
            v = int(v)
            for q in range(sl):
                w[sl * i + q] = (v >> (64 * q)) & _MASK64
        return w

    def gf256_mask_open(self, rows: Sequence[DevArray], coefs: Sequence[int], rbits: Sequence[DevArray],
                        mus: Sequence[int], out: Optional[DevArray] = None) -> DevArray:
        """GF(2^8): c = sum_r coefs[r] * rows[r] + sum_p mus[p] * from_bits(rbits[p]) -- the opened masked value of
        np_to_bits (runtime.py:4414-4421) in one pass (ffgpu_gf256_mask_open)."""
        if len(rows) != len(coefs) or len(rbits) != len(mus) or not (rows or rbits):
            raise ValueError('one coefficient per row')
        n = rows[0].n if rows else rbits[0].n // 8
        self._same(n, *rows, out, what='share row')
        self._same(8 * n, *rbits, what='bit-share row')
        out = out or self.empty(n)
        pr = (ctypes.c_void_p * max(1, len(rows)))(*[r.ptr for r in rows])
        pb = (ctypes.c_void_p * max(1, len(rbits)))(*[r.ptr for r in rbits])
        _ffi.check(self._L.ffgpu_gf256_mask_open(self._h, pr, self._scalars(coefs), len(rows), pb, self._scalars(mus),
                                                 len(rbits), out.ptr, n, self._stream()), 'gf256_mask_open')
        return out

    def gf256_bits_affine_fold(self, c: DevArray, rbits: DevMatrix, matrix: Sequence[Sequence[int]],
                               bias: Optional[Sequence[int]] = None, out: Optional[DevMatrix] = None) -> DevMatrix:
        """GF(2^8), all parties in one launch: out[y] = from_bits(M (bits(c) + rbits[y]) + bias) for every row y of
        the (parties, 8n) matrix of bit shares (runtime.py:4422-4423 + np_aes.py:40-42; ffgpu_gf256_bits_affine_fold)."""
        n = c.n
        if rbits.n != 8 * n:
            raise ValueError('need 8 bit shares per byte')
        out = self._out_matrix(out, rbits.rows, n)
        mm = self._scalars([v for row in matrix for v in row])
        bb = self._scalars(bias) if bias is not None else None
        _ffi.check(self._L.ffgpu_gf256_bits_affine_fold(self._h, mm, bb, c.ptr, rbits.ptr, rbits.stride, out.ptr, out.stride,
                                                        n, rbits.rows, self._stream()), 'gf256_bits_affine_fold')
        return out

    def gf256_sbox_layer(self, X: DevMatrix, R: DevMatrix, t: int, lam: Sequence[int], mu: Sequence[int],
                         matrix: Sequence[Sequence[int]], bias: Optional[Sequence[int]] = None, key: Optional[bytes] = None,
                         nonce: int = 0, rounds: int = 20, state: Optional['RngState'] = None,
                         out: Optional[DevMatrix] = None) -> DevMatrix:
        """The whole secure S-box layer of np_aes.py:37-43 for all parties in ONE launch (ffgpu_gf256_sbox_layer): X
        = the parties' shares (one row each), R = their shares of 8 random bits per byte.  Raises
        NotImplementedError for shapes the fused kernel does not cover (the caller composes the layer from the
        per-step kernels then)."""
        import secrets as _secrets
        m, n = X.rows, X.n
        if R.rows != m or R.n != 8 * n:
            raise ValueError('need 8 bit shares per byte and party')
        out = self._out_matrix(out, m, n)
        mm = self._scalars([v for row in matrix for v in row])
        bb = self._scalars(bias) if bias is not None else None
        if state is None and key is None:
            key = _secrets.token_bytes(32)
        defer = 0
        if state is not None:
            if state.pending:
                nonce, defer = state.take_offset(), 1          # inside a sequence of deferred launches: commit() follows
            else:
                nonce = 0          # the layer is ONE launch: it advances the device nonce itself (the last workgroup of the
                #                    kernel for grids of up to RNG_RELEASE_MAX_GRID = 512 workgroups -- up to ~5.2 * 10^5 bytes;
                #                    10^6 bytes are 977 workgroups --, a one-thread kernel, k_rng_advance, above that)
        _ffi.check(self._L.ffgpu_gf256_sbox_layer(self._h, mm, bb, self._scalars(lam), self._scalars(mu), t, m, X.ptr, X.stride,
                                                  R.ptr, R.stride, out.ptr, out.stride, n, key, nonce, rounds,
                                                  state.ptr if state is not None else None, defer, self._stream()),
                   'gf256_sbox_layer')
        return out

    def to_bits(self, x: DevArray, addend: Optional[DevArray] = None, out: Optional[DevArray] = None) -> DevArray:
        """GF(2^n<=8): bits of PUBLIC bytes as field elements (8 per byte), plus `addend` (runtime.py:4418-4423)."""
        out = out or self.empty(8 * x.n)
        _ffi.check(self._L.ffgpu_gf256_to_bits(self._h, x.ptr, addend.ptr if addend is not None else None, out.ptr,
                                               x.n, self._stream()), 'to_bits')
        return out

    def sbox(self, x: DevArray, rows8: Sequence[int], b: int, out=None):
        out = out or self.empty(x.n)
        r = (ctypes.c_uint8 * 8)(*rows8)
        _ffi.check(self._L.ffgpu_gf256_sbox(self._h, x.ptr, r, ctypes.c_uint8(b), out.ptr, x.n, self._stream()),
                   'sbox')
        return out

    def set_timing(self, enable: bool = True, accumulate: bool = False):
        """Record GPU events around every compute call (off by default).  accumulate: keep one event pair per call
        so that busy_ms() can return the summed GPU time of all calls (ffgpu_busy_ms)."""
        _ffi.check(self._L.ffgpu_ctx_set_timing(self._h, (2 if accumulate else 1) if enable else 0), 'set_timing')

    def busy_ms(self, reset: bool = True):
        """(summed GPU ms, number of compute calls) since the last reset; waits for the calls issued so far.  Needs
        set_timing(True, accumulate=True)."""
        ms, calls = ctypes.c_double(), ctypes.c_ulonglong()
        _ffi.check(self._L.ffgpu_busy_ms(self._h, ctypes.byref(ms), ctypes.byref(calls), int(reset)), 'busy_ms')
        return float(ms.value), int(calls.value)

    def last_kernel_ms(self) -> float:
        """GPU time of the most recent compute call (waits for it); needs set_timing(True)."""
        ms = ctypes.c_float()
        _ffi.check(self._L.ffgpu_last_kernel_ms(self._h, ctypes.byref(ms)), 'last_kernel_ms')
        return float(ms.value)

    def sync(self):
        _ffi.check(self._L.ffgpu_stream_sync(self._h, self._stream()), 'sync')

    # ---- timing (HIP events on the launch stream, inside the library) --------
    def time_mul(self, a, b, out, reps: int) -> float:
        ms = ctypes.c_float()
        _ffi.check(self._L.ffgpu_time_mul(self._h, a.ptr, b.ptr, out.ptr, a.n, reps, self._stream(),
                                          ctypes.byref(ms)), 'time_mul')
        return ms.value

    def time_split(self, secrets, coeffs, t, m, out, reps: int) -> float:
        ms = ctypes.c_float()
        _ffi.check(self._L.ffgpu_time_split(self._h, secrets.ptr, coeffs.ptr if t else None,
                                            coeffs.stride if t else 0, t, m, out.ptr, out.stride, secrets.n,
                                            reps, self._stream(), ctypes.byref(ms)), 'time_split')
        return ms.value

    def time_recombine(self, rows, lambdas, out, reps: int, w: int = 1) -> float:
        k, ptrs, lam = self._rec_args(rows, lambdas, w)
        ms = ctypes.c_float()
        stride = rows[0].n if w == 1 else out.stride
        _ffi.check(self._L.ffgpu_time_recombine(self._h, ptrs, lam, k, w, out.ptr, stride, rows[0].n, reps,
                                                self._stream(), ctypes.byref(ms)), 'time_recombine')
        return ms.value

    def copy(self, src: torch.Tensor, dst: torch.Tensor):
        """Streaming device copy with the library's own kernel (bandwidth yardstick)."""
        nbytes = src.numel() * src.element_size()
        _ffi.check(self._L.ffgpu_copy(self._h, src.data_ptr(), dst.data_ptr(), nbytes, self._stream()), 'copy')

    def valu_probe(self, op: int = 0, iters: int = 2000, waves_per_simd: int = 4):
        """(lane-operations per second, shader clock in MHz, shader cycles per wave instruction and SIMD) the integer VALU
        sustains right now for one instruction kind (0: v_bitop3_b32, 1: v_add_u32, 2: v_mad_u64_u32): the compute-side
        yardstick beside `copy` (ffgpu_valu_probe)."""
        scratch = torch.zeros(4, dtype=torch.int64, device=self.torch_device)
        out = (ctypes.c_double * 3)()
        _ffi.check(self._L.ffgpu_valu_probe(self._h, op, iters, waves_per_simd, scratch.data_ptr(), out, self._stream()), 'valu_probe')
        return float(out[0]), float(out[1]), float(out[2])

    def time_copy(self, src: torch.Tensor, dst: torch.Tensor, reps: int) -> float:
        ms = ctypes.c_float()
        nbytes = src.numel() * src.element_size()
        _ffi.check(self._L.ffgpu_time_copy(self._h, src.data_ptr(), dst.data_ptr(), nbytes, reps,
                                           self._stream(), ctypes.byref(ms)), 'time_copy')
        return ms.value
