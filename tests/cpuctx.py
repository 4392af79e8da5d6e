"""TEST INFRASTRUCTURE ONLY: a Python-integer stand-in for engine.FieldContext.

The build container has no GPU, the GPU box has no reference checkout.  To exercise the HOST logic of the
mirror (mpyc_amd/finfields.py, thresha.py, install()) under the real mpyc runtime in the build container,
the `-m "not gpu"` tests swap this context in for the device one (monkeypatching
mpyc_amd.finfields._context).  It computes with the oracle's Python-int functions on CPU torch tensors and
is never importable from the product (it lives in tests/ and imports oracle/); on a GPU box the same tests
run against libffgpu (tests/test_mpyc_dropin.py).
"""
from __future__ import annotations

import hashlib
import secrets as _secrets

import numpy as np
import torch

from mpyc_amd import engine
from mpyc_amd.engine import DevArray, DevMatrix, ints_to_np, limbs_of
from oracle import pyoracle as po
from oracle.coracle import elem_bytes


class CpuFieldContext(engine.FieldContext):
    def __init__(self, modulus, binary=False, device=None):   # noqa: super().__init__ loads libffgpu
        self.modulus, self.binary, self.device = int(modulus), bool(binary), 0
        self.F = po.Field(self.modulus, self.binary)
        self.elem_bytes = elem_bytes(self.modulus, self.binary)
        self.limbs = limbs_of(self.elem_bytes)
        self.scalar_limbs = 3 if self.elem_bytes == 24 else 2
        self.reduction = 'cpu-emulation'
        self.order = self.F.order
        self._h = None

    torch_device = torch.device('cpu')

    def _stream(self):
        return 0

    # ---- helpers ----
    def _put(self, out, vals):
        src = self.from_numpy(ints_to_np(list(vals), self.elem_bytes)) if len(vals) else self.empty(0)
        out.t.copy_(src.t.reshape(out.t.shape))
        return out

    def _map(self, fn, out, *arrs):
        n = arrs[0].n
        cols = [a.to_ints() for a in arrs]
        out = out or self.empty(n)
        return self._put(out, [fn(*xs) for xs in zip(*cols)])

    def add(self, a, b, out=None):
        self._chk(a, b)
        return self._map(lambda x, y: po.add(self.F, x, y), out, a, b)

    def sub(self, a, b, out=None):
        self._chk(a, b)
        return self._map(lambda x, y: po.sub(self.F, x, y), out, a, b)

    def mul(self, a, b, out=None):
        self._chk(a, b)
        return self._map(lambda x, y: po.mul(self.F, x, y), out, a, b)

    @staticmethod
    def _chk(a, b):
        if a.n != b.n:
            raise ValueError('length mismatch')

    def neg(self, a, out=None):
        return self._map(lambda x: po.neg(self.F, x), out, a)

    def reduce(self, raw, out=None):
        return self._map(lambda x: po.reduce(self.F, x), out, raw)

    def add_scalar(self, a, s, out=None):
        return self._map(lambda x: po.add(self.F, x, s), out, a)

    def mul_scalar(self, a, s, out=None):
        return self._map(lambda x: po.mul(self.F, x, s), out, a)

    def rsub_scalar(self, a, s, out=None):
        return self._map(lambda x: po.sub(self.F, s, x), out, a)

    def muladd(self, a, b, c, out=None):
        return self._map(lambda x, y, z: po.add(self.F, po.mul(self.F, x, y), z), out, a, b, c)

    def _pow1(self, x, e):
        r = 1
        while e:
            if e & 1:
                r = po.mul(self.F, r, x)
            x = po.mul(self.F, x, x)
            e >>= 1
        return r

    def pow(self, a, e, out=None):
        return self._map(lambda x: self._pow1(x, e), out, a)

    def inv(self, a, out=None, check_zero=True):
        vals = a.to_ints()
        if any(v == 0 for v in vals):
            if check_zero:
                raise ZeroDivisionError('inverse of 0 does not exist')
        out = out or self.empty(a.n)
        return self._put(out, [po.inv(self.F, v) if v else 0 for v in vals])

    def sqrt_cl(self, a, out=None):
        return self._map(lambda x: po.sqrt_prime(self.F, x), out, a)

    # ---- sharing ----
    def _split_host(self, s, C, t, m, out):
        n = len(s)
        draws = [v for row in C for v in row]
        rows = po.np_random_split(self.F, s, t, m, draws) if n else [[] for _ in range(m)]
        out = out or self.empty_matrix(m, n)
        for i in range(m):
            self._put(out.row(i), rows[i])
        return out

    def split(self, secrets, coeffs, t, m, out=None, mul_by=None):
        s = secrets.to_ints()
        if mul_by is not None:
            s = [po.mul(self.F, x, y) for x, y in zip(s, mul_by.to_ints())]
        C = [coeffs.row(j).to_ints() for j in range(t)] if t else []
        return self._split_host(s, C, t, m, out)

    def split_rng(self, secrets, t, m, key=None, nonce=0, rounds=20, out=None, mul_by=None, state=None):
        s = secrets.to_ints()
        if mul_by is not None:
            s = [po.mul(self.F, x, y) for x, y in zip(s, mul_by.to_ints())]
        C = [[_secrets.randbelow(self.order) for _ in s] for _ in range(t)]
        return self._split_host(s, C, t, m, out)

    def rng_state(self, key=None, nonce=0, rounds=20):
        return engine.RngState(self, torch.zeros(64, dtype=torch.uint8))

    def _rec_host(self, rows, lam):
        cols = [r.to_ints() for r in rows]
        return [po._dot(self.F, lam, xs) for xs in zip(*cols)]

    def gate(self, rows_a, lam_a, rows_b, lam_b, t, m, key=None, nonce=0, rounds=20, state=None, out=None):
        A = self._rec_host(rows_a, lam_a)
        B = self._rec_host(rows_b, lam_b) if rows_b else A
        s = [po.mul(self.F, x, y) for x, y in zip(A, B)]
        C = [[_secrets.randbelow(self.order) for _ in s] for _ in range(t)]
        return self._split_host(s, C, t, m, out)

    def recombine(self, rows, lambdas, w=1, out=None):
        k = len(rows)
        if len(lambdas) != w * k:
            raise ValueError('need w*k lambda values')
        n = rows[0].n
        if any(r.n != n for r in rows):
            raise ValueError('share rows of different lengths')
        if w == 1:
            out = out or self.empty(n)
            return self._put(out, self._rec_host(rows, list(lambdas)))
        out = out or self.empty_matrix(w, n)
        for r in range(w):
            self._put(out.row(r), self._rec_host(rows, list(lambdas[r * k:(r + 1) * k])))
        return out

    # ---- linear algebra ----
    def matmul(self, A, B, M, K, N, out=None):
        a, b = A.to_ints(), B.to_ints()
        Am = [a[i * K:(i + 1) * K] for i in range(M)]
        Bm = [b[i * N:(i + 1) * N] for i in range(K)]
        C = po.matmul(self.F, Am, Bm) if M and N and K else [[0] * N for _ in range(M)]
        out = out or self.empty(M * N)
        return self._put(out, [v for row in C for v in row])

    def gauss(self, a, n, ncols, batch=1, det=False):
        vals = a.to_ints()
        sing = torch.zeros(max(batch, 1), dtype=torch.int32)
        dets, res = [], []
        for bi in range(batch):
            blk = vals[bi * n * ncols:(bi + 1) * n * ncols]
            Mx = [blk[i * ncols:(i + 1) * ncols] for i in range(n)]
            if det:
                dets.append(po.gauss_det(self.F, [r[:n] for r in Mx]))
                res += blk
            else:
                try:
                    X = po.gauss_solve(self.F, [r[:n] for r in Mx], [r[n:] for r in Mx])
                    for i in range(n):
                        res += Mx[i][:n] + X[i]
                except ZeroDivisionError:
                    sing[bi] = 1
                    res += blk
        self._put(a, res)
        d = self._put(self.empty(batch), dets) if det else None
        return d, sing

    def group_matvec(self, x, matrix, bias=None, out=None):
        r, g = len(matrix), len(matrix[0])
        if x.n % g:
            raise ValueError('array length is not a multiple of the group size')
        v = x.to_ints()
        res = []
        for i in range(x.n // g):
            grp = v[i * g:(i + 1) * g]
            for a in range(r):
                acc = po._dot(self.F, [po.reduce(self.F, c) for c in matrix[a]], grp)
                res.append(po.add(self.F, acc, bias[a]) if bias is not None else acc)
        out = out or self.empty(len(res))
        return self._put(out, res)

    def dot(self, a, b):
        self._chk(a, b)
        return self._put(self.empty(1), [po._dot(self.F, a.to_ints(), b.to_ints())])

    def sum(self, a):
        acc = 0
        for v in a.to_ints():
            acc = po.add(self.F, acc, v)
        return self._put(self.empty(1), [acc])

    def download_bytes(self, t):
        return t.contiguous().view(torch.uint8).reshape(-1).numpy()

    def upload_bytes(self, data, dtype, shape):
        return torch.frombuffer(bytearray(data), dtype=torch.uint8).view(dtype).reshape(shape)

    def shake128_streams(self, msgs, out_len, threads=0):
        assert all(type(mg) is bytes for mg in msgs), 'the C entry point takes bytes (ctypes refuses a bytearray)'
        return [torch.frombuffer(bytearray(hashlib.shake_128(mg).digest(out_len)), dtype=torch.uint8) if out_len
                else torch.empty(0, dtype=torch.uint8) for mg in msgs]

    def prss_combine(self, streams, d, l, weights, n, mask_bits=0, out=None, accumulate=False):
        out = out or self.empty(n)
        acc = out.to_ints() if accumulate else [0] * n
        bound = (1 << mask_bits) if mask_bits else self.order
        for s, sb in enumerate(streams):
            raw = bytes(sb.numpy().tobytes()) if isinstance(sb, torch.Tensor) else bytes(sb)
            for h in range(n):
                for j in range(d):
                    o = (h * d + j) * l
                    x = po.reduce(self.F, int.from_bytes(raw[o:o + l], 'little') % bound)
                    acc[h] = po.add(self.F, acc[h], po.mul(self.F, x, weights[s * d + j]))
        return self._put(out, acc)

    def prss_chacha(self, keys40, d, l, weights, n, mask_bits=0, rounds=20, out=None, accumulate=False):
        out = out or self.empty(n)
        acc = out.to_ints() if accumulate else [0] * n
        bound = (1 << mask_bits) if mask_bits else self.order
        tb, dpt = po.prss_chacha_layout(l)
        lw = (l + 3) // 4
        for s, k40 in enumerate(keys40):
            for h in range(n):
                tile, slot = divmod(h, dpt)
                for j in range(d):
                    ksb = b''.join(po.chacha_block(k40[:32], (tile * d + j) * tb + b, k40[32:], rounds) for b in range(tb))
                    x = po.reduce(self.F, int.from_bytes(ksb[4 * slot * lw:4 * slot * lw + l], 'little') % bound)
                    acc[h] = po.add(self.F, acc[h], po.mul(self.F, x, weights[s * d + j]))
        return self._put(out, acc)

    def bit_affine(self, bits, matrix, bias=None, from_bits=False, out=None):
        y = self.group_matvec(bits, matrix, bias)
        if not from_bits:
            return y if out is None else self._put(out, y.to_ints())
        z = self.group_matvec(y, [[1 << r for r in range(8)]])
        return z if out is None else self._put(out, z.to_ints())

    def to_bits(self, x, addend=None, out=None):
        v = x.to_ints()
        bits = [(b >> r) & 1 for b in v for r in range(8)]
        if addend is not None:
            bits = [po.add(self.F, a, b) for a, b in zip(bits, addend.to_ints())]
        out = out or self.empty(8 * x.n)
        return self._put(out, bits)

    def sbox(self, x, rows8, b, out=None):
        out = out or self.empty(x.n)
        return self._put(out, po.sbox(x.to_ints(), rows8, b))

    def sync(self):
        pass


def use_cpu_contexts(monkeypatch=None):
    """Route mpyc_amd.finfields._context to CpuFieldContext (tests only)."""
    import mpyc_amd.finfields as gff
    cache = {}

    def _context(field, device=None):
        ctx = cache.get(field)
        if ctx is None:
            ops = gff._fops(field)
            ctx = cache[field] = CpuFieldContext(ops.modulus, binary=ops.binary)
        return ctx
    if monkeypatch is not None:
        monkeypatch.setattr(gff, '_context', _context)
    else:
        gff._context = _context
    import mpyc_amd.thresha as gth
    if hasattr(gth, '_context'):
        if monkeypatch is not None:
            monkeypatch.setattr(gth, '_context', _context)
        else:
            gth._context = _context
    return _context
