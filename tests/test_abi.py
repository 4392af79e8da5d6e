"""C-ABI boundary checks that need no GPU: the library loads, exports exactly what
include/ffgpu.h declares, classifies moduli, and rejects bad arguments with status codes
(no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    from mpyc_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _ffi.lib()


def declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'ffgpu.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(ffgpu_[a-z0-9_]+)\s*\(', hdr)))


def test_header_symbols_exported(L):
    from mpyc_amd import _ffi
    decl = declared_symbols()
    assert len(decl) >= 30
    out = subprocess.run(['nm', '-D', '--defined-only', _ffi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (ffgpu_[a-z0-9_]+)', out))
    missing = [s for s in decl if s not in exported]
    assert not missing, f'declared in include/ffgpu.h but not exported: {missing}'
    assert sorted(_ffi.EXPORTED) == decl, 'python binding and header disagree'
    for s in decl:
        getattr(L, s)


def test_version_and_strerror(L):
    assert L.ffgpu_abi_version() == 1
    assert L.ffgpu_strerror(0) == b'ok'
    assert b'invalid' in L.ffgpu_strerror(1)


def mk(L, kind, modulus, nl=3):
    from mpyc_amd import _ffi
    h = ctypes.c_void_p()
    rc = L.ffgpu_ctx_create(kind, _ffi.limbs(modulus, nl), nl, 0, ctypes.byref(h))
    return rc, h


def test_ctx_classification(L):
    from mpyc_amd import _ffi
    cases = [
        (_ffi.PRIME, 2**61 - 1, 8, 1), (_ffi.PRIME, 2**64 - 189, 8, 1), (_ffi.PRIME, 2**128 - 173, 16, 1),
        (_ffi.PRIME, 2**127 - 1, 16, 1), (_ffi.PRIME, 2**96 - 17, 12, 1), (_ffi.PRIME, 2**80 - 65, 12, 1),
        (_ffi.PRIME, 2**97 - 141, 16, 1), (_ffi.PRIME, 19, 4, 2),
        (_ffi.PRIME, 2**31 - 1, 4, 2), (_ffi.PRIME, 6616326157076047771, 8, 2),
        (_ffi.PRIME, 258797994007609146293811961253269568351, 16, 5),
        (_ffi.PRIME, 2**136 - 113, 24, 1), (_ffi.PRIME, 2**135 + 4823, 24, 5), (_ffi.PRIME, 2**192 - 237, 24, 1),
        (_ffi.BINARY, 0x11b, 1, 3), (_ffi.BINARY, 0b111, 1, 3), (_ffi.BINARY, (1 << 64) | 0x1b, 8, 4),
        (_ffi.BINARY, (1 << 128) | 0x87, 16, 4),
    ]
    for kind, mod, eb, red in cases:
        rc, h = mk(L, kind, mod)
        assert rc == 0, (hex(mod), rc)
        assert L.ffgpu_ctx_elem_bytes(h) == eb, hex(mod)
        assert L.ffgpu_ctx_reduction(h) == red, hex(mod)
        assert L.ffgpu_ctx_device(h) == 0
        L.ffgpu_ctx_destroy(h)


def test_bad_arguments(L):
    from mpyc_amd import _ffi
    assert mk(L, _ffi.PRIME, 0)[0] == _ffi.EMODULUS
    assert mk(L, _ffi.PRIME, 1)[0] == _ffi.EMODULUS
    assert mk(L, _ffi.PRIME, 1 << 128)[0] == _ffi.EMODULUS         # even three-limb modulus
    assert mk(L, _ffi.PRIME, (1 << 192) + 7, nl=4)[0] != 0          # above 192 bits
    assert mk(L, _ffi.PRIME, (1 << 127) + 2**40)[0] == _ffi.EMODULUS  # even two-limb modulus
    assert mk(L, 7, 19)[0] == _ffi.EINVAL
    assert mk(L, _ffi.BINARY, 1)[0] == _ffi.EMODULUS
    rc, h = mk(L, _ffi.PRIME, 2**61 - 1)
    assert rc == 0
    # split: 0 <= t < m (thresha.py:26), null pointers, strides
    assert L.ffgpu_split(h, 16, 16, 8, 3, 3, 16, 8, 8, None) == _ffi.EINVAL
    assert L.ffgpu_split(h, None, 16, 8, 1, 3, 16, 8, 8, None) == _ffi.EINVAL
    assert L.ffgpu_split(h, 16, 16, 8, 1, 3, 16, 4, 8, None) == _ffi.EINVAL      # share_stride < n
    assert L.ffgpu_split(h, 16, None, 0, 0, 1, 16, 0, 0, None) == _ffi.OK        # n == 0 is a no-op
    assert L.ffgpu_mul(h, None, None, None, 0, None) == _ffi.OK
    assert L.ffgpu_mul(h, None, 16, 16, 4, None) == _ffi.EINVAL
    assert L.ffgpu_recombine(h, None, None, 0, 1, 16, 8, 8, None) == _ffi.EINVAL
    rows8 = (ctypes.c_uint8 * 8)()
    assert L.ffgpu_gf256_sbox(h, 16, rows8, 0, 16, 8, None) == _ffi.ENOTSUP    # not GF(2^8)
    L.ffgpu_ctx_destroy(h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mpyc_amd import _ffi
    monkeypatch.setattr(_ffi, '_lib', None)
    monkeypatch.setattr(_ffi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_ffi.FfgpuError):
        _ffi.lib()


def test_empty_inputs_are_ok_everywhere(L):
    """n = 0 (empty arrays, tests/test_thresha.py and finfields edge cases): every compute entry point
    returns FFGPU_OK before touching a device (so this runs without a GPU) and without reading pointers."""
    from mpyc_amd import _ffi
    rc, h = mk(L, _ffi.PRIME, 2**61 - 1)
    assert rc == 0
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    two = (ctypes.c_uint64 * 2)(1, 0)
    rows = (ctypes.c_void_p * 3)(p.value, p.value, p.value)
    lam = (ctypes.c_uint64 * 6)(1, 0, 1, 0, 1, 0)
    key = bytes(32)
    assert L.ffgpu_add(h, p, p, p, 0, None) == 0
    assert L.ffgpu_mul(h, p, p, p, 0, None) == 0
    assert L.ffgpu_neg(h, p, p, 0, None) == 0
    assert L.ffgpu_reduce(h, p, p, 0, None) == 0
    assert L.ffgpu_mul_scalar(h, p, two, p, 0, None) == 0
    assert L.ffgpu_muladd(h, p, p, p, p, 0, None) == 0
    assert L.ffgpu_pow(h, p, two, 1, p, 0, None) == 0
    assert L.ffgpu_inv(h, p, p, 0, p, None) == 0
    assert L.ffgpu_split(h, p, p, 0, 1, 3, p, 0, 0, None) == 0
    assert L.ffgpu_mul_split(h, p, p, p, 0, 1, 3, p, 0, 0, None) == 0
    assert L.ffgpu_split_rng(h, p, key, 0, 20, 1, 3, p, 0, 0, None) == 0
    assert L.ffgpu_recombine(h, rows, lam, 3, 1, p, 0, 0, None) == 0
    assert L.ffgpu_gate_rng(h, rows, lam, 3, None, None, 0, key, 0, 20, None, 1, 3, p, 0, 0, None) == 0
    assert L.ffgpu_group_matvec(h, lam, None, 1, 3, p, p, 0, None) == 0
    assert L.ffgpu_gauss(h, p, 3, 3, 0, 0, None, p, None) == 0
    assert L.ffgpu_shake128_expand(None, None, 0, 10, None, 1) == 0
    # and bad shapes are still rejected when n = 0
    assert L.ffgpu_split(h, p, p, 0, 3, 3, p, 0, 0, None) == _ffi.EINVAL      # t must be < m
    assert L.ffgpu_gate_rng(h, rows, lam, 3, None, None, 0, key, 0, 20, None, 4, 9, p, 0, 0, None) == _ffi.ENOTSUP
    L.ffgpu_ctx_destroy(h)
