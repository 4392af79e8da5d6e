"""ctypes binding of libffgpu.so (C ABI in include/ffgpu.h).

This is the only place the shared library is loaded.  There is no fallback: if the
library is missing or a call fails, an exception is raised (FfgpuError or the
Python exception the reference raises for the same condition).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libffgpu.so')

OK, EINVAL, ENOTSUP, EHIP, EMODULUS, ENOMEM = range(6)
PRIME, BINARY = 1, 2
RED_NAMES = {1: 'pseudo-mersenne', 2: 'reciprocal', 3: 'gf2-swar', 4: 'gf2-wide', 5: 'montgomery'}


class FfgpuError(RuntimeError):
    pass


_lib = None

_vp, _sz, _int = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
_u64p = ctypes.POINTER(ctypes.c_uint64)
_fp = ctypes.POINTER(ctypes.c_float)

# name -> argtypes; every function returns int status unless listed in _RESTYPES
_SIGS = {
    'ffgpu_abi_version': [],
    'ffgpu_strerror': [_int],
    'ffgpu_last_hip_error': [],
    'ffgpu_device_count': [ctypes.POINTER(_int)],
    'ffgpu_device_pci_bus_id': [_int, ctypes.c_char_p, _int],
    'ffgpu_ctx_create': [_int, _u64p, _int, _int, ctypes.POINTER(_vp)],
    'ffgpu_ctx_destroy': [_vp],
    'ffgpu_ctx_set_timing': [_vp, _int],
    'ffgpu_last_kernel_ms': [_vp, ctypes.POINTER(ctypes.c_float)],
    'ffgpu_busy_ms': [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong), _int],
    'ffgpu_ctx_elem_bytes': [_vp],
    'ffgpu_ctx_reduction': [_vp],
    'ffgpu_ctx_scalar_limbs': [_vp],
    'ffgpu_ctx_device': [_vp],
    'ffgpu_malloc': [_vp, _sz, ctypes.POINTER(_vp)],
    'ffgpu_free': [_vp, _vp],
    'ffgpu_h2d': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_d2h': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_stream_sync': [_vp, _vp],
    'ffgpu_reduce': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_add': [_vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_sub': [_vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_mul': [_vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_neg': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_add_scalar': [_vp, _vp, _u64p, _vp, _sz, _vp],
    'ffgpu_mul_scalar': [_vp, _vp, _u64p, _vp, _sz, _vp],
    'ffgpu_rsub_scalar': [_vp, _vp, _u64p, _vp, _sz, _vp],
    'ffgpu_muladd': [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_pow': [_vp, _vp, _u64p, _int, _vp, _sz, _vp],
    'ffgpu_inv': [_vp, _vp, _vp, _sz, _vp, _vp],
    'ffgpu_beaver_combine': [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _sz, _vp],
    'ffgpu_split': [_vp, _vp, _vp, _sz, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_mul_split': [_vp, _vp, _vp, _vp, _sz, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_rng_coeffs': [_vp, ctypes.c_char_p, ctypes.c_uint64, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_split_rng': [_vp, _vp, ctypes.c_char_p, ctypes.c_uint64, _int, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_mul_split_rng': [_vp, _vp, _vp, ctypes.c_char_p, ctypes.c_uint64, _int, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_gate_rng': [_vp, ctypes.POINTER(_vp), _u64p, _int, ctypes.POINTER(_vp), _u64p, _int, ctypes.c_char_p,
                       ctypes.c_uint64, _int, _vp, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_gate_rng_batch': [_vp, ctypes.POINTER(_vp), _u64p, _int, _sz, ctypes.POINTER(_vp), _u64p, _int, _sz, ctypes.c_char_p,
                             ctypes.c_uint64, _int, _vp, _int, _int, _int, _vp, _sz, _sz, _sz, _int, _vp],
    'ffgpu_rng_state_advance': [_vp, _vp, ctypes.c_uint32, _vp],
    'ffgpu_rng_state_bytes': [],
    'ffgpu_rng_state_init': [_vp, _vp, ctypes.c_char_p, ctypes.c_uint64, _int, _vp],
    'ffgpu_split_rng_state': [_vp, _vp, _vp, _vp, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_recombine': [_vp, ctypes.POINTER(_vp), _u64p, _int, _int, _vp, _sz, _sz, _vp],
    'ffgpu_matmul': [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _sz, _sz, _sz, _vp],
    'ffgpu_sqrt_cl': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_gauss': [_vp, _vp, _int, _int, _sz, _int, _vp, _vp, _vp],
    'ffgpu_group_matvec': [_vp, _u64p, _u64p, _int, _int, _vp, _vp, _sz, _vp],
    'ffgpu_dot': [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_sum': [_vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_shake128_backend': [],
    'ffgpu_shake128_expand': [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), _int, _sz,
                              ctypes.POINTER(ctypes.c_void_p), _int],
    'ffgpu_shake128_open': [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), _int, ctypes.POINTER(_vp)],
    'ffgpu_shake128_squeeze': [_vp, ctypes.POINTER(ctypes.c_void_p), _sz, _int],
    'ffgpu_shake128_close': [_vp],
    'ffgpu_prss_combine': [_vp, ctypes.POINTER(_vp), _int, _int, _int, _int, _u64p, _int, _vp, _sz, _vp],
    'ffgpu_prss_chacha': [_vp, ctypes.POINTER(ctypes.c_uint8), _int, _int, _int, _int, _int, _u64p, _int, _vp, _sz, _vp],
    'ffgpu_prss_chacha_layout': [_int, ctypes.POINTER(_int), ctypes.POINTER(_int)],
    'ffgpu_gf256_bit_affine': [_vp, _u64p, _u64p, _int, _vp, _vp, _sz, _vp],
    'ffgpu_gf256_to_bits': [_vp, _vp, _vp, _vp, _sz, _vp],
    'ffgpu_gf256_mask_open': [_vp, ctypes.POINTER(_vp), _u64p, _int, ctypes.POINTER(_vp), _u64p, _int, _vp, _sz, _vp],
    'ffgpu_gf256_bits_affine_fold': [_vp, _u64p, _u64p, _vp, _vp, _sz, _vp, _sz, _sz, _int, _vp],
    'ffgpu_gf256_sbox_layer': [_vp, _u64p, _u64p, _u64p, _u64p, _int, _int, _vp, _sz, _vp, _sz, _vp, _sz, _sz, ctypes.c_char_p,
                               ctypes.c_uint64, _int, _vp, _int, _vp],
    'ffgpu_gf256_sbox': [_vp, _vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint8, _vp, _sz, _vp],
    'ffgpu_time_mul': [_vp, _vp, _vp, _vp, _sz, _int, _vp, _fp],
    'ffgpu_time_split': [_vp, _vp, _vp, _sz, _int, _int, _vp, _sz, _sz, _int, _vp, _fp],
    'ffgpu_time_recombine': [_vp, ctypes.POINTER(_vp), _u64p, _int, _int, _vp, _sz, _sz, _int, _vp, _fp],
    'ffgpu_time_copy': [_vp, _vp, _vp, _sz, _int, _vp, _fp],
    'ffgpu_valu_probe': [_vp, _int, _int, _int, _vp, ctypes.POINTER(ctypes.c_double), _vp],
    'ffgpu_copy': [_vp, _vp, _vp, _sz, _vp],
    'ffgpu_ipc_export': [_vp, _vp, _sz, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_char_p, _vp],
    'ffgpu_ipc_open': [_vp, ctypes.c_char_p, ctypes.POINTER(_vp)],
    'ffgpu_ipc_read': [_vp, _vp, ctypes.c_ulonglong, _vp, _sz, ctypes.c_char_p, _vp],
    'ffgpu_ipc_read_reduced': [_vp, _vp, ctypes.c_ulonglong, _vp, _sz, ctypes.c_char_p, _vp],
    'ffgpu_ipc_close': [_vp, _vp],
}
_RESTYPES = {'ffgpu_strerror': ctypes.c_char_p, 'ffgpu_last_hip_error': ctypes.c_char_p,
             'ffgpu_rng_state_bytes': ctypes.c_size_t, 'ffgpu_shake128_close': None}

EXPORTED = tuple(_SIGS)


def lib():
    """Load libffgpu.so (once).  Raises FfgpuError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FfgpuError(f'{LIB_PATH} not found: build it with `make -C mpyc_amd/csrc` '
                             '(or python -c "import __graft_entry__ as g; g.build()"); '
                             'there is no CPU fallback')
        # torch ships its own HIP runtime (SONAME libamdhip64.so.7); load it FIRST so that
        # libffgpu's DT_NEEDED resolves to the same runtime instance torch uses (one HIP
        # runtime per process: device pointers and streams are shared with torch).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)      # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, _int)
        if L.ffgpu_abi_version() != 1:
            raise FfgpuError('libffgpu ABI version mismatch')
        _lib = L
    return _lib


def check(rc: int, what: str = ''):
    if rc == OK:
        return
    L = lib()
    msg = L.ffgpu_strerror(rc).decode()
    if rc == EHIP:
        msg += ': ' + L.ffgpu_last_hip_error().decode()
    if rc == EINVAL:
        raise ValueError(f'ffgpu {what}: {msg}')
    if rc == EMODULUS:
        raise ValueError(f'ffgpu {what}: {msg}')
    if rc == ENOMEM:
        raise MemoryError(f'ffgpu {what}: {msg}')
    if rc == ENOTSUP:
        raise NotImplementedError(f'ffgpu {what}: {msg}')
    raise FfgpuError(f'ffgpu {what}: {msg}')


def limbs(x: int, n: int = 2):
    return (ctypes.c_uint64 * n)(*[(int(x) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)])
