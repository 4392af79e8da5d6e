"""Host SHAKE128 expansion rate of libffgpu (ffgpu_shake128_expand; libcrypto backend unless FFGPU_SHAKE_OWN=1) next to
hashlib: one stream, and 20 streams on 20 threads (the PRSS call of m = 7, t = 3).  Buffers are touched first so that
page faults of fresh allocations are not timed."""
import ctypes, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mpyc_amd import _ffi
L = _ffi.lib()
n = 56_000_000
for k in (1, 20):
    msgs = [bytes([j]) * 16 + b'uci' for j in range(k)]
    bufs = [np.zeros(n, dtype=np.uint8) for _ in range(k)]
    keep = [ctypes.create_string_buffer(m, len(m)) for m in msgs]
    mp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in keep])
    ml = (ctypes.c_size_t * k)(*[len(m) for m in msgs])
    op = (ctypes.c_void_p * k)(*[b.ctypes.data for b in bufs])
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.ffgpu_shake128_expand(mp, ml, k, n, op, k)
        best = min(best, time.perf_counter() - t0)
    assert rc == 0
    assert bytes(bufs[-1][:64]) == hashlib.shake_128(msgs[-1]).digest(64)
    t0 = time.perf_counter()
    hashlib.shake_128(msgs[0]).digest(n)
    dh = time.perf_counter() - t0
    print(f'backend {L.ffgpu_shake128_backend()} (1 = libcrypto, 0 = own Keccak): {k} stream(s) x {n} B on {k} thread(s): '
          f'{best*1e3:.1f} ms = {k*n/best/1e9:.2f} GB/s total, {n/best/1e9:.3f} GB/s per stream; hashlib one stream {n/dh/1e9:.3f} GB/s')
