"""Host SHAKE128 of libffgpu (PRSS key expansion, mpyc_amd/csrc/shake.hip) against hashlib -- the function
the reference calls (thresha.py:255).  No GPU needed."""
import ctypes
import hashlib
import random
import time

import numpy as np


def expand(L, msgs, out_len, threads):
    n = len(msgs)
    bufs = [ctypes.create_string_buffer(max(out_len, 1)) for _ in range(n)]
    keep = [ctypes.create_string_buffer(m, max(len(m), 1)) for m in msgs]
    mp = (ctypes.c_void_p * n)(*[ctypes.addressof(k) for k in keep])
    ml = (ctypes.c_size_t * n)(*[len(m) for m in msgs])
    op = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in bufs])
    rc = L.ffgpu_shake128_expand(mp, ml, n, out_len, op, threads)
    assert rc == 0
    return [b.raw[:out_len] for b in bufs]


def test_shake128_matches_hashlib():
    from mpyc_amd import _ffi
    L = _ffi.lib()
    rng = random.Random(202)
    # FIPS 202 known answer: SHAKE128 of the empty message
    assert expand(L, [b''], 32, 1)[0].hex() == '7f9c2ba4e88f827d616045507605853ed73b8093f6efbc88eb1a6eacfa66ef26'
    for mlen in (0, 1, 16, 23, 167, 168, 169, 335, 336, 337, 1000):
        msgs = [bytes(rng.randrange(256) for _ in range(mlen)) for _ in range(5)]
        for out_len in (0, 1, 24, 167, 168, 169, 336, 337, 5000):
            got = expand(L, msgs, out_len, rng.choice([1, 2, 5, 0]))
            assert got == [hashlib.shake_128(m).digest(out_len) for m in msgs], (mlen, out_len)


def test_shake128_many_keys_in_parallel():
    """C(7,3) = 35 subset keys with a long output each: identical to hashlib, and not slower than one thread."""
    from mpyc_amd import _ffi
    L = _ffi.lib()
    msgs = [bytes([i]) * 16 + b'uci-0042' for i in range(35)]
    out_len = 24 * 40_000
    t0 = time.perf_counter()
    one = expand(L, msgs, out_len, 1)
    t1 = time.perf_counter()
    par = expand(L, msgs, out_len, 0)
    t2 = time.perf_counter()
    assert one == par
    assert par[7] == hashlib.shake_128(msgs[7]).digest(out_len)
    assert np.frombuffer(par[3], dtype=np.uint8).std() > 60          # looks like random bytes
    assert (t2 - t1) < 2.0 * (t1 - t0) + 0.05


def test_shake128_resumable_streams():
    """ffgpu_shake128_open / _squeeze / _close (the sliced PRSS path, engine.prss_streamed): whatever the slice sizes --
    within a 168-byte block, across blocks, zero -- the concatenation is hashlib's stream, for every message length around
    the rate; and a closed / null handle is refused, not dereferenced."""
    from mpyc_amd import _ffi
    L = _ffi.lib()
    rng = random.Random(303)
    msgs = [bytes(rng.randrange(256) for _ in range(mlen)) for mlen in (0, 3, 19, 167, 168, 169, 336, 400)]
    k = len(msgs)
    keep = [ctypes.create_string_buffer(m, max(len(m), 1)) for m in msgs]
    mp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in keep])
    ml = (ctypes.c_size_t * k)(*[len(m) for m in msgs])
    h = ctypes.c_void_p()
    assert L.ffgpu_shake128_open(mp, ml, k, ctypes.byref(h)) == 0 and h.value
    outs = [bytearray() for _ in msgs]
    for nb in (1, 7, 160, 168, 169, 1000, 4096, 0, 333, 167, 1, 100_000):
        bufs = [np.zeros(max(nb, 1), dtype=np.uint8) for _ in msgs]
        op = (ctypes.c_void_p * k)(*[b.ctypes.data for b in bufs])
        assert L.ffgpu_shake128_squeeze(h, op, nb, rng.choice([1, 3, 0])) == 0
        for o, b in zip(outs, bufs):
            o += bytes(b[:nb])
    L.ffgpu_shake128_close(h)
    for m, o in zip(msgs, outs):
        assert bytes(o) == hashlib.shake_128(m).digest(len(o)), len(m)
    assert L.ffgpu_shake128_squeeze(None, None, 10, 1) != 0
    assert L.ffgpu_shake128_open(None, None, 1, ctypes.byref(h)) != 0
    L.ffgpu_shake128_close(None)
