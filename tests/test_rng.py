"""Device CSPRNG (mpyc_amd/csrc/rng.hpp): ChaCha block function against the RFC 8439 test vector,
the field sampler and keystream layout against an independent restatement (oracle/fforacle.c) and
against Python integers.  CPU-only (device code compiled with g++ by the hostcheck harness)."""
import ctypes
import struct

import numpy as np
import pytest

from oracle.coracle import elem_bytes
from fieldutil import field_of, unpack

KEY = bytes(range(32))


def rfc8439_block_vector():
    # RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00
    w = [1, 0x09000000, 0x4a000000, 0x00000000]
    expect = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
              0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    return w, expect


def test_chacha_rfc8439_kat(hostcheck, coracle):
    w, expect = rfc8439_block_vector()
    assert coracle.chacha_block(KEY, w, 20) == expect                      # oracle restatement
    key = (ctypes.c_uint32 * 8).from_buffer_copy(KEY)
    out = (ctypes.c_uint32 * 16)()
    hostcheck.hc_chacha_block(key, (ctypes.c_uint32 * 4)(*w), 20, out)     # device code, host-compiled
    assert list(out) == expect
    for rounds in (8, 12):
        hostcheck.hc_chacha_block(key, (ctypes.c_uint32 * 4)(*w), rounds, out)
        assert list(out) == coracle.chacha_block(KEY, w, rounds)


def hc_coeffs(hostcheck, F, key, nonce, rounds, t, n):
    eb = elem_bytes(F.modulus, F.binary)
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64, 16: np.uint64}[eb]
    out = np.zeros((t, n, 2) if eb == 16 else (t, n), dtype=dt)
    lim = (ctypes.c_uint64 * 3)(*[(F.modulus >> (64 * i)) & (2**64 - 1) for i in range(3)])
    rc = hostcheck.hc_rng_coeffs(int(F.binary), lim, 3, key, ctypes.c_uint64(nonce), rounds, t,
                                 out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_size_t(n))
    assert rc == 0
    return out


def test_sampler_and_layout_match_oracle(hostcheck, coracle, golden_fields):
    for name, case in golden_fields.items():
        F = field_of(case)
        cf = coracle.CField(F.modulus, F.binary)
        for (t, n, rounds, nonce) in [(1, 37, 20, 0), (3, 50, 20, 0x1122334455667788), (4, 33, 12, 7), (2, 16, 8, 1),
                                      (6, 21, 20, 0xffffffff00000005)]:
            got = hc_coeffs(hostcheck, F, KEY, nonce, rounds, t, n)
            want = coracle.rng_coeffs(cf, KEY, nonce, rounds, t, n)
            assert (got == want).all(), (name, t, n)
            vals = unpack(got.reshape(-1, 2) if cf.eb == 16 else got.reshape(-1), cf.eb)
            assert all(0 <= v < F.order for v in vals), name


def test_sampler_is_wide_sample_mod_p(hostcheck, coracle):
    """First coefficient of the first pack for P64: (128 keystream bits) mod p, by hand."""
    from oracle import pyoracle as po
    p = 2**64 - 189
    F = po.Field(p)
    blk = coracle.chacha_block(KEY, [0, 0, 5, 0], 20)
    wide = blk[0] | (blk[1] << 32) | (blk[2] << 64) | (blk[3] << 96)
    got = hc_coeffs(hostcheck, F, KEY, 5, 20, 1, 2)
    assert int(got[0, 0]) == wide % p
    wide2 = blk[4] | (blk[5] << 32) | (blk[6] << 64) | (blk[7] << 96)
    assert int(got[0, 1]) == wide2 % p


def test_uniformity_smoke(hostcheck):
    """Cheap sanity check: mean and bit balance of 2^16 samples of GF(2^61-1)."""
    from oracle import pyoracle as po
    F = po.Field(2**61 - 1)
    c = hc_coeffs(hostcheck, F, KEY, 99, 20, 1, 65536)[0]
    mean = float(c.astype(np.float64).mean()) / (2**61)
    assert 0.49 < mean < 0.51
    for bit in (0, 17, 40, 60):
        frac = float(((c >> np.uint64(bit)) & np.uint64(1)).mean())
        assert 0.48 < frac < 0.52, bit
