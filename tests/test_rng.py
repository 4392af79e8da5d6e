"""Device CSPRNG (mpyc_amd/csrc/rng.hpp): ChaCha block function against the RFC 8439 test vector,
the field sampler and keystream layout against an independent restatement (oracle/fforacle.c) and
against Python integers.  CPU-only (device code compiled with g++ by the hostcheck harness)."""
import ctypes

import numpy as np
import pytest

from oracle.coracle import elem_bytes
from fieldutil import field_of, unpack, lshape

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
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64, 12: np.uint32, 16: np.uint64}[eb]
    out = np.zeros(lshape(eb, t, n), dtype=dt)
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
            vals = unpack(got.reshape(lshape(cf.eb, -1)), cf.eb)
            assert all(0 <= v < F.order for v in vals), name


def test_sampler_by_hand(hostcheck, coracle):
    """First coefficients of the first pack, by hand: P64 (pseudo-Mersenne -> rejection sampling: the
    64 keystream bits themselves unless >= p) and a generic 63-bit prime (128 bits mod p)."""
    from oracle import pyoracle as po
    blk = coracle.chacha_block(KEY, [0, 0, 5, 0], 20)
    p = 2**64 - 189
    got = hc_coeffs(hostcheck, po.Field(p), KEY, 5, 20, 1, 2)
    w0, w1 = blk[0] | (blk[1] << 32), blk[2] | (blk[3] << 32)
    assert w0 < p and w1 < p                       # (true for this key; rejection has probability 2^-56)
    assert int(got[0, 0]) == w0 and int(got[0, 1]) == w1
    g = 6616326157076047771
    got = hc_coeffs(hostcheck, po.Field(g), KEY, 5, 20, 1, 2)
    wide = blk[0] | (blk[1] << 32) | (blk[2] << 64) | (blk[3] << 96)
    assert int(got[0, 0]) == wide % g
    wide2 = blk[4] | (blk[5] << 32) | (blk[6] << 64) | (blk[7] << 96)
    assert int(got[0, 1]) == wide2 % g


def test_rejection_branch(hostcheck, coracle):
    """Rejection is rare for the default primes (2^-56), so exercise the branch on an admissible
    pseudo-Mersenne prime with the largest possible c: p = 2^33 - c, c just below 2^16 (rejection
    probability ~2^-17 per sample -> a handful of hits in the groups restated below)."""
    from mpyc_amd.finfields import is_prime
    from oracle import pyoracle as po
    c = 65535
    while not is_prime(2**33 - c):
        c -= 2
    p = 2**33 - c
    assert c > 60000
    F = po.Field(p)
    cf = coracle.CField(p)
    n = 1 << 20
    got = hc_coeffs(hostcheck, F, KEY, 123, 8, 2, n)
    want = coracle.rng_coeffs(cf, KEY, 123, 8, 2, n)
    assert (got == want).all()
    assert int(got.max()) < p
    # Independent restatement in Python of the documented layout for this case (S = 8, pack = 2 elements, t = 2:
    # NS = 4 samples per pack, G = 2 packs per 64-byte block, group g serves packs g and g + NG) and of the
    # re-draw rule: a rejected sample takes the first candidate < p of ITS OWN block (counter 2^63 + global sample
    # index), so no two rejected samples can ever receive the same replacement.
    mask, NS, G = 2**33 - 1, 4, 2
    npacks = n // 2
    ng = (npacks + G - 1) // G
    hits = 0
    for g in list(range(0, 3000)) + list(range(ng - 50, ng)):
        blk = coracle.chacha_block(KEY, [g, 0, 123, 0], 8)
        for u in range(G):
            pack = u * ng + g
            for sn in range(NS):
                w = blk[2 * (u * NS + sn):2 * (u * NS + sn) + 2]
                v = (w[0] | (w[1] << 32)) & mask
                if v >= p:
                    hits += 1
                    sidx = g * G * NS + u * NS + sn
                    rb = coracle.chacha_block(KEY, [sidx & 0xffffffff, (sidx >> 32) | 0x80000000, 123, 0], 8)
                    cands = [(rb[2 * i] | (rb[2 * i + 1] << 32)) & mask for i in range(8)]
                    ok = [x for x in cands if x < p]
                    v = ok[0] if ok else cands[-1] - p
                assert int(got[sn // 2, pack * 2 + sn % 2]) == v, (g, u, sn)
    expected = 3050 * G * NS * c / 2**33
    assert abs(hits - expected) < 6 * max(1.0, expected) ** 0.5 + 3
    # the re-draw counter is a function of the sample index alone: distinct samples -> distinct blocks
    assert coracle.chacha_block(KEY, [7, 0x80000000, 123, 0], 8) != coracle.chacha_block(KEY, [8, 0x80000000, 123, 0], 8)
    for q in (2**61 - 1, 2**64 - 189, 2**40 - 87, 2**127 - 1, 2**128 - 173, 2**96 - 17):
        got = hc_coeffs(hostcheck, po.Field(q), KEY, 77, 20, 3, 4099)
        assert (got == coracle.rng_coeffs(coracle.CField(q), KEY, 77, 20, 3, 4099)).all(), q


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
