"""PRSS in PRODUCTION mode: the reference's combination (mpyc/thresha.py:163-173, 201-217) over a
counter-mode PRF -- one ChaCha stream per subset key, expanded by the lanes that consume the draws (ffgpu_prss_chacha,
mpyc_amd/csrc/kernels.hpp k_prss_chacha; opt-in: mpyc_amd.thresha.prss_prf = 'chacha' / MPYC_AMD_PRSS_PRF=chacha).

The PRF has no reference counterpart (the reference's is SHAKE128, which stays the default and the parity mode: tests/
test_gpu_api.py::test_prss_matches_reference), so this mode is pinned the way the share-generation CSPRNG is: to RFC 8439's
known-answer vector, to two independent restatements of the public keystream layout (oracle/pyoracle.py in Python integers,
oracle/fforacle.c in C), and to the properties the reference's own tests check for PRSS (tests/test_thresha.py:56-86):
every set of t+1 parties recombines to the same secret, zero sharings recombine to 0 at degree 2t.

  `-m "not gpu"`: the vector, the layout rule in all three places, the two oracles against each other, the properties on
                  the oracle, and the mirror's host logic on tests/cpuctx.py;
  `-m gpu`:       the kernel against the oracles bit for bit (all field policies, both kinds of bound, ragged sizes,
                  several (m, t), accumulate-in-chunks), the properties on the device at 10^6, and that the default mode
                  still gives the reference's golden shares.
"""
import ctypes
import itertools

import numpy as np
import pytest

from fieldutil import unpack
from oracle import pyoracle as po

FIELDS = {                                    # name -> (modulus, binary): one per device policy
    'P61': (2**61 - 1, False), 'P64': (2**64 - 189, False), 'P40': (2**40 - 87, False), 'RC64': (6616326157076047771, False),
    'RC32': (2**31 - 1, False), 'P96': (2**96 - 17, False), 'P80': (2**80 - 65, False), 'P128': (2**128 - 173, False),
    'P127': (2**127 - 1, False), 'MONT128': (258797994007609146293811961253269568351, False),
    'GF2_8': (0x11b, True), 'GF2_5': (0b100101, True), 'GF2_64': ((1 << 64) | 0x1b, True), 'GF2_128': ((1 << 128) | 0x87, True),
    'GF2_16': (0x1002b, True), 'GF2_32': (0x10000008d, True),          # four-byte storage (GF2W32, round 6)
}


def keys_for(m, t, i):
    return {S: bytes([(11 * sum(S) + 3 * len(S) + b) % 251 for b in range(16)])
            for S in itertools.combinations(range(m), m - t) if i in S}


def prf_len(key, bound):
    l = ((bound - 1).bit_length() + 7) // 8
    return l + len(key) if bound & (bound - 1) else l


def test_chacha_block_rfc8439_vector():
    """RFC 8439 section 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00 (the RFC's 96-bit
    nonce occupies state words 13..15; here word 13 is the high half of the 64-bit counter)."""
    blk = po.chacha_block(bytes(range(32)), 1 | (0x09000000 << 32), (0x4a000000).to_bytes(4, 'little') + bytes(4))
    assert blk.hex() == ('10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e'
                         'd2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e')
    from oracle import coracle as co
    w = [1, 0x09000000, 0x4a000000, 0]
    assert b''.join(int(v).to_bytes(4, 'little') for v in co.chacha_block(bytes(range(32)), w)) == blk


def test_layout_rule_in_all_three_places():
    from oracle import coracle as co
    from mpyc_amd import _ffi
    L = _ffi.lib()
    for l in range(1, 65):
        tb, dpt = ctypes.c_int(), ctypes.c_int()
        co.lib().orc_prss_chacha_layout(l, ctypes.byref(tb), ctypes.byref(dpt))
        assert (tb.value, dpt.value) == po.prss_chacha_layout(l), l
        assert L.ffgpu_prss_chacha_layout(l, ctypes.byref(tb), ctypes.byref(dpt)) == 0
        assert (tb.value, dpt.value) == po.prss_chacha_layout(l), l
        lw = (l + 3) // 4
        assert 1 <= tb.value <= 3 and 1 <= dpt.value <= 8 and dpt.value * lw <= 16 * tb.value
    assert po.prss_chacha_layout(24) == (3, 8)        # 64-bit primes: 8 draws of 6 words fill 3 blocks exactly
    assert po.prss_chacha_layout(32) == (1, 2) and po.prss_chacha_layout(1) == (1, 8)


def weights_share(F, m, i, keys):
    return [po.f_S_i(F, m, i, S) for S in keys]


def weights_zero(F, m, i, keys, d):
    i1 = po.reduce(F, i + 1)
    out = []
    for S in keys:
        f = po.f_S_i(F, m, i, S)
        for j in range(d):
            w = f
            for _ in range(j + 1):
                w = po.mul(F, w, i1)
            out.append(w)
    return out


def test_c_oracle_equals_python_oracle():
    from oracle import coracle as co
    for name, (mod, binary) in FIELDS.items():
        F, cf = po.Field(mod, binary), co.CField(mod, binary)
        m, t, i, n = 5, 2, 1, 45
        keys = keys_for(m, t, i)
        bounds = (F.order,) if binary else (F.order, 2, 1 << min(40, mod.bit_length() - 2))
        for bound in bounds:
            l = prf_len(next(iter(keys.values())), bound)
            mb = 0 if bound & (bound - 1) else bound.bit_length() - 1
            k40 = [po.prss_chacha_stream_key(k, b'uci') for k in keys.values()]
            got = unpack(co.prss_chacha(cf, k40, 1, l, mb, 20, weights_share(F, m, i, keys), n), cf.eb)
            assert got == po.np_pseudorandom_share_chacha(F, m, i, keys, bound, b'uci', n), (name, bound)
            got = unpack(co.prss_chacha(cf, k40, t, l, mb, 12, weights_zero(F, m, i, keys, t), n), cf.eb)
            assert got == po.np_pseudorandom_share_0_chacha(F, m, i, keys, bound, b'uci', n, rounds=12), (name, bound)


def test_oracle_shares_lie_on_one_polynomial():
    """tests/test_thresha.py:56-86 on the production PRF: the parties' shares are points of ONE degree-t polynomial
    (any t+1 of them open the same secrets), the zero sharing opens to 0 from 2t+1 points and has degree <= 2t... here
    m = 2t+1, so all of them."""
    for mod, binary in ((2**61 - 1, False), (0x11b, True), (2**128 - 173, False)):
        F = po.Field(mod, binary)
        for m, t in ((3, 1), (5, 2), (7, 3)):
            if m >= F.order:
                continue
            n = 9
            shares, zeros = [], []
            for i in range(m):
                keys = keys_for(m, t, i)
                shares.append(po.np_pseudorandom_share_chacha(F, m, i, keys, F.order, b'pc', n))
                zeros.append(po.np_pseudorandom_share_0_chacha(F, m, i, keys, F.order, b'pc', n))
            secret = po.np_recombine(F, [(i + 1, shares[i]) for i in range(t + 1)])
            for start in range(1, m - t):
                assert po.np_recombine(F, [(i + 1, shares[i]) for i in range(start, start + t + 1)]) == secret
            assert po.np_recombine(F, [(i + 1, zeros[i]) for i in range(2 * t + 1)]) == [0] * n
            assert any(zeros[0])                       # ... and is not the all-zero sharing
            # the secret is the sum of the subsets' draws: sum_S prf_S(h) (f_S(0) = 1)
            allkeys = {S: keys_for(m, t, S[0])[S] for S in itertools.combinations(range(m), m - t)}
            want = [0] * n
            for S, key in allkeys.items():
                for h, v in enumerate(po.prf_values_chacha(key, F.order, b'pc', n)):
                    want[h] = po.add(F, want[h], po.reduce(F, v))
            assert secret == want


def test_draws_do_not_depend_on_the_field():
    """runtime.py:758-761 evaluates ONE set of PRFs (one bound) over two fields with the same common input: the integers
    drawn must be the same -- the stream key and the layout depend on (key, input, bound) only."""
    key = bytes(range(16))
    a = po.prf_values_chacha(key, 1 << 20, b'conv', 33)
    F1, F2 = po.Field(2**61 - 1), po.Field(2**128 - 173)
    k = {(0,): key}
    assert po.np_pseudorandom_share_chacha(F1, 1, 0, k, 1 << 20, b'conv', 33) == a
    assert po.np_pseudorandom_share_chacha(F2, 1, 0, k, 1 << 20, b'conv', 33) == a


def test_mirror_host_logic_on_cpu_context(monkeypatch):
    """mpyc_amd.thresha._prss_device in production mode on tests/cpuctx.py: stream keys, chunking (more subset keys than one
    launch takes), weights of both conventions; the default mode is untouched."""
    from cpuctx import use_cpu_contexts
    import mpyc_amd.finfields as gff
    import mpyc_amd.thresha as gth
    use_cpu_contexts(monkeypatch)
    monkeypatch.setattr(gff, '_ctx_cache', {})
    gff._pGF.cache_clear()
    F, OF = gff.GF(2**61 - 1), po.Field(2**61 - 1)
    n = 7
    for m, t, i in ((3, 1, 2), (9, 4, 0)):                     # C(8, 4) = 70 subset keys for party 0: three launches
        keys = keys_for(m, t, i)
        prfs = {S: gth.PRF(k, F.order) for S, k in keys.items()}
        ref_mode = [int(v) for v in np.asarray(gth.np_pseudorandom_share(F, m, i, prfs, b'u', n).value)]
        assert ref_mode == po.np_pseudorandom_share(OF, m, i, keys, F.order, b'u', n)
        monkeypatch.setattr(gth, 'prss_prf', 'chacha')
        got = [int(v) for v in np.asarray(gth.np_pseudorandom_share(F, m, i, prfs, b'u', n).value)]
        assert got == po.np_pseudorandom_share_chacha(OF, m, i, keys, F.order, b'u', n) and got != ref_mode
        got0 = [int(v) for v in np.asarray(gth.np_pseudorandom_share_0(F, m, i, prfs, b'u', n).value)]
        assert got0 == po.np_pseudorandom_share_0_chacha(OF, m, i, keys, F.order, b'u', n)
        lst = [int(v.value) for v in gth.pseudorandom_share_zero(F, m, i, prfs, b'u', n)]
        assert lst == po.np_pseudorandom_share_0_chacha(OF, m, i, keys, F.order, b'u', n, list_convention=True)
        monkeypatch.setattr(gth, 'prss_prf', 'shake')
    assert gth.prss_chacha_stream_key(b'k' * 16, b's') == po.prss_chacha_stream_key(b'k' * 16, b's')
    monkeypatch.setattr(gth, 'prss_prf', 'keccak')
    with pytest.raises(ValueError):
        gth.np_pseudorandom_share(F, 3, 0, {(0, 1): gth.PRF(b'k' * 16, F.order)}, b'u', n)
    # the mode is validated on every call: rounds outside {20, 12, 8}; ChaCha8 only on request; a PRF object that is not
    # the SHAKE128 one is neither replaced nor silently expanded by SHAKE; draws wider than the kernel's 64 bytes
    monkeypatch.setattr(gth, 'prss_prf', 'chacha')
    one = {(0, 1): gth.PRF(b'k' * 16, F.order)}
    for bad in (7, 10, 0):
        monkeypatch.setattr(gth, 'prss_rounds', bad)
        with pytest.raises(ValueError):
            gth.np_pseudorandom_share(F, 3, 0, one, b'u', n)
    monkeypatch.setattr(gth, 'prss_rounds', 8)
    monkeypatch.setattr(gth, 'prss_allow_chacha8', False)
    with pytest.raises(ValueError, match='ChaCha8'):
        gth.np_pseudorandom_share(F, 3, 0, one, b'u', n)
    monkeypatch.setattr(gth, 'prss_rounds', 20)
    assert gth.prss_mode_tag() == 'chacha20/v1'

    class OtherPRF(gth.PRF):
        def __call__(self, s, n=None):
            return [0] * (n or 1)
    with pytest.raises(TypeError, match='SHAKE128 PRF objects only'):
        gth.np_pseudorandom_share(F, 3, 0, {(0, 1): OtherPRF(b'k' * 16, F.order)}, b'u', n)
    with pytest.raises(NotImplementedError, match='64'):
        gth.np_pseudorandom_share(F, 3, 0, {(0, 1): gth.PRF(b'k' * 60, F.order)}, b'u', n)
    monkeypatch.setattr(gth, 'prss_prf', 'shake')
    assert gth.prss_mode_tag() == 'shake'


# ------------------------------------------------------------------------------------------------------------ GPU
def gpu_field(api, mod, binary):
    finfields, gfpx, _ = api
    return finfields.GF(gfpx.BinaryPolynomial(mod)) if binary else finfields.GF(mod)


@pytest.fixture(scope='module')
def api():
    import torch
    assert torch.cuda.is_available()
    from mpyc_amd import finfields, gfpx, thresha
    return finfields, gfpx, thresha


def dev_ints(a):
    return [int(v) for v in np.asarray(a.value).reshape(-1)]


@pytest.mark.gpu
def test_device_equals_c_oracle_all_policies(api, monkeypatch):
    """every field policy, bound = order and two power-of-two bounds, share and zero sharing, ragged n (tiles of 8, 5, 3,
    2 draws cut by the end of the array), ChaCha20 / 12 / 8"""
    from oracle import coracle as co
    finfields, gfpx, thresha = api
    monkeypatch.setattr(thresha, 'prss_prf', 'chacha')
    monkeypatch.setattr(thresha, 'prss_allow_chacha8', True)
    for name, (mod, binary) in FIELDS.items():
        F, OF, cf = gpu_field(api, mod, binary), po.Field(mod, binary), co.CField(mod, binary)
        bounds = (OF.order,) if binary else (OF.order, 2, 1 << min(40, mod.bit_length() - 2))
        for (m, t, i), n, rounds in (((3, 1, 0), 4099, 20), ((5, 2, 4), 1001, 12), ((7, 3, 2), 517, 8)):
            if m >= OF.order:
                continue
            keys = keys_for(m, t, i)
            monkeypatch.setattr(thresha, 'prss_rounds', rounds)
            for bound in bounds:
                prfs = {S: thresha.PRF(k, bound) for S, k in keys.items()}
                l = prf_len(next(iter(keys.values())), bound)
                mb = 0 if bound & (bound - 1) else bound.bit_length() - 1
                k40 = [po.prss_chacha_stream_key(k, b'pc7') for k in keys.values()]
                want = unpack(co.prss_chacha(cf, k40, 1, l, mb, rounds, weights_share(OF, m, i, keys), n), cf.eb)
                got = thresha.np_pseudorandom_share(F, m, i, prfs, b'pc7', n)
                assert isinstance(got, F.array) and dev_ints(got) == want, (name, m, bound)
                want0 = unpack(co.prss_chacha(cf, k40, t, l, mb, rounds, weights_zero(OF, m, i, keys, t), n), cf.eb)
                assert dev_ints(thresha.np_pseudorandom_share_0(F, m, i, prfs, b'pc7', n)) == want0, (name, m, bound, 'zero')
    # n = 0 and n = 1
    monkeypatch.setattr(thresha, 'prss_rounds', 20)
    F = gpu_field(api, 2**61 - 1, False)
    prfs = {S: thresha.PRF(k, F.order) for S, k in keys_for(3, 1, 0).items()}
    assert dev_ints(thresha.np_pseudorandom_share(F, 3, 0, prfs, b'x', 0)) == []
    assert dev_ints(thresha.np_pseudorandom_share(F, 3, 0, prfs, b'x', 1)) == \
        po.np_pseudorandom_share_chacha(po.Field(2**61 - 1), 3, 0, keys_for(3, 1, 0), F.order, b'x', 1)


@pytest.mark.gpu
def test_device_equals_python_oracle_three_limb_primes(api, monkeypatch):
    """129..192-bit primes (24-byte elements; the C oracle stops at 128 bits): l = 33..40 bytes per draw, tiles of
    3 blocks / 5 draws and 2 blocks / 3 draws"""
    finfields, gfpx, thresha = api
    monkeypatch.setattr(thresha, 'prss_prf', 'chacha')
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'wide.json')) as fh:
        generic = [int(c['modulus'], 16) for c in json.load(fh).values()]
    generic = [q for q in generic if (1 << q.bit_length()) - q >= 1 << 31][:1]      # not 2^k - c: the Montgomery policy
    assert generic
    for p in [finfields.find_prime_root(bits)[0] for bits in (129, 136, 160, 192)] + generic:
        bits = p.bit_length()
        F, OF = finfields.GF(p), po.Field(p)
        for (m, t, i), n in (((3, 1, 1), 203), ((5, 2, 0), 64)):
            keys = keys_for(m, t, i)
            for bound in (p, 1 << 100):
                prfs = {S: thresha.PRF(k, bound) for S, k in keys.items()}
                assert dev_ints(thresha.np_pseudorandom_share(F, m, i, prfs, b'w', n)) == \
                    po.np_pseudorandom_share_chacha(OF, m, i, keys, bound, b'w', n), (bits, m, bound)
                assert dev_ints(thresha.np_pseudorandom_share_0(F, m, i, prfs, b'w', n)) == \
                    po.np_pseudorandom_share_0_chacha(OF, m, i, keys, bound, b'w', n), (bits, m, bound)


@pytest.mark.gpu
def test_parties_shares_recombine_on_device_1e6(api, monkeypatch):
    """tests/test_thresha.py:56-86 at n = 10^6 on the kernels: m = 3 and m = 7 parties' production-mode shares recombine to
    ONE secret from different sets of t+1 parties; zero sharings recombine to 0 from 2t+1 parties and are not zero; the
    same call twice gives the same shares, another common input gives others."""
    finfields, gfpx, thresha = api
    monkeypatch.setattr(thresha, 'prss_prf', 'chacha')
    n = 1_000_003
    for mod, binary in ((2**61 - 1, False), (2**64 - 189, False), ((1 << 64) | 0x1b, True)):
        F = gpu_field(api, mod, binary)
        for m, t in ((3, 1), (7, 3)):
            shares, zeros = [], []
            for i in range(m):
                prfs = {S: thresha.PRF(k, F.order) for S, k in keys_for(m, t, i).items()}
                shares.append(thresha.np_pseudorandom_share(F, m, i, prfs, b'pc', n))
                zeros.append(thresha.np_pseudorandom_share_0(F, m, i, prfs, b'pc', n))
            a = thresha.np_recombine(F, [(i + 1, shares[i]) for i in range(t + 1)])
            b = thresha.np_recombine(F, [(i + 1, shares[i]) for i in range(m - t - 1, m)])
            assert bool((a == b).all())
            z = thresha.np_recombine(F, [(i + 1, zeros[i]) for i in range(2 * t + 1)])
            assert not bool((z != 0).any()) and bool((zeros[0] != 0).any())
            prfs = {S: thresha.PRF(k, F.order) for S, k in keys_for(m, t, 0).items()}
            again = thresha.np_pseudorandom_share(F, m, 0, prfs, b'pc', n)
            other = thresha.np_pseudorandom_share(F, m, 0, prfs, b'pd', n)
            assert bool((again == shares[0]).all()) and bool((other != shares[0]).any())


@pytest.mark.gpu
def test_default_mode_is_still_the_reference_prf(api):
    """parity mode byte-identical to the reference: with prss_prf untouched ('shake') the golden shares come out (the full
    check is tests/test_gpu_api.py::test_prss_matches_reference); switching the mode changes the values"""
    finfields, gfpx, thresha = api
    assert thresha.prss_prf == 'shake'
    F, OF = finfields.GF(2**61 - 1), po.Field(2**61 - 1)
    keys = keys_for(3, 1, 1)
    prfs = {S: thresha.PRF(k, F.order) for S, k in keys.items()}
    assert dev_ints(thresha.np_pseudorandom_share(F, 3, 1, prfs, b'u', 100)) == \
        po.np_pseudorandom_share(OF, 3, 1, keys, F.order, b'u', 100)


@pytest.mark.gpu
def test_every_draw_width_and_mask(api):
    """The kernel takes the draw width l (1..64 bytes) and the mask width as run-time values: every l -- i.e. every tile shape
    (1-3 blocks, 1-8 draws), every partial last word, every number of limbs in the wide reduction -- and a sweep of mask
    widths, for one-, two- and three-limb primes and 32-bit storage, against a restatement from RFC 8439 blocks and the
    documented layout in Python integers (ks = 2 streams, d = 2 draws per element, ragged n)."""
    finfields, gfpx, thresha = api
    from mpyc_amd.finfields import _context

    def expected(F, k40s, d, l, mask_bits, weights, n, rounds):
        tb, dpt = po.prss_chacha_layout(l)
        lw = (l + 3) // 4
        out = []
        for h in range(n):
            tile, slot = divmod(h, dpt)
            acc = 0
            for s, k40 in enumerate(k40s):
                for j in range(d):
                    ks = b''.join(po.chacha_block(k40[:32], (tile * d + j) * tb + b, k40[32:], rounds) for b in range(tb))
                    v = int.from_bytes(ks[4 * slot * lw:4 * slot * lw + l], 'little')
                    v = v & ((1 << mask_bits) - 1) if mask_bits else v % F.order
                    acc = po.add(F, acc, po.mul(F, po.reduce(F, v), weights[s * d + j]))
            out.append(acc)
        return out

    k40s = [bytes((7 * i + 3 * s) % 256 for i in range(40)) for s in range(2)]
    for mod in (2**61 - 1, 2**128 - 173, 2**31 - 1, finfields.find_prime_root(160)[0], 2**96 - 17):
        F, OF = finfields.GF(mod), po.Field(mod)
        ctx = _context(F)
        weights = [(mod // 3 + 11 * i) % mod for i in range(4)]
        for l in range(1, 65):
            tb, dpt = po.prss_chacha_layout(l)
            n = 2 * dpt + (l % 3) + 1
            got = ctx.prss_chacha(k40s, 2, l, weights, n, mask_bits=0, rounds=12).to_ints()
            assert got == expected(OF, k40s, 2, l, 0, weights, n, 12), (hex(mod), l)
        bits = mod.bit_length()
        for mb in sorted({1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 65, bits - 2, bits - 1} & set(range(1, bits))):
            l = (mb + 7) // 8
            n = 19
            got = ctx.prss_chacha(k40s, 2, l, weights, n, mask_bits=mb, rounds=8).to_ints()
            assert got == expected(OF, k40s, 2, l, mb, weights, n, 8), (hex(mod), 'mask', mb)
