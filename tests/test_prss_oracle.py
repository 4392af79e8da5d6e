"""PRSS (thresha.py:135-266): the oracle restatement against golden vectors from the reference."""
import json
import os

from oracle import pyoracle as po
from fieldutil import unhex

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def load():
    out = {}
    for fname in ('prss.json', 'prss_wide.json'):             # prss_wide.json: three-limb prime fields
        with open(os.path.join(GOLDEN, fname)) as fh:
            out.update(json.load(fh))
    return out


def settings(case):
    F = po.Field(int(case['modulus'], 16), case['binary'])
    uci, n = bytes.fromhex(case['uci']), case['n']
    for st in case['settings']:
        keys = {tuple(int(x) for x in k.split(',')): bytes.fromhex(v) for k, v in st['keys'].items()}
        yield F, uci, n, st, keys


def test_prf_and_shares_match_reference():
    for name, case in load().items():
        for F, uci, n, st, keys in settings(case):
            m, bound = st['m'], int(st['bound'], 16)
            first = next(iter(keys))
            assert po.prf_values(keys[first], bound, uci, n) == unhex(st['prf0']), name
            for i, party in enumerate(st['parties']):
                mine = {S: k for S, k in keys.items() if i in S}
                assert po.np_pseudorandom_share(F, m, i, mine, bound, uci, n) == unhex(party['share']), (name, m, i)
                if 'zero_np' in party:
                    assert po.np_pseudorandom_share_0(F, m, i, mine, bound, uci, n) == unhex(party['zero_np'])
                    assert po.np_pseudorandom_share_0(F, m, i, mine, bound, uci, n, True) == unhex(party['zero_list'])
            # all parties' shares lie on one degree-t polynomial through the recorded secret
            t = st['t']
            for start in range(0, m - t):
                pts = [(i + 1, unhex(st['parties'][i]['share'])) for i in range(start, start + t + 1)]
                assert po.np_recombine(F, pts) == unhex(st['secret']), (name, m, start)
            if 'zero_np' in st['parties'][0]:
                pts = [(i + 1, unhex(st['parties'][i]['zero_np'])) for i in range(min(m, 2 * t + 1))]
                assert po.np_recombine(F, pts) == [0] * n


def test_prf_reference_kats():
    """tests/test_thresha.py:42-54: bound 1 -> 0; determinism; range."""
    key = int('0x00112233445566778899aabbccddeeff', 16).to_bytes(16, byteorder='little')
    assert po.prf_values(key, 1, b'test', 1) == [0]
    y = po.prf_values(key, 100, b'', 1)
    assert 0 <= y[0] < 100 and y == po.prf_values(key, 100, b'', 1)


def test_mirror_prss_takes_bytearray_keys_and_foreign_prf_classes(monkeypatch):
    """Host logic of mpyc_amd.thresha._prss_device on the CPU context: the runtime keeps the PRSS keys it RECEIVED as
    bytearray slices (runtime.py:139) and builds its PRFs with mpyc.thresha.PRF, a different class with the same
    algorithm.  Both must give the shares of the golden vectors through the engine's expansion path (whose C entry point
    takes bytes only -- the CPU context asserts that), and a PRF class that is NOT registered is called as it is."""
    from cpuctx import use_cpu_contexts
    import mpyc_amd.finfields as gff
    import mpyc_amd.thresha as gth
    use_cpu_contexts(monkeypatch)
    monkeypatch.setattr(gff, '_ctx_cache', {})
    gff._pGF.cache_clear()

    class RefPRF:                                 # stands for mpyc.thresha.PRF: same fields, same draws
        def __init__(self, key, bound):
            self.key, self.max = key, bound
            self.byte_length = ((bound - 1).bit_length() + 7) // 8 + (len(key) if bound & (bound - 1) else 0)

    class OtherPRF(RefPRF):                       # not registered: must be CALLED (here: it counts the calls)
        calls = 0

        def raw(self, s, n):
            OtherPRF.calls += 1
            import hashlib
            return hashlib.shake_128(bytes(self.key) + s).digest(n * self.byte_length)
    monkeypatch.setattr(gth, 'SHAKE_PRF_TYPES', {gth.PRF, RefPRF})
    try:
        case = load()['P61']
        F = gff.GF(int(case['modulus'], 16))
        uci, n = bytes.fromhex(case['uci']), case['n']
        for st in case['settings']:
            m, bound = st['m'], int(st['bound'], 16)
            keys = {tuple(int(x) for x in k.split(',')): bytes.fromhex(v) for k, v in st['keys'].items()}
            for i, party in enumerate(st['parties']):
                for cls in (gth.PRF, RefPRF, OtherPRF):
                    prfs = {S: cls(bytearray(k), bound) for S, k in keys.items() if i in S}
                    got = gth.np_pseudorandom_share(F, m, i, prfs, bytearray(uci), n)
                    assert [int(v) for v in got.value] == unhex(party['share']), (m, i, cls.__name__)
        assert OtherPRF.calls > 0
    finally:
        gff._pGF.cache_clear()


def test_foreign_prf_class_is_admitted_only_after_a_known_answer_check(monkeypatch):
    """install() hands mpyc.thresha.PRF to register_shake_prf: a class whose draws are shake_128(key + s) chopped by the
    reference's byte_length rule is admitted to the engine's expansion; one with another XOF, another length rule or
    another constructor is not (it keeps being called as the object it is)."""
    import hashlib
    import mpyc_amd.thresha as gth
    monkeypatch.setattr(gth, 'SHAKE_PRF_TYPES', {gth.PRF})

    class Same:
        def __init__(self, key, bound):
            self.key, self.max = key, bound
            self.byte_length = ((bound - 1).bit_length() + 7) // 8 + (len(key) if bound & (bound - 1) else 0)

        def __call__(self, s, n=None):
            n_ = 1 if n is None else n
            dk = hashlib.shake_128(self.key + s).digest(n_ * self.byte_length)
            x = [int.from_bytes(dk[i:i + self.byte_length], 'little') % self.max for i in range(0, len(dk), self.byte_length)]
            return x[0] if n is None else x

    class OtherXof(Same):
        def __call__(self, s, n=None):
            n_ = 1 if n is None else n
            dk = hashlib.shake_256(self.key + s).digest(n_ * self.byte_length)
            x = [int.from_bytes(dk[i:i + self.byte_length], 'little') % self.max for i in range(0, len(dk), self.byte_length)]
            return x[0] if n is None else x

    class OtherLength(Same):
        def __init__(self, key, bound):
            super().__init__(key, bound)
            self.byte_length += 1

    class OtherCtor:
        def __init__(self, key):
            self.key = key

    assert gth.register_shake_prf(Same) and Same in gth.SHAKE_PRF_TYPES
    for cls in (OtherXof, OtherLength, OtherCtor):
        assert gth.register_shake_prf(cls) is False and cls not in gth.SHAKE_PRF_TYPES
    import os
    import sys
    for root in (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '_refstage'), '/root/reference'):
        if os.path.isdir(os.path.join(root, 'mpyc')):
            monkeypatch.syspath_prepend(root)
            sys.modules.pop('mpyc.thresha', None) if 'mpyc' not in sys.modules else None
            from mpyc import thresha as ref
            assert gth.register_shake_prf(ref.PRF)            # the reference's own class passes
            break
