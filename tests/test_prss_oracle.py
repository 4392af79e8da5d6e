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
