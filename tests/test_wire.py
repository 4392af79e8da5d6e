"""Wire/marshal format against bytes produced by the REFERENCE:
tests/golden/wire.json holds field.to_bytes outputs (finfields.py:91-102) and frames written by
asyncoro.MessageExchanger.send (asyncoro.py:54-64), see tests/golden/make_golden.py::wire_cases."""
import json
import os
import pickle
import subprocess
import sys

import pytest

from fieldutil import unhex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def wire_golden():
    with open(os.path.join(GOLDEN, 'wire.json')) as fh:
        return json.load(fh)


def _field(case):
    from mpyc_amd import finfields as gff, gfpx
    mod = int(case['modulus'], 16)
    return gff.GF(gfpx.GFpX(2)(mod)) if case['binary'] else gff.GF(mod)


def test_frames_match_reference_send(wire_golden):
    from mpyc_amd import wire
    stream = b''
    for fr in wire_golden['frames']:
        got = wire.frame(fr['pc'], bytes.fromhex(fr['payload']))
        assert got.hex() == fr['frame'], fr['pc']
        stream += got
    want = [(fr['pc'], bytes.fromhex(fr['payload'])) for fr in wire_golden['frames']]
    assert list(wire.unframe(stream)) == want
    # reassembly from odd-sized chunks, as the reference's data_received does it (asyncoro.py:66-106)
    rd, got, pos, step = wire.FrameReader(), [], 0, 1
    while pos < len(stream):
        got += rd.feed(stream[pos:pos + step])
        pos += step
        step += 2
    assert got == want and rd.pending == 0
    with pytest.raises(ValueError):
        list(wire.unframe(stream[:-1]))


def test_host_to_bytes_matches_reference(wire_golden):
    """the mirror's element-layer functions on host integers (scalars, list path)"""
    for name, case in wire_golden['fields'].items():
        F = _field(case)
        vals = unhex(case['values'])
        assert F.byte_length == case['byte_length'], name
        assert F.to_bytes(vals).hex() == case['bytes'], name
        assert F.from_bytes(bytes.fromhex(case['bytes'])) == vals, name


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present (build container only)')
def test_committed_fixtures_reproduce_from_committed_generator(tmp_path):
    """`committed script => committed fixtures`: re-run tests/golden/make_golden.py against the reference into a
    scratch directory and compare every file byte for byte."""
    env = dict(os.environ, PYTHONPATH='/root/reference', GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden.py'), '--no-log'], capture_output=True, text=True,
                       cwd='/tmp', env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith('.json'))
    assert sorted(os.listdir(tmp_path)) == names
    for f in names:
        with open(os.path.join(GOLDEN, f), 'rb') as a, open(os.path.join(tmp_path, f), 'rb') as b:
            assert a.read() == b.read(), f'{f} differs from what the committed generator produces'


@pytest.mark.gpu
def test_device_rows_marshal_to_reference_bytes(wire_golden):
    from mpyc_amd import wire
    for name, case in wire_golden['fields'].items():
        F = _field(case)
        vals = unhex(case['values'])
        want = bytes.fromhex(case['bytes'])
        A = F.array(vals)
        assert A.to_wire() == want, name                                   # device limbs -> field.to_bytes format
        B = F.array.from_wire(want)
        assert [int(v) for v in B._dev.to_ints()] == vals, name            # field.from_bytes format -> device limbs
        assert wire.unmarshal(F, wire.marshal(A), shape=(len(vals),)).to_wire() == want
        # 2-D arrays travel row-major; pickle (runtime.py:484,571,655) carries the same bytes
        A2 = F.array([vals[:8], vals[8:16]])
        assert A2.to_wire() == want[:16 * F.byte_length]
        blob = pickle.dumps(A2)
        assert want[:16 * F.byte_length] in blob
        C = pickle.loads(blob)
        assert type(C) is type(A2) and C.shape == (2, 8) and C.to_wire() == A2.to_wire()
        assert pickle.loads(pickle.dumps(A2.value)).to_wire() == A2.to_wire()          # `.value` is what the runtime pickles
        # frames of device rows
        fr = wire.frame_rows(-7, [A, B])
        (pc, payload), = wire.unframe(fr)
        assert pc == -7 and payload == want + want
        # non-canonical / malformed peer input
        if F.byte_length > 1:
            with pytest.raises(ValueError):
                F.array.from_wire(want[:-1])
        if not case['binary'] and F.byte_length * 8 > (F.order - 1).bit_length():
            q = F.order
            raw = b''.join(int(v).to_bytes(F.byte_length, 'little') for v in (q, q + 1))
            assert [int(v) for v in F.array.from_wire(raw)._dev.to_ints()] == [0, 1], name     # reduced, like field.array()


def test_find_prime_root_matches_reference():
    """finfields.find_prime_root incl. n-th roots of unity for n > 2 (finfields.py:311-344) against the reference's
    outputs (tests/golden/roots.json)."""
    from mpyc_amd import finfields as gff
    with open(os.path.join(GOLDEN, 'roots.json')) as fh:
        cases = json.load(fh)
    for c in cases:
        p, n, w = gff.find_prime_root(c['l'], c['blum'], c['n'])
        assert (p, n, w) == (int(c['p'], 16), c['n_out'], int(c['w'], 16)), c
        if c['n'] > 2:
            assert pow(w, n, p) == 1 and w != 1 and p % 4 == 3 and (p - 1) % (2 * n) == 0
