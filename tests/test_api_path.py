"""The hot path THROUGH the reference's public API, side by side with the reference itself.

tests/api_program.py is an ordinary MPyC program -- `mpc.output(a * b)` on SecFld(2^61-1) arrays, i.e.
Runtime.np_multiply -> _reshare -> output (runtime.py:1096-1141, 603-689, 513-600) -- run twice with the SAME
replayed coefficient draws (secrets.randbelow patched in the reference; the mirror's randbelow hook):
once on the unmodified reference, once with mpyc_amd.install() underneath.  Every party records the SHA-256 of every
share row np_random_split hands out, of every array np_recombine returns and of every opened result; the two runs
must agree digest for digest, for one party (t = 0) and for three local parties over TCP (t = 1).

  * build container (`-m "not gpu"`): the mirror's host logic on tests/cpuctx.py, n = 3000;
  * GPU box (`-m gpu`): the kernels, n = 10^6, against the staged reference copy (_refstage/, see
    tests/test_mpyc_dropin.py); skipped if no copy was staged.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, '_refstage')
PROG = os.path.join(ROOT, 'tests', 'api_program.py')


def run_program(ref, mode, n, parties, tmp, seed=11, reps=2, prime=None, chain=1, timeout=1500, ipc_wire=False):
    env = dict(os.environ)
    env['MPYC_AMD_IPC_WIRE'] = '1' if ipc_wire else '0'
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref])
    env.update(API_MODE=mode, API_N=str(n), API_REPS=str(reps), API_SEED=str(seed), API_CHAIN=str(chain),
               API_DIGEST=os.path.join(tmp, f'dg_{mode}_{parties}'))
    env.pop('MPYC_GPU', None)
    if prime is not None:
        env['API_PRIME'] = str(prime)
    cmd = [sys.executable, PROG, '--no-log'] + ([f'-M{parties}'] if parties > 1 else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, env=env, timeout=timeout)
    assert r.returncode == 0 and 'API_RESULT' in r.stdout, (r.stdout + r.stderr)[-3000:]
    out = []
    for pid in range(parties):
        with open(os.path.join(tmp, f'dg_{mode}_{parties}.{pid}.json')) as fh:
            out.append(json.load(fh))
    return out


def compare(ref_runs, dev_runs, parties, t):
    for pid in range(parties):
        a, b = ref_runs[pid], dev_runs[pid]
        assert a['m'] == b['m'] == parties and a['t'] == b['t'] == t
        tags_a, tags_b = [d[0] for d in a['digests']], [d[0] for d in b['digests']]
        assert tags_a == tags_b, (pid, tags_a, tags_b)
        bad = [(i, x[0]) for i, (x, y) in enumerate(zip(a['digests'], b['digests'])) if x[1] != y[1]]
        assert not bad, f'party {pid}: digests differ from the reference at {bad}'
    # party 0 shares both inputs; with t > 0 every party re-shares in the gate and recombines twice per repetition
    n0 = len(ref_runs[0]['digests'])
    assert n0 >= (2 * parties if t else 0) + 2 * 2, n0
    return n0


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present')
@pytest.mark.parametrize('parties,t', [(1, 0), (3, 1)])
def test_api_path_matches_reference_host_logic(tmp_path, parties, t):
    ref = run_program('/root/reference', 'ref', 3000, parties, str(tmp_path))
    dev = run_program('/root/reference', 'cpuctx', 3000, parties, str(tmp_path))
    compare(ref, dev, parties, t)


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present')
def test_api_path_chain_and_wide_prime_host_logic(tmp_path):
    """a chain of three multiplications (np_recombine -> * -> np_random_split = the fused chain gate in the mirror) over
    the 128-bit prime of configs[3]"""
    ref = run_program('/root/reference', 'ref', 500, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1)
    dev = run_program('/root/reference', 'cpuctx', 500, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1)
    compare(ref, dev, 3, 1)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
@pytest.mark.parametrize('parties,t', [(1, 0), (3, 1)])
def test_api_path_matches_reference_on_gpu_1e6(tmp_path, parties, t):
    """n = 10^6 SecFld(2^61-1) elements through mpc.input / a * b / mpc.output on the GPU box: every share row, every
    recombination and the opened result equal the reference's, which runs beside it on the host cores."""
    n = 1_000_000
    ref = run_program(STAGE, 'ref', n, parties, str(tmp_path), reps=1)
    dev = run_program(STAGE, 'gpu', n, parties, str(tmp_path), reps=1)
    assert dev[0]['gpu_calls'] and dev[0]['gpu_busy_ms'] > 0          # the kernels ran (libffgpu event accounting)
    compare(ref, dev, parties, t)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_api_path_matches_reference_on_gpu_at_the_north_star_size(tmp_path):
    """north_star: "bit-exact vs mpyc.thresha on 10^7-element SecFld arrays" -- through the API, against the reference
    itself run beside it: n = 10^7 over 2^61 - 1, one party (mpc.input / a * b / mpc.output = np_multiply + output,
    runtime.py:1096-1141, 513-600), every digest equal."""
    n = 10_000_000
    ref = run_program(STAGE, 'ref', n, 1, str(tmp_path), reps=1)
    dev = run_program(STAGE, 'gpu', n, 1, str(tmp_path), reps=1)
    assert dev[0]['gpu_calls'] and dev[0]['gpu_busy_ms'] > 0
    compare(ref, dev, 1, 0)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_api_path_three_parties_at_the_north_star_size(tmp_path):
    """The same at n = 10^7 for three parties (t = 1: np_random_split by every party, np_recombine of 2t+1 rows,
    runtime.py:603-689) with the engine's parties exchanging rows over the device-side wire; the reference's three
    parties run the identical program over its TCP mesh beside it."""
    n = 10_000_000
    ref = run_program(STAGE, 'ref', n, 3, str(tmp_path), reps=1, timeout=900)
    dev = run_program(STAGE, 'gpu', n, 3, str(tmp_path), reps=1, ipc_wire=True)
    assert dev[0]['ipc_wire'] is True and dev[0]['gpu_calls']
    compare(ref, dev, 3, 1)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_api_path_chain_wide_prime_on_gpu(tmp_path):
    n = 100_000
    ref = run_program(STAGE, 'ref', n, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1)
    dev = run_program(STAGE, 'gpu', n, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1)
    compare(ref, dev, 3, 1)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_api_path_device_side_wire_on_gpu(tmp_path):
    """Three co-located parties with MPYC_AMD_IPC_WIRE=1: the pickled share rows carry an interprocess handle of the
    device buffer instead of the limb bytes (mpyc_amd/finfields.py, _array_from_ipc); what every party splits, receives,
    recombines and opens is still the reference's, digest for digest -- for 10^6 elements over 2^61 - 1 and for a chain
    of three gates over the two-limb prime 2^128 - 173.  The messages shrink from n x 8 B to a few hundred bytes."""
    n = 1_000_000
    ref = run_program(STAGE, 'ref', n, 3, str(tmp_path), reps=1)
    dev = run_program(STAGE, 'gpu', n, 3, str(tmp_path), reps=2, ipc_wire=True)
    assert dev[0]['ipc_wire'] is True
    compare(ref, run_program(STAGE, 'gpu', n, 3, str(tmp_path), reps=1, ipc_wire=True), 3, 1)
    assert dev[0]['bytes_sent'] < 200_000, dev[0]['bytes_sent']          # inline rows would be >= 4 x 8 MB per repetition
    ref = run_program(STAGE, 'ref', 100_000, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1)
    dev = run_program(STAGE, 'gpu', 100_000, 3, str(tmp_path), prime=2**128 - 173, chain=3, reps=1, ipc_wire=True)
    compare(ref, dev, 3, 1)
    # five parties, t = 2: `output` hands ONE marshalled share to two peers (two acknowledgements per descriptor),
    # and only 2t + 1 = 5 of 5 parties re-share
    ref = run_program(STAGE, 'ref', 100_000, 5, str(tmp_path), reps=1)
    dev = run_program(STAGE, 'gpu', 100_000, 5, str(tmp_path), reps=3, ipc_wire=True)
    assert dev[0]['ipc_wire'] is True and dev[0]['ipc_stats']['inline'] == 0
    compare(ref, run_program(STAGE, 'gpu', 100_000, 5, str(tmp_path), reps=1, ipc_wire=True), 5, 2)
