"""BASELINE.json configs[0] at its stated size: thresha.random_split + recombine on the LIST path,
m = 3, t = 1, GF(2^61 - 1), n = 10^4 secrets (mpyc/thresha.py:23-44, 88-116).

tests/list_path_program.py runs the mirror under install() (device path: n >= mpyc_amd.list_path_min) beside the
reference's own functions on the same replayed draws and compares, element by element: the share matrix (raw-int and
field-element secrets; also against oracle/pyoracle.random_split), and recombination from every pair of points, from all
three and in a rotated x order, at x_r = 0 and x_rs = [0, 5] -- for raw-int shares against the reference's UN-REDUCED sums
(thresha.py:109; mirror == reference % p) and against oracle.recombine_unreduced, for field-element shares by value and type.

  * build container (`-m "not gpu"`): the mirror's host logic on tests/cpuctx.py against /root/reference;
  * GPU box (`-m gpu`): the kernels, against the staged reference copy (_refstage/).
Both must produce the same digests (same seed): the CPU twin's digests are pinned below.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, '_refstage')
PROG = os.path.join(ROOT, 'tests', 'list_path_program.py')

# SHA-256 over the (m, n) share matrix / the opened secrets of the seeded run (LP_SEED=5, n=10^4, m=3, t=1, 2^61-1), as the
# REFERENCE produced them in the build container; the GPU run must reproduce them
SPLIT_DIGEST = '7cd57acb62396d476da028bc0ab94c6fabe010eb487fbaeacdf73ccb5e94c987'
OPENED_DIGEST = '527fefd7b0d5ef8072cb0610a4d36f45decb486fca60f935d54450c5ff4a17a6'


def run(ref, mode, n=10_000, m=3, t=1, prime=2**61 - 1, timeout=900):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref])
    env.update(LP_MODE=mode, LP_N=str(n), LP_M=str(m), LP_T=str(t), LP_PRIME=str(prime), LP_SEED='5')
    env.pop('MPYC_AMD_CPUCTX', None)
    r = subprocess.run([sys.executable, PROG], capture_output=True, text=True, cwd='/tmp', env=env, timeout=timeout)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith('LIST_PATH_RESULT ')), None)
    assert r.returncode == 0 and line is not None, (r.stdout + r.stderr)[-3000:]
    return json.loads(line[len('LIST_PATH_RESULT '):])


def check(res, n):
    assert res['n'] == n and res['m'] == 3 and res['t'] == 1 and res['prime_bits'] == 61
    assert res['list_path_min'] <= n                  # the device path was taken (the program refuses otherwise)
    assert res['subsets_checked'] == 5                # (1,2) (1,3) (2,3) (1,2,3) (3,2,1)
    if n == 10_000 and SPLIT_DIGEST is not None:
        assert res['split_digest'] == SPLIT_DIGEST and res['opened_digest'] == OPENED_DIGEST


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present')
def test_configs0_list_path_1e4_host_logic():
    check(run('/root/reference', 'cpuctx'), 10_000)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_configs0_list_path_1e4_on_gpu():
    res = run(STAGE, 'gpu')
    check(res, 10_000)
    assert res['mode'] == 'gpu'


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_list_path_other_settings_on_gpu():
    """the same comparison at m = 7, t = 3 over 2^64 - 189 (t draws per secret: the c[0] -> X^t convention matters) and
    over the two-limb prime 2^128 - 173"""
    for prime, m, t in ((2**64 - 189, 7, 3), (2**128 - 173, 5, 2)):
        res = run(STAGE, 'gpu', n=3000, m=m, t=t, prime=prime)
        assert res['n'] == 3000 and res['m'] == m and res['t'] == t and res['prime_bits'] == prime.bit_length()
