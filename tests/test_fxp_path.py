"""Digest-level parity of the FIXED-POINT product through the reference's API.

tests/fxp_program.py -- `mpc.output(a * b)` on SecFxp(32) arrays: Runtime.np_multiply -> np_trunc (runtime.py:1096-1141,
838-873) -> np_random_bits / _np_randoms (runtime.py:4187-4273, 4062-4103; PRSS, thresha.py:163-217) -> output -- runs on
the unmodified reference and under mpyc_amd.install() with the SAME PRSS keys (secrets.token_bytes patched before
mpyc.runtime is imported) and the same replayed np_random_split coefficients.  Party by party the two runs must agree on the
SHA-256 of every opened array (the squares opened inside np_random_bits, the MASKED value `c` of np_trunc, the final
output), of the party's share `y` of the truncated product, and of the float64 result.

This replaces the instrumented replay of round 3 (tools/fxp_api_probe.py, removed): the reference's own behaviour for ARRAY
types -- `issubclass(sftype, SecureFixedPoint)` at runtime.py:852 is false for array types, so the mask of np_trunc is f bits
short and about one element in 10^6 comes out off by 2^(l-f) = 2^48 -- is reproduced bit for bit, outliers included: a run
whose seeded randomness produces such an element (FXP_SEED chosen for it, see OUTLIER_SEED) has the same digests in both.

  `-m "not gpu"`: host logic of the mirror (HostView integer algebra, PRSS combination, lazy readers) on tests/cpuctx.py,
                  n = 300, one party and -M3;
  `-m gpu`:       the kernels, n = 10^5, one party and -M3, against the staged reference copy.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, '_refstage')
PROG = os.path.join(ROOT, 'tests', 'fxp_program.py')
OUTLIER_SEED = 4      # FXP_SEED for which the reference itself (one party, n = 10^5) produces ONE short-mask outlier (found by
#                       running the reference over seeds 1..10 in the build container: seeds 4 and 9 have one, the others none)


def run_program(ref, mode, n, parties, tmp, seed=3, timeout=1500, prf=None):
    env = dict(os.environ)
    env['MPYC_AMD_IPC_WIRE'] = '0'
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref])
    env.update(FXP_MODE=mode, FXP_N=str(n), FXP_SEED=str(seed), FXP_DIGEST=os.path.join(tmp, f'fx_{mode}_{parties}_{prf}'))
    for k_ in ('MPYC_AMD_CPUCTX', 'MPYC_AMD_PRSS_PRF', 'FXP_REPS'):
        env.pop(k_, None)
    if prf:
        env['MPYC_AMD_PRSS_PRF'] = prf
    cmd = [sys.executable, PROG, '--no-log'] + ([f'-M{parties}'] if parties > 1 else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, env=env, timeout=timeout)
    assert r.returncode == 0 and 'FXP_RESULT' in r.stdout, (r.stdout + r.stderr)[-3000:]
    out = []
    for pid in range(parties):
        with open(os.path.join(tmp, f'fx_{mode}_{parties}_{prf}.{pid}.json')) as fh:
            out.append(json.load(fh))
    return out


def compare(ref_runs, dev_runs, parties):
    for pid in range(parties):
        a, b = ref_runs[pid], dev_runs[pid]
        assert a['m'] == b['m'] == parties and a['field_bits'] == b['field_bits'] == 80
        da, db = dict(a['digests']), dict(b['digests'])
        assert list(da) == list(db), (pid, list(da), list(db))
        opened = [k_ for k_ in da if k_.startswith('opened')]
        assert len(opened) >= 3                          # squares of np_random_bits, masked value of np_trunc, final output
        # concurrent coroutines (the re-sharing of a * b and of r * r + z) may finish in either order: the openings are
        # compared as a multiset, the last two (np_trunc's masked value `c`, then the final output) in order
        assert sorted(da[k_] for k_ in opened) == sorted(db[k_] for k_ in opened), f'party {pid}: opened arrays differ'
        for k_ in opened[-2:] + ['y', 'out']:
            assert da[k_] == db[k_], f'party {pid}: {k_} differs from the reference'
        assert a['outliers_reference_trunc_mask'] == b['outliers_reference_trunc_mask']
        assert a['max_abs_error'] == b['max_abs_error']


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present')
@pytest.mark.parametrize('parties', [1, 3])
def test_fxp_product_matches_reference_host_logic(tmp_path, parties):
    ref = run_program('/root/reference', 'ref', 300, parties, str(tmp_path))
    dev = run_program('/root/reference', 'cpuctx', 300, parties, str(tmp_path))
    compare(ref, dev, parties)
    assert ref[0]['max_abs_error'] < 0.01


@pytest.mark.skipif(not os.path.isdir('/root/reference/mpyc'), reason='reference checkout not present')
def test_fxp_product_production_prf_host_logic(tmp_path):
    """MPYC_AMD_PRSS_PRF=chacha (every party): other random bits, so other roundings -- the result is still the product to
    fixed-point accuracy, and differs from the parity-mode run only in the probabilistic rounding (one unit of 2^-16)"""
    import numpy as np                                # noqa: F401
    shake = run_program('/root/reference', 'cpuctx', 200, 3, str(tmp_path))
    chacha = run_program('/root/reference', 'cpuctx', 200, 3, str(tmp_path), prf='chacha')
    assert chacha[0]['prss_prf'] == 'chacha' and chacha[0]['max_abs_error'] < 0.01
    assert dict(chacha[0]['digests'])['y'] != dict(shake[0]['digests'])['y']


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
@pytest.mark.parametrize('parties', [1, 3])
def test_fxp_product_matches_reference_on_gpu_1e5(tmp_path, parties):
    n = 100_000
    seed = OUTLIER_SEED if parties == 1 else 3
    ref = run_program(STAGE, 'ref', n, parties, str(tmp_path), seed=seed)
    dev = run_program(STAGE, 'gpu', n, parties, str(tmp_path), seed=seed)
    compare(ref, dev, parties)
    if parties == 1:
        # the REFERENCE's own run has one element off by 2^48 with these keys -- and so has the engine's, same element
        assert ref[0]['outliers_reference_trunc_mask'] == dev[0]['outliers_reference_trunc_mask'] == 1
        assert ref[0]['max_abs_error'] == 2.0**48 and dev[0]['max_abs_error_without_outliers'] < 0.01


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, 'mpyc')), reason='no staged reference copy (_refstage/)')
def test_fxp_product_production_prf_on_gpu(tmp_path):
    """the same program with the device PRF at n = 10^6, one party and three: correct to fixed-point accuracy (apart from
    the reference's own short-mask outliers, counted separately)"""
    for parties in (1, 3):
        dev = run_program(STAGE, 'gpu', 1_000_000, parties, str(tmp_path), prf='chacha')
        assert dev[0]['prss_prf'] == 'chacha'
        assert dev[0]['max_abs_error_without_outliers'] < 0.01 and dev[0]['outliers_reference_trunc_mask'] <= 20
