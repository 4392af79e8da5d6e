import ctypes
import json
import os

# the parity tests use modest shapes: send them to the matrix-core product already (production default: 8e7 MACs)
os.environ.setdefault('FFGPU_MM_MFMA_MIN', '1.6e7')
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


@pytest.fixture(scope='session')
def hostcheck():
    """tests/_hostcheck.so: mpyc_amd/csrc/fields.hpp compiled with g++ (test-only harness)."""
    src = os.path.join(ROOT, 'tests', 'hostcheck.cpp')
    dst = os.path.join(ROOT, 'tests', '_hostcheck.so')
    deps = [src, os.path.join(ROOT, 'mpyc_amd', 'csrc', 'fields.hpp'),
            os.path.join(ROOT, "mpyc_amd", "csrc", "policy_build.hpp"),
            os.path.join(ROOT, "mpyc_amd", "csrc", "rng.hpp"), os.path.join(ROOT, "mpyc_amd", "csrc", "bitslice.hpp")]
    if any(_newer(d, dst) for d in deps):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-o', dst, src], check=True)
    return ctypes.CDLL(dst)


@pytest.fixture(scope='session')
def golden_fields():
    with open(os.path.join(GOLDEN, 'fields.json')) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def golden_wide():
    """three-limb primes (129..192 bits), same case format as fields.json"""
    with open(os.path.join(GOLDEN, 'wide.json')) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def golden_sbox():
    with open(os.path.join(GOLDEN, 'sbox.json')) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def coracle():
    from oracle import coracle as co
    src = os.path.join(ROOT, 'oracle', 'fforacle.c')
    dst = os.path.join(ROOT, 'oracle', 'liboracle.so')
    if _newer(src, dst):
        co.build()
    return co


# Development aid (build container, no GPU): MPYC_AMD_CPUCTX=1 runs the API-level `-m gpu` tests of the mirror on
# tests/cpuctx.py's Python-integer context, to debug HOST logic before spending GPU time.  Never set on the GPU box.
if os.environ.get('MPYC_AMD_CPUCTX') == '1':
    import torch as _torch
    from cpuctx import use_cpu_contexts as _use
    _use()
    _torch.cuda.is_available = lambda: True
