"""pytest plugin (-p refplugin) that runs mpyc_amd.install() before the REFERENCE's own test modules are
imported, so that lschoe/mpyc's tests/test_finfields.py, test_thresha.py, test_runtime.py exercise the
substituted array type and sharing functions.  MPYC_AMD_CPUCTX=1 (build container, no GPU) swaps in the
Python-integer context of tests/cpuctx.py; on a GPU box the kernels of libffgpu run."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

import mpyc_amd  # noqa: E402

if os.environ.get('MPYC_AMD_LIST_MIN') is not None:
    mpyc_amd.list_path_min = int(os.environ['MPYC_AMD_LIST_MIN'])
mpyc_amd.install()
if os.environ.get('MPYC_AMD_CPUCTX') == '1':
    from cpuctx import use_cpu_contexts
    use_cpu_contexts()
if os.environ.get('MPYC_AMD_LAZY_MIN') is not None:
    import mpyc_amd.finfields as _gff
    _gff.HostView.lazy_min = int(os.environ['MPYC_AMD_LAZY_MIN'])
