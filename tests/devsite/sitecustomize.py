"""Build-container aid: like mpyc_amd/autoinstall/sitecustomize.py, plus MPYC_AMD_CPUCTX=1 swaps in the
Python-integer context of tests/cpuctx.py (no GPU here).  Used to debug host logic of multi-party demo runs."""
import os
import sys

if os.environ.get('MPYC_GPU') == '1':
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.dirname(os.path.dirname(here))):
        if p not in sys.path:
            sys.path.insert(0, p)
    import mpyc_amd
    mpyc_amd.install()
    if os.environ.get('MPYC_AMD_CPUCTX') == '1':
        from cpuctx import use_cpu_contexts
        use_cpu_contexts()
