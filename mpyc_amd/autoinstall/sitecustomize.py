"""Put this directory on PYTHONPATH and set MPYC_GPU=1 to run an unmodified MPyC program on the MI355X engine:

    MPYC_GPU=1 PYTHONPATH=<repo>/mpyc_amd/autoinstall:<repo>:<mpyc checkout> python demos/np_aes.py -M3

Python imports `sitecustomize` at start-up in every process, including the m-1 party processes that mpyc
spawns itself (runtime.py:5157-5189 re-runs sys.argv with the inherited environment), so every party gets
mpyc_amd.install() before mpyc creates its first field.  Same pattern as mpyc's own MPYC_NOGMPY / MPYC_NONUMPY
environment switches (mpyc/__init__.py:155-166); a maintainer would put the two lines below into
mpyc/__init__.py instead."""
import os

if os.environ.get('MPYC_GPU') == '1':
    import mpyc_amd
    mpyc_amd.install()
