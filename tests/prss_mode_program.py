"""Party program for the start-up confirmation of the PRSS mode (mpyc_amd._hook_prss_confirmation): every party
of a production-mode computation must run the same PRF and round count.  PM_MODES = comma-separated tags, one per party
('shake', 'chacha20', 'chacha12'): party i sets its own mode from entry i, then `mpc.start()` runs.  Prints `PM_STARTED <tag>`
at party 0 when the start-up (including the confirmation) went through; a disagreement ends party 0 with a RuntimeError."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p_ in (HERE, os.path.dirname(HERE)):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
idx = int(sys.argv[sys.argv.index('-I') + 1]) if '-I' in sys.argv else 0
import mpyc_amd                          # noqa: E402
mpyc_amd.install()
if os.environ.get('PM_CPUCTX') == '1':
    from cpuctx import use_cpu_contexts
    use_cpu_contexts()
import mpyc_amd.thresha as gth           # noqa: E402

mode = os.environ['PM_MODES'].split(',')[idx]
if mode.startswith('chacha'):
    gth.prss_prf, gth.prss_rounds = 'chacha', int(mode[6:])
else:
    gth.prss_prf = mode
mpyc_amd.prss_confirm_timeout = float(os.environ.get('PM_TIMEOUT', '4'))

from mpyc.runtime import mpc             # noqa: E402


async def main():
    await mpc.start()
    if mpc.pid == 0:
        print('PM_STARTED', gth.prss_mode_tag(), flush=True)
    secfld = mpc.SecFld(2**61 - 1)
    x = mpc._np_randoms(secfld, 5)       # PRSS draws (runtime.py:4062-4103) in whatever mode was confirmed
    y = await mpc.output(x - x)
    assert [int(v) for v in y] == [0] * 5
    await mpc.shutdown()

mpc.run(main())
