"""Start-up confirmation of the PRSS mode between the parties (mpyc_amd._hook_prss_confirmation): a party in
production mode (ChaCha PRF on the device) must not compute next to a party on the reference PRF or on another round count
-- the shares of a common subset key would be inconsistent and openings silently wrong.  tests/prss_mode_program.py, three
local parties (-M3) on the Python-integer context (no GPU): agreement starts and computes; disagreement and a silent peer end
party 0 with a RuntimeError that names the cause; the default mode adds nothing (no message, plain-mpyc peers interoperate)."""
import os
import signal
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = next((r for r in ('/root/reference', os.path.join(ROOT, '_refstage')) if os.path.isdir(os.path.join(r, 'mpyc'))), None)
PROG = os.path.join(ROOT, 'tests', 'prss_mode_program.py')


def run(modes, tmp, port, timeout=120):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, REF])
    for k_ in ('MPYC_AMD_PRSS_PRF', 'MPYC_AMD_PRSS_ROUNDS', 'MPYC_AMD_CPUCTX'):
        env.pop(k_, None)
    env.update(PM_MODES=','.join(modes), PM_CPUCTX='1', PM_TIMEOUT='4', MPYC_AMD_IPC_WIRE='0')
    cmd = [sys.executable, PROG, '--no-log', f'-M{len(modes)}', '-B', str(port)]
    # own process group: the parties spawned by party 0 (runtime.py:5171-5189) are ended with it when a start-up fails
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=tmp, env=env, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
    finally:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    return p.returncode, out


@pytest.mark.skipif(REF is None, reason='no importable mpyc checkout')
def test_parties_confirm_the_prss_mode(tmp_path):
    rc, out = run(['chacha20'] * 3, str(tmp_path), 11800)
    assert rc == 0 and 'PM_STARTED chacha20/v1' in out, out[-2000:]
    rc, out = run(['shake'] * 3, str(tmp_path), 11810)
    assert rc == 0 and 'PM_STARTED shake' in out, out[-2000:]


@pytest.mark.skipif(REF is None, reason='no importable mpyc checkout')
def test_disagreement_on_the_prss_mode_fails_loudly(tmp_path):
    rc, out = run(['chacha20', 'chacha20', 'chacha12'], str(tmp_path), 11820)
    assert rc != 0 and 'disagree on the PRSS PRF' in out and 'PM_STARTED' not in out, out[-2000:]
    # a peer on the default mode sends no tag: the production-mode party gives up after the timeout and says why
    rc, out = run(['chacha20', 'shake', 'chacha20'], str(tmp_path), 11830)
    assert rc != 0 and ('did not confirm its mode' in out or 'disagree on the PRSS PRF' in out) and 'PM_STARTED' not in out, out[-2000:]
