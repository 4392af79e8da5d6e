"""Party program for the API-level parity test and the `api` leg of bench.py.

An ordinary MPyC program: `a * b` on SecFld arrays followed by `mpc.output`, i.e. Runtime.np_multiply ->
_reshare -> output (runtime.py:1096-1141, 603-689, 513-600) driven through the public API.  It runs either on the
unmodified reference (API_MODE=ref) or with mpyc_amd.install() substituted underneath (API_MODE=gpu; cpuctx = the
Python-integer stand-in of tests/cpuctx.py for the build container).  Launch it like any MPyC program:

    API_MODE=gpu API_N=1000000 python api_program.py            # one party, t = 0
    API_MODE=gpu API_N=1000000 python api_program.py -M3        # three local parties over TCP, t = 1

Environment:
    API_PRIME    modulus (default 2^61-1)             API_N      elements per array
    API_REPS     timed repetitions of the gate        API_CHAIN  multiplications per repetition (default 1)
    API_WARMUP   untimed repetitions before the timed ones (default 0)
    API_SEED     if set: secrets.randbelow (reference, thresha.py:58) / mpyc_amd.thresha.randbelow (mirror) are
                 replaced by random.Random(seed + pid).randrange, so both runs draw identical coefficients
    API_DIGEST   if set: path prefix; party i writes <prefix>.<i>.json with the SHA-256 of every share row that
                 np_random_split produced, every array np_recombine returned and every opened result (canonical
                 little-endian limbs), plus its timings
"""
import hashlib
import json
import os
import random
import sys
import time

MODE = os.environ.get('API_MODE', 'ref')
HERE = os.path.dirname(os.path.abspath(__file__))
if MODE != 'ref':
    for p_ in (HERE, os.path.dirname(HERE)):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import mpyc_amd
    mpyc_amd.install()
    if MODE == 'cpuctx':
        from cpuctx import use_cpu_contexts
        use_cpu_contexts()

import numpy as np                      # noqa: E402
from mpyc.runtime import mpc            # noqa: E402
from mpyc import thresha                # noqa: E402

P = int(os.environ.get('API_PRIME', str(2**61 - 1)))
N = int(os.environ.get('API_N', '100000'))
REPS = int(os.environ.get('API_REPS', '3'))
CHAIN = int(os.environ.get('API_CHAIN', '1'))
WARMUP = int(os.environ.get('API_WARMUP', '0'))
SEED = os.environ.get('API_SEED')
DIGEST = os.environ.get('API_DIGEST')
WIDTH = (P.bit_length() + 7) // 8

digests = []


def canon_bytes(x):
    """canonical little-endian bytes of a share row / field array, whichever representation it has"""
    if hasattr(x, 'to_wire'):                     # device array (mirror)
        return x.to_wire()
    v = getattr(x, 'value', x)                    # reference field array -> object ndarray
    v = np.asarray(v).reshape(-1)
    if WIDTH <= 8:
        return v.astype(np.uint64).astype(f'<u{8}').tobytes() if WIDTH == 8 else \
            b''.join(int(e).to_bytes(WIDTH, 'little') for e in v)
    return b''.join(int(e).to_bytes(WIDTH, 'little') for e in v)


def note(tag, x):
    if DIGEST:
        digests.append([tag, hashlib.sha256(canon_bytes(x)).hexdigest()])


def hook_sharing():
    """wrap whatever thresha.np_random_split / np_recombine currently are (the reference's functions, or the ones
    install() routed to the device); the runtime looks them up at call time (runtime.py:479-485, 566-572, 648-656)"""
    split, rec = thresha.np_random_split, thresha.np_recombine

    def np_random_split(field, s, t, m):
        shares = split(field, s, t, m)
        for i in range(m):
            note(f'split[{i}]', shares[i])
        return shares

    def np_recombine(field, points, x_rs=0):
        y = rec(field, points, x_rs)
        note('recombine', y)
        return y
    thresha.np_random_split, thresha.np_recombine = np_random_split, np_recombine


def hook_randomness(pid):
    rnd = random.Random(int(SEED) + pid)
    draw = rnd.randrange
    if MODE == 'ref':
        import secrets
        secrets.randbelow = draw                  # thresha.py:37,58 call secrets.randbelow
    else:
        import mpyc_amd.thresha as gth
        gth.randbelow = draw


async def main():
    await mpc.start()
    pid, m = mpc.pid, len(mpc.parties)
    if SEED is not None:
        hook_randomness(pid)
    if DIGEST:
        hook_sharing()
    secfld = mpc.SecFld(modulus=P)
    F = secfld.field
    rng = np.random.default_rng(20260925)
    if P < 2**63:
        xa, xb = rng.integers(0, P, size=N, dtype=np.int64), rng.integers(0, P, size=N, dtype=np.int64)
    else:
        xa = np.array([int.from_bytes(rng.bytes(WIDTH + 8), 'little') % P for _ in range(N)], dtype=object)
        xb = np.array([int.from_bytes(rng.bytes(WIDTH + 8), 'little') % P for _ in range(N)], dtype=object)
    t0 = time.perf_counter()
    a = mpc.input(secfld.array(F.array(xa)), senders=0)          # _distribute: np_random_split at party 0
    b = mpc.input(secfld.array(F.array(xb)), senders=0)
    await mpc.gather(a, b)
    t_input = time.perf_counter() - t0
    sync = getattr(sys.modules.get('torch'), 'cuda', None)
    ctx = None
    if MODE == 'gpu':
        import mpyc_amd.finfields as gff
        ctx = gff._context(F)
        ctx.set_timing(True, accumulate=True)       # one event pair per libffgpu call: GPU-busy numerator
    prof = None
    if os.environ.get('API_CPROFILE') and pid == 0:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    times = []
    for r in range(WARMUP + REPS):
        if sync is not None and MODE == 'gpu':
            sync.synchronize()
        if r == WARMUP:
            times.clear()
            if ctx is not None:
                ctx.busy_ms()                        # reset the GPU-busy accumulator
        t0 = time.perf_counter()
        c = a * b                                                # np_multiply (+ _reshare when t > 0)
        for _ in range(CHAIN - 1):
            c = c * b
        y = await mpc.output(c)                                  # np_recombine at every party
        if hasattr(y, 'device_array'):
            y.device_array                                       # materialise a deferred recombination
            if sync is not None and MODE == 'gpu':
                sync.synchronize()
        times.append(time.perf_counter() - t0)
        note('opened', y)
    if prof is not None:
        prof.disable()
        prof.dump_stats(os.environ['API_CPROFILE'])
    busy_ms, calls = ctx.busy_ms() if ctx is not None else (None, None)
    want = [int(u) * int(v) % P for u, v in zip(xa[:4].tolist(), xb[:4].tolist())]
    for _ in range(CHAIN - 1):
        want = [w * int(v) % P for w, v in zip(want, xb[:4].tolist())]
    got = [int(e) for e in (y[:4].value if MODE == 'ref' else np.asarray(y[:4].value)).tolist()]
    assert got == want, (got, want)
    sent = sum(getattr(pty.protocol, 'nbytes_sent', 0) for pty in mpc.parties if getattr(pty, 'protocol', None) is not None)
    await mpc.shutdown()
    ipcw = sys.modules.get('mpyc_amd.ipcwire')
    ipc = bool(ipcw is not None and ipcw.ENABLED and ipcw.stats['exported'] + ipcw.stats['imported'] > 0) if MODE == 'gpu' else False
    dev_index = sync.current_device() if (sync is not None and MODE == 'gpu') else None
    res = {'pid': pid, 'device': dev_index, 'ipc_wire': ipc, 'ipc_stats': dict(ipcw.stats) if ipcw is not None else None, 'bytes_sent': sent, 'm': m, 't': mpc.threshold, 'n': N, 'prime_bits': P.bit_length(), 'mode': MODE, 'chain': CHAIN,
           'input_s': t_input, 'times_s': times, 'gpu_busy_ms': busy_ms, 'gpu_calls': calls, 'digests': digests}
    if DIGEST:
        with open(f'{DIGEST}.{pid}.json', 'w') as fh:
            json.dump(res, fh)
    if pid == 0:
        print('API_RESULT ' + json.dumps({k: v for k, v in res.items() if k != 'digests'}), flush=True)


mpc.run(main())
