"""Party program for the fixed-point product THROUGH the reference's API: `mpc.output(a * b)` on
SecFxp(32) arrays, i.e. Runtime.np_multiply -> np_trunc (runtime.py:1096-1141, 838-873) -> np_random_bits / _np_randoms
(runtime.py:4187-4273, 4062-4103: PRSS draws, thresha.py:163-217) -> output.  It runs on the unmodified reference
(FXP_MODE=ref) or with mpyc_amd.install() underneath (gpu; cpuctx = the Python-integer stand-in for the build container).

Both runs are made REPRODUCIBLE: the PRSS keys (secrets.token_bytes(16) in Runtime.threshold's setter, runtime.py:104-117)
come from a stream seeded by (FXP_SEED, party index), patched in BEFORE mpyc.runtime is imported; the coefficients of every
np_random_split from random.Random(seed + pid) (secrets.randbelow in the reference, the mirror's `randbelow` hook).  Every
party records SHA-256 digests (canonical little-endian limbs) of
    opened[k]   every array np_recombine returns, in call order: the opened squares of np_random_bits (:4256), the opened
                MASKED value `c` of np_trunc (:870), the final output
    y           this party's share of the truncated product (the value np_trunc returns, :872)
    out         the fixed-point result as float64 bytes
so tests/test_fxp_path.py can compare the two runs digest for digest.  With more than one party use `-M3`.

    FXP_MODE ref|gpu|cpuctx   FXP_N elements   FXP_SEED   FXP_DIGEST path prefix   FXP_REPS timed repetitions (default 1)
Prints `FXP_RESULT {...}` at party 0: seconds per secure product (np_multiply + np_trunc, the share on the device) and per
product + opening (which ends in the reference's conversion of n field elements to Python floats), max |error| against float64, and the number of
elements that are off by the reference's own short-mask quirk (see test_fxp_path.py).
"""
import hashlib
import json
import os
import random
import sys
import time

MODE = os.environ.get('FXP_MODE', 'ref')
N = int(os.environ.get('FXP_N', '2000'))
SEED = os.environ.get('FXP_SEED')
DIGEST = os.environ.get('FXP_DIGEST')
REPS = int(os.environ.get('FXP_REPS', '1'))
HERE = os.path.dirname(os.path.abspath(__file__))

import secrets                                       # noqa: E402

if SEED is not None:
    # the party index is on the command line before mpyc.runtime parses it (`-I <i>`, runtime.py:5171); party 0 has none
    _idx = int(sys.argv[sys.argv.index('-I') + 1]) if '-I' in sys.argv else 0
    _ctr = [0]

    def _token_bytes(nbytes=32):
        _ctr[0] += 1
        return hashlib.shake_128(f'fxp-keys {SEED} {_idx} {_ctr[0]}'.encode()).digest(nbytes)
    secrets.token_bytes = _token_bytes

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

digests = []


def canon_bytes(x, width):
    if hasattr(x, 'to_wire'):                     # device array (mirror)
        return x.to_wire()
    v = np.asarray(getattr(x, 'value', x)).reshape(-1)
    return b''.join(int(e).to_bytes(width, 'little') for e in v)


def main_hooks(pid, width):
    rec = thresha.np_recombine
    count = [0]

    def np_recombine(field, points, x_rs=0):
        y = rec(field, points, x_rs)
        if DIGEST:
            digests.append([f'opened[{count[0]}]', hashlib.sha256(canon_bytes(y, width)).hexdigest()])
        count[0] += 1
        return y
    thresha.np_recombine = np_recombine
    if SEED is not None:
        draw = random.Random(int(SEED) + pid).randrange
        if MODE == 'ref':
            secrets.randbelow = draw              # thresha.py:37,58 call secrets.randbelow
        else:
            import mpyc_amd.thresha as gth
            gth.randbelow = draw


async def main():
    await mpc.start()
    pid, m = mpc.pid, len(mpc.parties)
    secfxp = mpc.SecFxp(32)
    F = secfxp.field
    width = (F.order.bit_length() + 7) // 8
    main_hooks(pid, width)
    rng = np.random.default_rng(5)
    xa, xb = rng.uniform(-100, 100, N), rng.uniform(-100, 100, N)
    a = mpc.input(secfxp.array(xa), senders=0)
    b = mpc.input(secfxp.array(xb), senders=0)
    await mpc.gather(a, b)
    sync = getattr(sys.modules.get('torch'), 'cuda', None) if MODE == 'gpu' else None
    times, times_product = [], []
    y = c = share = None
    for _ in range(REPS):
        if sync is not None:
            sync.synchronize()
        t0 = time.perf_counter()
        c = a * b                                    # np_multiply + np_trunc (random bits and masks from PRSS, one masked opening)
        share = await mpc.gather(c)                  # this party's share of the truncated product (a field array): the product is done
        if hasattr(share, 'device_array'):
            share.device_array                       # (materialise a deferred device expression)
        if sync is not None:
            sync.synchronize()
        times_product.append(time.perf_counter() - t0)
        y = await mpc.output(c)                      # opening + the reference's conversion to n Python floats (sectypes.py:1426-1447)
        if sync is not None:
            sync.synchronize()
        times.append(time.perf_counter() - t0)
    if DIGEST:
        digests.append(['y', hashlib.sha256(canon_bytes(share, width)).hexdigest()])
        digests.append(['out', hashlib.sha256(np.asarray(y, dtype=np.float64).tobytes()).hexdigest()])
    diff = np.abs(np.asarray(y, dtype=float) - xa * xb)
    # np_trunc's mask for ARRAYS is f bits short (issubclass(sftype, SecureFixedPoint) is false for array types, runtime.py:852):
    # about one element in 10^6 wraps by 2^(l - f) = 2^48 -- in the reference and, bit for bit, here
    outliers = int(np.count_nonzero(diff > 1.0))
    rest = float(np.max(diff[diff <= 1.0])) if outliers < N else float('nan')
    res = {'pid': pid, 'm': m, 't': mpc.threshold, 'n': N, 'mode': MODE, 'field_bits': F.order.bit_length(), 'times_s': times,
           's_per_product_and_opening': min(times), 's_per_product': min(times_product), 'times_product_s': times_product,
           'max_abs_error': float(np.max(diff)),
           'outliers_reference_trunc_mask': outliers, 'max_abs_error_without_outliers': rest,
           'prss_prf': os.environ.get('MPYC_AMD_PRSS_PRF', 'shake') if MODE != 'ref' else 'shake', 'digests': digests}
    await mpc.shutdown()
    if DIGEST:
        with open(f'{DIGEST}.{pid}.json', 'w') as fh:
            json.dump(res, fh)
    if pid == 0:
        print('FXP_RESULT ' + json.dumps({k: v for k, v in res.items() if k != 'digests'}), flush=True)


mpc.run(main())
