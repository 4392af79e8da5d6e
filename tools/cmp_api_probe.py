"""Secure comparison through the reference's public API under install(): `a < b` on SecInt(32) arrays (np_sgn / np_to_bits /
np_random_bits, runtime.py:1500-1600, 4391-4423, 4187-4273), one party, production PRF -- seconds per comparison of n elements;
run it under rocprofv3 --kernel-trace --stats to see which kernels carry it.   CMP_N (default 10^5), CMP_REPS."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, 'tests'), ROOT):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
N = int(os.environ.get('CMP_N', '100000'))
REPS = int(os.environ.get('CMP_REPS', '5'))
import mpyc_amd
mpyc_amd.install()
import numpy as np
import torch
from mpyc.runtime import mpc


async def main():
    await mpc.start()
    secint = mpc.SecInt(32)
    rng = np.random.default_rng(5)
    xa, xb = rng.integers(-2**30, 2**30, N), rng.integers(-2**30, 2**30, N)
    a = mpc.input(secint.array(xa), senders=0)
    b = mpc.input(secint.array(xb), senders=0)
    await mpc.gather(a, b)
    times, cpu = [], []
    for _ in range(REPS):
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        c = a < b
        share = await mpc.gather(c)
        if hasattr(share, 'device_array'):
            share.device_array
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        cpu.append(time.process_time() - c0)
    y = await mpc.output(c)
    assert (np.asarray(y) == (xa < xb)).all()
    await mpc.shutdown()
    print('CMP_RESULT n=%d field_bits=%d ms per comparison: %s' % (N, secint.field.order.bit_length(), [round(t * 1e3, 2) for t in times]), flush=True)
    print('CMP_CPU ms of process CPU time per comparison: %s' % [round(t * 1e3, 2) for t in cpu], flush=True)

mpc.run(main())
