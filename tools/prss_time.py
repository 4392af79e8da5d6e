"""PRSS share generation (np_pseudorandom_share, thresha.py:163-173) for one party of m = 7, t = 3: parallel host SHAKE128
expansion of the C(6,3) = 20 subset keys + device combination, next to hashlib expanding the same streams one by one."""
import os, sys, time, itertools, hashlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from mpyc_amd import finfields, thresha
F = finfields.GF(2**61 - 1)
m, t, i = 7, 3, 2
keys = {S: bytes([sum(S) % 256]) * 16 + bytes(S) for S in itertools.combinations(range(m), m - t) if i in S}
prfs = {S: thresha.PRF(k, F.order) for S, k in keys.items()}
print('subset keys for this party:', len(prfs), 'bytes per draw:', next(iter(prfs.values())).byte_length)
thresha.np_pseudorandom_share(F, m, i, prfs, b'warm', 1000)      # one-time costs: context, Lagrange scalars, staging buffer
for n in (10**5, 10**6, 10**7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = thresha.np_pseudorandom_share(F, m, i, prfs, b'uci', n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for prf in prfs.values():
        hashlib.shake_128(prf.key + b'uci').digest(n * prf.byte_length)
    dh = time.perf_counter() - t1
    print(f'n={n}: PRSS share on device {dt*1e3:.1f} ms ({n/dt/1e6:.2f} M shares/s); hashlib alone for the same streams {dh*1e3:.1f} ms')
