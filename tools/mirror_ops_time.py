"""Time the common FieldArray operations of the mirror at realistic sizes: anything in the millisecond range
for a memory-bound op points at a host detour."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpyc_amd import finfields, thresha
P = 2**61 - 1
F = finfields.GF(P)
n = 4_000_000
rng = np.random.default_rng(1)
A = rng.integers(0, P, size=n)
B = rng.integers(1, P, size=n)
a, b = F.array(A), F.array(B)
m2 = a.reshape(2000, 2000)
v = F.array(rng.integers(0, P, size=2000))
s1k = F.array(rng.integers(0, P, size=(1000, 1000)))


def T(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
        if hasattr(r, '_dev'):
            r._dev            # materialise deferred results
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f'{name:38s} {dt*1e6:10.1f} us')


T('construct from int64 ndarray (4M)', lambda: F.array(A))
T('a + b', lambda: a + b)
T('a * b (materialised)', lambda: (a * b)._dev)
T('a * 3 + 7', lambda: a * 3 + 7)
T('-a', lambda: -a)
T('a ** 5', lambda: a ** 5)
T('a.reciprocal() (b nonzero)', lambda: b.reciprocal())
T('a == b', lambda: a == b)
T('a[::2]', lambda: a[::2])
T('a[1000:3000000]', lambda: a[1000:3000000])
T('concatenate((a, b))', lambda: np.concatenate((a, b)))
T('np.roll(a, 5)', lambda: np.roll(a, 5))
T('np.flip(a)', lambda: np.flip(a))
T('m2.T (2000x2000)', lambda: m2.T)
T('m2 @ v', lambda: m2 @ v)
T('v @ m2', lambda: v @ m2)
T('m2 @ m2 (2000^3)', lambda: m2 @ m2, reps=2)
T('np.sum(a)', lambda: np.sum(a))
T('np.sum(m2, axis=0)', lambda: np.sum(m2, axis=0))
T('np.sum(m2, axis=1)', lambda: np.sum(m2, axis=1))
T('a @ b (inner product)', lambda: a @ b)
T('a.to_wire()', lambda: a.to_wire(), reps=2)
w = a.to_wire()
T('from_wire', lambda: F.array.from_wire(w), reps=2)
T('np.tile(m2[:100], (2, 2))', lambda: np.tile(m2[:100], (2, 2)))
mask = A % 2 == 0
T('np.where(mask, a, b)', lambda: np.where(mask, a, b), reps=2)
T('np.outer(v, v)', lambda: np.outer(v, v))
T('np.convolve(a[:20000], v[:500])', lambda: np.convolve(a[:20000], v[:500]), reps=2)
T('np.prod(a)', lambda: np.prod(a), reps=2)
T('np.trace(s1k)', lambda: np.trace(s1k))
T('np.linalg.inv(256x256)', lambda: np.linalg.inv(s1k[:256, :256]), reps=2)
T('np.linalg.det(256x256)', lambda: np.linalg.det(s1k[:256, :256]), reps=2)
T('np_random_split m=3 t=1', lambda: thresha.np_random_split(F, a, 1, 3))
sh = thresha.np_random_split(F, a, 1, 3)
T('np_recombine k=2 (materialised)', lambda: thresha.np_recombine(F, [(1, sh[0]), (2, sh[1])]))
T('a.value (host objects, 4M)', lambda: F.array(A).value, reps=1)
