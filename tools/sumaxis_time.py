import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mpyc_amd import finfields
F = finfields.GF(2**61 - 1)
a = F.array(np.random.randint(0, 2**61 - 1, size=(1_000_000, 8)))
for ax in (1, 0):
    s = a.sum(axis=ax); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        s = a.sum(axis=ax)
    torch.cuda.synchronize()
    print('sum axis', ax, s.shape, '%.1f us' % ((time.perf_counter() - t) / 10 * 1e6))
ref = np.asarray(a.value)
assert [int(v) for v in a.sum(axis=1).value[:5]] == [int(sum(r) % (2**61 - 1)) for r in ref[:5]]
assert [int(v) for v in a.sum(axis=0).value] == [int(sum(ref[:, j]) % (2**61 - 1)) for j in range(8)]
print('ok')
