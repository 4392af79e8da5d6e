"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: element sharding, the party-major
column exchange, all-gather, and the MAX reduction bench.py uses.  The field arithmetic itself
needs a GPU and is covered by the -m gpu tests; here only torch.distributed bookkeeping runs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exchange_world2():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533', os.path.join(ROOT, 'tests', 'dist_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK' in r.stdout


def test_shard_range_properties():
    from mpyc_amd.multigpu import shard_range
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 9, 10**7, 10**7 + 3):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
