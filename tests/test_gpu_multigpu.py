"""Multi-GPU data path through the kernels, on however many GPUs the box has:
  * a process that never initialised torch.distributed is a world of one: the party-major functions degenerate to
    local views and the recombination runs on the kernels;
  * RCCL process group (world = number of GPUs; 1 on the build pool's boxes): exchange, all_gather_into_tensor and
    the recombination kernels under the `nccl` backend;
  * two ranks sharing GPU 0 over gloo (device tensors staged through the host): the N > 1 data flow -- column
    slices travelling between ranks, every rank recombining its own range -- checked against the oracle."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_worker(backend, nproc, port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'dist_gpu_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and 'DIST_GPU_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


@pytest.mark.gpu
def test_party_major_world_of_one_runs_on_kernels():
    import torch
    from mpyc_amd import multigpu
    from mpyc_amd.engine import FieldContext, ints_to_np
    from oracle import pyoracle as po
    modulus = 2**64 - 189
    F = po.Field(modulus, False)
    ctx = FieldContext(modulus, device=0)
    n, k = 5003, 7
    r = random.Random(5)
    rows = [[r.randrange(modulus) for _ in range(n)] for _ in range(k)]
    xs = [2, 5, 1, 7, 3, 6, 4]
    lam = po.recombination_vector(F, xs, 0)
    want = po.np_recombine(F, list(zip(xs, rows)))
    local = {j: ctx.from_numpy(ints_to_np(rows[j], 8)).t for j in range(k)}
    y = multigpu.recombine_party_major(ctx, local, list(range(k)), lam, n)
    assert y.to_ints() == want
    for chunks in (2, 3, 16):
        assert multigpu.recombine_party_major(ctx, local, list(range(k)), lam, n, chunks=chunks).to_ints() == want
    with pytest.raises(ValueError):                                      # ragged row: refused, never read out of bounds
        multigpu.recombine_party_major(ctx, {**local, 3: local[3][:-1]}, list(range(k)), lam, n)
    with pytest.raises(ValueError):
        ctx.recombine([ctx.from_numpy(ints_to_np(rows[0], 8)), ctx.from_numpy(ints_to_np(rows[1][:-2], 8))], lam[:2])
    assert torch.cuda.is_available()


@pytest.mark.gpu
def test_party_major_under_rccl():
    import torch
    out = _run_worker('nccl', torch.cuda.device_count(), 29541)
    assert f'DIST_GPU_OK nccl {torch.cuda.device_count()}' in out


@pytest.mark.gpu
def test_party_major_two_ranks_one_gpu_gloo_staged():
    out = _run_worker('gloo', 2, 29542)
    assert 'DIST_GPU_OK gloo 2' in out
