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


def _bench_lines(args, env_extra, timeout=900):
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    bare = [ln for ln in lines if ln.startswith('{')]
    assert len(bare) == 1 and lines[-1] is not None and lines[-1] == bare[0], [ln[:80] for ln in lines]
    assert len(bare[0]) < 8000
    detail = [ln for ln in lines if ln.startswith('# detail ')]
    assert len(detail) == 1
    return json.loads(bare[0]), json.loads(detail[0][len('# detail '):])


@pytest.mark.gpu
def test_bench_eight_ranks_control_flow_on_one_gpu():
    """The command the driver runs for the scaling record, with the 8 ranks forced onto GPU 0 (gloo, since RCCL refuses
    two ranks on one device): self-launch under torch.distributed.run, headline + roofline before the first collective
    of the configs[3] section, the section itself on 8 ranks, one compact line that parses."""
    line, detail = _bench_lines(['--gpus', '8', '--steps', '2', '--warmup', '1'],
                                {'FFGPU_BENCH_DEVICE': '0', 'FFGPU_BENCH_LEG_TIMEOUT': '600'})
    assert line['n_gpus'] == 8 and line['distributed']['world_size'] == 8 and line['distributed']['backend'] == 'gloo'
    assert len(line['distributed']['ranks']) == 8 and line['distributed']['distinct_devices'] == 1
    assert all(r['pci_bus_id'] for r in line['distributed']['ranks'])
    assert line['config']['workload'].startswith('configs[1]') and line['config']['n_per_gpu'] == 10_000_000
    assert line['scaling'] == 'weak' and line['value'] > 0 and line['steps'] == 2
    assert line['roofline']['bound'] == 'hbm' and 0 < line['roofline']['frac'] < 1
    assert 'cpu_baseline' not in line                      # rank 0 at N = 1 only
    assert 'error' not in line['multi_gpu'], line['multi_gpu']
    assert detail['multi_gpu']['config']['n_gpus'] == 8
    for leg in ('gate_sharded', 'party_major_all_to_all', 'party_major_allgather'):
        assert line['multi_gpu'][leg]['ms_per_step'] > 0
    # every rank's parity checks of the section, all-reduced (MIN) into one flag
    assert line['multi_gpu']['parity_passed_on_every_rank'] is True and detail['multi_gpu']['parity']['failures_on_rank_0'] == []


@pytest.mark.gpu
def test_bench_two_ranks_party_major_layout_on_one_gpu():
    """`--layout party-major` (only the configs[3] section, full step count) with two ranks on GPU 0 over gloo: the
    exchange + recombination legs run, their parity checks pass on both ranks and the flag says so."""
    line, detail = _bench_lines(['--gpus', '2', '--steps', '3', '--warmup', '1', '--layout', 'party-major'],
                                {'FFGPU_BENCH_DEVICE': '0', 'FFGPU_BENCH_LEG_TIMEOUT': '600', 'FFGPU_BENCH_N': '2000000'})
    assert line['n_gpus'] == 2 and line['distributed']['backend'] == 'gloo' and line['scaling'] == 'weak'
    mg = line['multi_gpu']
    assert 'error' not in mg and mg['parity_passed_on_every_rank'] is True
    for leg in ('gate_sharded', 'party_major_all_to_all', 'party_major_all_to_all_pipelined', 'party_major_allgather'):
        assert mg[leg]['ms_per_step'] > 0
    assert detail['multi_gpu']['config']['steps'] == 3 and detail['multi_gpu']['config']['n_gpus'] == 2
    assert 'kernels' not in detail                          # party-major layout: no extras


@pytest.mark.gpu
def test_bench_refuses_two_rccl_ranks_on_one_device():
    """Two NCCL ranks that land on ONE device must not produce a scaling number: the run ends with a non-zero status and
    a line that says `invalid` (RCCL itself refuses the communicator on current versions -- either way rc != 0)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', FFGPU_BENCH_DEVICE='0', FFGPU_BENCH_BACKEND='nccl',
               FFGPU_BENCH_N='1000000', FFGPU_BENCH_LEG_TIMEOUT='60')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-multi-gpu-leg'], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    bare = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if bare:                                                # the communicator came up: the line must carry the verdict
        import json
        assert json.loads(bare[-1])['scaling'].startswith('invalid: ranks share a device')


@pytest.mark.gpu
def test_bench_single_gpu_line_has_every_required_object():
    line, detail = _bench_lines(['--steps', '5', '--warmup', '2', '--no-api-leg'], {})
    assert line['n_gpus'] == 1 and line['dtype'] == 'u64' and line['unit'] == 'field-ops/s'
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'ms_per_launch', 'bytes_per_launch'):
        assert key in line['roofline'], key
    cb = line['cpu_baseline']
    assert cb['kind'] in ('reference', 'port') and cb['value'] > 0 and cb['host_cores'] >= cb['cores'] >= 1
    assert 'extras_error' not in line, line.get('extras_error')
    assert len(detail['kernels']) > 40
