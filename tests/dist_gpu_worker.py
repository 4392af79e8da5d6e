"""Worker for tests/test_gpu_multigpu.py: party-major exchange + Lagrange recombination THROUGH THE KERNELS under
torch.distributed.  Backend from DIST_BACKEND: `nccl` (= RCCL; one rank per GPU, so world 1 on a 1-GPU box) or `gloo`
(two ranks sharing GPU 0, device tensors staged through the host by mpyc_amd.multigpu -- checks the N > 1 data flow
on a 1-GPU box).  Results are compared with the oracle's recombination on the host."""
import os
import random
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mpyc_amd import multigpu  # noqa: E402
from mpyc_amd.engine import DevArray, FieldContext, ints_to_np  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)


def main():
    backend = os.environ.get('DIST_BACKEND', 'nccl')
    torch.cuda.set_device(0 if backend == 'gloo' else int(os.environ.get('LOCAL_RANK', '0')))
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    for modulus, binary in ((2**61 - 1, False), (2**128 - 173, False), (2**96 - 17, False), (0x11b, True)):
        F = po.Field(modulus, binary)
        ctx = FieldContext(modulus, binary, device=torch.cuda.current_device())
        eb = ctx.elem_bytes
        for n, k in ((1001, 7), (4099, 3)):
            r = random.Random(1234 + n)                       # same rows on every rank
            rows = [[r.randrange(F.order) for _ in range(n)] for _ in range(k)]
            xs = [((3 + j) % k) + 1 for j in range(k)]        # rotated x-coordinates, as in _reshare (runtime.py:658,677)
            lam = po.recombination_vector(F, xs, 0)
            want = po.np_recombine(F, list(zip(xs, rows)))     # thresha.py:119-132 restated
            dev_rows = {j: ctx.from_numpy(ints_to_np(rows[j], eb)).t for j in range(k) if multigpu.row_owner(j, world) == rank}
            template = ctx.empty(0).t
            lo, hi = multigpu.shard_range(n, rank, world)
            y = multigpu.recombine_party_major(ctx, dev_rows, list(range(k)), lam, n, template=template)
            assert y.to_ints() == want[lo:hi], ('all-to-all', hex(modulus), n, k, rank)
            for chunks in (2, 5):                              # exchange pipelined with the recombination kernel
                yp = multigpu.recombine_party_major(ctx, dev_rows, list(range(k)), lam, n, template=template, chunks=chunks)
                assert yp.to_ints() == want[lo:hi], ('pipelined', chunks, hex(modulus), n, k, rank)
            # all-gather form
            pg = multigpu.PartyMajorGather(k, n, template)
            for j, t_ in dev_rows.items():
                pg.block_row(j).copy_(t_)
            pg.gather()
            y2 = ctx.recombine([DevArray(ctx, pg.row(j)[lo:hi], hi - lo) for j in range(k)], lam)
            assert y2.to_ints() == want[lo:hi], ('all-gather', hex(modulus), n, k, rank)
            full = ctx.recombine([DevArray(ctx, pg.row(j), n) for j in range(k)], lam)
            assert full.to_ints() == want
            # inverse exchange: slices back to whole rows on their owners
            sl = multigpu.exchange_party_major(dev_rows, list(range(k)), n, template=template)
            back = {j: torch.zeros_like(t_) for j, t_ in dev_rows.items()}
            multigpu.scatter_party_major(sl, list(range(k)), n, back)
            for j, t_ in dev_rows.items():
                assert torch.equal(back[j], t_)
    dist.barrier()
    if rank == 0:
        print('DIST_GPU_OK', backend, world)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
