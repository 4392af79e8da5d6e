"""Worker for tests/test_multiprocess.py (world_size 2, gloo, CPU): exercises the sharding and
the party-major exchange bookkeeping of mpyc_amd.multigpu on plain limb tensors."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpyc_amd import multigpu  # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    for n in (0, 1, 5, 1000, 1001):
        # element sharding covers [0, n) exactly once, in order
        ranges = [multigpu.shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
    for limbs in (1, 2):          # one-limb (n,) and two-limb (n, 2) elements
        for n in (7, 1000):
            k = 5
            shape = (n,) if limbs == 1 else (n, 2)
            full = [torch.arange(n * limbs, dtype=torch.int64).reshape(shape) + 1_000_000 * (j + 1) for j in range(k)]
            local = {j: full[j] for j in range(k) if multigpu.row_owner(j, world) == rank}
            row_ids = [3, 0, 4, 1, 2]           # any order (x-coordinates rotate in _reshare)
            got = multigpu.exchange_party_major(local, row_ids, n)
            lo, hi = multigpu.shard_range(n, rank, world)
            for idx, j in enumerate(row_ids):
                assert torch.equal(got[idx], full[j][lo:hi]), (rank, j)
            # concatenating the shards of all ranks gives back the whole row
            mine = got[0]
            gathered = [torch.empty_like(full[0][slice(*multigpu.shard_range(n, r, world))]) for r in range(world)]
            dist.all_gather(gathered, mine) if all(g.shape == mine.shape for g in gathered) else None
            # the inverse exchange puts the slices back into whole rows on their owners
            back = {j: torch.zeros_like(full[j]) for j in local}
            multigpu.scatter_party_major(got, row_ids, n, back)
            for j in local:
                assert torch.equal(back[j], full[j]), (rank, j)
            # all-gather form: owners write their rows into the block, everyone reads whole rows afterwards
            pg = multigpu.PartyMajorGather(k, n, full[0])
            for j in local:
                pg.block_row(j).copy_(full[j])
            pg.gather()
            for j in range(k):
                assert torch.equal(pg.row(j), full[j]), (rank, j)
            assert pg.bytes_received == (world - 1) * pg.rows_per_rank * full[0].numel() * 8
            # ragged rows are refused before anything is sent
            if local:
                bad = dict(local)
                j0 = next(iter(bad))
                bad[j0] = bad[j0][:-1]
                try:
                    multigpu.exchange_party_major(bad, row_ids, n)
                    raise AssertionError('ragged row accepted')
                except ValueError:
                    pass
    # exchange pipelined with the recombination (recombine_party_major(chunks=...)): a stand-in context whose
    # "kernel" is a wrapping int64 dot product -- only the chunk / buffer bookkeeping is under test here
    class StubCtx:
        def empty(self, n):
            from mpyc_amd.engine import DevArray
            return DevArray(self, torch.zeros(n, dtype=torch.int64), n)

        def recombine(self, rows, lam, w=1, out=None):
            out = out or self.empty(rows[0].n)
            acc = torch.zeros(rows[0].n, dtype=torch.int64)
            for r, l_ in zip(rows, lam):
                acc += r.t * l_
            out.t.copy_(acc)
            return out
    ctx = StubCtx()
    for n in (5, 1000, 4099):
        k, lam = 5, [3, 1, 4, 1, 5]
        full = [torch.arange(n, dtype=torch.int64) * (j + 2) + 17 * j for j in range(k)]
        local = {j: full[j] for j in range(k) if multigpu.row_owner(j, world) == rank}
        row_ids = [2, 0, 3, 1, 4]
        lo, hi = multigpu.shard_range(n, rank, world)
        want = sum(full[j][lo:hi] * lam[i] for i, j in enumerate(row_ids))
        for chunks in (1, 2, 3, 8):
            got = multigpu.recombine_party_major(ctx, local, row_ids, lam, n, template=full[0], chunks=chunks)
            assert torch.equal(got.t, want), (rank, n, chunks)
        recv = [torch.empty(hi - lo, dtype=torch.int64) for _ in row_ids]
        got = multigpu.recombine_party_major(ctx, local, row_ids, lam, n, template=full[0], chunks=3, out=ctx.empty(hi - lo), recv=recv)
        assert torch.equal(got.t, want), (rank, n, 'recv')
    rows = multigpu.allgather_rows(torch.full((11,), rank, dtype=torch.int64))
    assert [int(r[0]) for r in rows] == list(range(world))
    # MAX-over-ranks reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world
    dist.barrier()
    if rank == 0:
        print('DIST_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
