"""Sharding one party's batch across the GPUs of a node.

Every element / secret / gate is independent, so the default layout is ELEMENT-SHARDED: rank g owns
the contiguous range shard_range(n, g, G) of every array and of every share row; mul, split and
recombine then need no collective at all.

The one layout with an exchange step is PARTY-MAJOR (whole share row j lives on rank j % G, e.g. when
the GPUs of a node stand in for the parties): before recombining, every rank needs its column range
of all k rows.  `exchange_party_major` does that as an all-to-all of column slices (each rank sends
n/G elements per row it owns to every peer: 1/G of the all-gather traffic); `allgather_rows` is the
all-gather the north star names, for callers that want whole rows everywhere.  Both move opaque limb
tensors; residues are never summed by a collective (u64 sums of k products overflow).

The exchange is written with batched point-to-point operations, which map onto xGMI links directly
under RCCL and also run under gloo (CPU tests); torch.distributed is plumbing here, the arithmetic
stays in libffgpu.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous element range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_owner(row: int, world: int) -> int:
    """Party-major placement: share row j lives on rank j % world."""
    return row % world


def exchange_party_major(local_rows: Dict[int, torch.Tensor], row_ids: Sequence[int], n: int,
                         group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """local_rows: {row id: full-length limb tensor (n, ...)} for the rows this rank owns
    (row_owner).  Returns, for every row in row_ids (in that order), this rank's column slice
    [lo, hi) -- ready to be passed to FieldContext.recombine."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n, rank, world)
    any_row = next(iter(local_rows.values())) if local_rows else None
    out: List[Optional[torch.Tensor]] = [None] * len(row_ids)
    ops = []
    recv_bufs = []
    for idx, j in enumerate(row_ids):
        owner = row_owner(j, world)
        if owner == rank:
            row = local_rows[j]
            out[idx] = row[lo:hi].contiguous()
            for peer in range(world):
                if peer == rank:
                    continue
                plo, phi = shard_range(n, peer, world)
                if phi > plo:
                    ops.append(dist.P2POp(dist.isend, row[plo:phi].contiguous(), peer, group=group, tag=idx))
        elif hi > lo:
            if any_row is None:
                raise ValueError('a rank that owns no row must pass a template via local_rows')
            buf = torch.empty((hi - lo,) + tuple(any_row.shape[1:]), dtype=any_row.dtype, device=any_row.device)
            recv_bufs.append((idx, buf))
            ops.append(dist.P2POp(dist.irecv, buf, owner, group=group, tag=idx))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for idx, buf in recv_bufs:
        out[idx] = buf
    for idx, t in enumerate(out):
        if t is None:       # empty shard
            out[idx] = torch.empty((0,) + tuple(any_row.shape[1:]), dtype=any_row.dtype, device=any_row.device)
    return out  # type: ignore[return-value]


def allgather_rows(local_row: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """All-gather of one full row per rank (rank r contributes row r): every rank ends up with all
    `world` rows.  RCCL all-gather over xGMI when the tensors are on the GPU."""
    world = dist.get_world_size(group)
    rows = [torch.empty_like(local_row) for _ in range(world)]
    dist.all_gather(rows, local_row.contiguous(), group=group)
    return rows


def recombine_party_major(ctx, local_rows: Dict[int, torch.Tensor], row_ids: Sequence[int],
                          lambdas: Sequence[int], n: int, group: Optional[dist.ProcessGroup] = None):
    """Exchange + local Lagrange recombination: returns this rank's slice of the secrets as a
    DevArray (element-sharded result)."""
    from .engine import DevArray
    slices = exchange_party_major(local_rows, row_ids, n, group)
    rows = [DevArray(ctx, t, t.shape[0]) for t in slices]
    if rows[0].n == 0:
        return ctx.empty(0)
    return ctx.recombine(rows, lambdas)
