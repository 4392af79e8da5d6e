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

The exchange is written with batched point-to-point operations (ncclSend/ncclRecv pairs grouped per call under
RCCL: every pair of GPUs of an MI355X node has a direct xGMI link, so a pairwise exchange needs no relay; whether
RCCL actually routes them that way is to be read off `bench.py --gpus 8` -- it has not been measured on more than
one GPU) and also runs under gloo (CPU tests); torch.distributed is plumbing here, the arithmetic stays in libffgpu.
recombine_party_major(chunks=...) pipelines the exchange with the recombination kernel.
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


def _world(group) -> int:
    """world size; a process that never initialised torch.distributed is a world of one (single-GPU run)"""
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group) -> int:
    return dist.get_rank(group) if dist.is_initialized() else 0


def _host_staged(group) -> bool:
    """gloo has no point-to-point / all-gather for device tensors: stage through host memory.  Only the CPU
    tests and the 1-GPU validation of the N > 1 control flow take this route; RCCL moves device buffers."""
    return dist.is_initialized() and dist.get_backend(group) == 'gloo'


def _check_rows(local_rows: Dict[int, torch.Tensor], n: int):
    for j, row in local_rows.items():
        if row.shape[0] != n:
            raise ValueError(f'share row {j} has {row.shape[0]} elements, expected {n}')     # ragged rows: never exchanged


def part_range(lo: int, hi: int, part: int, parts: int, align: int = 64) -> Tuple[int, int]:
    """Sub-range `part` of `parts` of the column range [lo, hi): boundaries at multiples of `align` elements from
    lo (so that every chunk of a 16-byte-per-lane kernel stays aligned), the last part takes the remainder."""
    if parts <= 1:
        return lo, hi
    step = -(-(hi - lo) // parts)
    step = -(-step // align) * align
    a = min(hi, lo + part * step)
    b = hi if part == parts - 1 else min(hi, a + step)
    return a, b


def _exchange_start(local_rows, row_ids, n, group, template, recv, part, parts):
    """Issue the point-to-point operations of one (part of an) exchange; returns (slices, requests).  The slices
    that arrive from peers are valid after _exchange_finish(requests)."""
    world = _world(group)
    rank = _rank(group)
    lo, hi = part_range(*shard_range(n, rank, world), part, parts)
    any_row = next(iter(local_rows.values())) if local_rows else template
    out: List[Optional[torch.Tensor]] = [None] * len(row_ids)
    ops = []
    for idx, j in enumerate(row_ids):
        owner = row_owner(j, world)
        if owner == rank:
            row = local_rows[j]
            out[idx] = row[lo:hi]                     # the owner's slice is a view: nothing moves
            for peer in range(world):
                if peer == rank:
                    continue
                plo, phi = part_range(*shard_range(n, peer, world), part, parts)
                if phi > plo:
                    ops.append(dist.P2POp(dist.isend, row[plo:phi], peer, group=group, tag=idx))
        elif hi > lo:
            if any_row is None:
                raise ValueError('a rank that owns no row must pass `template`')
            buf = recv[idx][:hi - lo] if recv is not None else \
                torch.empty((hi - lo,) + tuple(any_row.shape[1:]), dtype=any_row.dtype, device=any_row.device)
            out[idx] = buf
            ops.append(dist.P2POp(dist.irecv, buf, owner, group=group, tag=idx))
    for idx, t in enumerate(out):
        if t is None:       # empty shard
            out[idx] = torch.empty((0,) + tuple(any_row.shape[1:]), dtype=any_row.dtype, device=any_row.device)
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return out, reqs


def _exchange_finish(reqs):
    for req in reqs:
        req.wait()          # RCCL: the CURRENT STREAM waits for the transfer (the host does not block); gloo: host wait


def exchange_party_major(local_rows: Dict[int, torch.Tensor], row_ids: Sequence[int], n: int,
                         group: Optional[dist.ProcessGroup] = None,
                         template: Optional[torch.Tensor] = None,
                         recv: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """local_rows: {row id: full-length limb tensor (n, ...)} for the rows this rank owns
    (row_owner).  Returns, for every row in row_ids (in that order), this rank's column slice
    [lo, hi) -- ready to be passed to FieldContext.recombine.  `template`: any tensor with the rows' dtype,
    device and trailing (limb) shape, for a rank that owns no row.  `recv`: optional preallocated receive
    buffers, one per entry of row_ids (used for the rows that arrive from peers)."""
    _check_rows(local_rows, n)
    any_row = next(iter(local_rows.values())) if local_rows else template
    staged = _host_staged(group) and any_row is not None and any_row.is_cuda
    if staged:
        dev = any_row.device
        got = exchange_party_major({j: r.cpu() for j, r in local_rows.items()}, row_ids, n, group, any_row[:0].cpu())
        return [g.to(dev) for g in got]
    out, reqs = _exchange_start(local_rows, row_ids, n, group, template, recv, 0, 1)
    _exchange_finish(reqs)
    return out


def allgather_rows(local_row: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """All-gather of one full row per rank (rank r contributes row r): every rank ends up with all
    `world` rows.  RCCL all-gather over xGMI when the tensors are on the GPU."""
    world = _world(group)
    if world == 1:
        return [local_row]
    if _host_staged(group) and local_row.is_cuda:
        return [r.to(local_row.device) for r in allgather_rows(local_row.cpu(), group)]
    rows = [torch.empty_like(local_row) for _ in range(world)]
    dist.all_gather(rows, local_row.contiguous(), group=group)
    return rows


class PartyMajorGather:
    """The all-gather form of the exchange (the north star's wording: "RCCL all-gather over xGMI only for the
    recombine step"): every rank contributes the block of rows it owns, padded to ceil(k / world) rows, in ONE
    all_gather_into_tensor; afterwards every rank holds every whole row and recombines the column range it
    wants (its own shard, or everything).  Moves world x the bytes of exchange_party_major; offered because
    it is one collective on a preallocated buffer (graph-friendly, no per-peer bookkeeping).

    The block and the gathered buffer are allocated once and reused (`block_row(j)` is where the owner of row j
    writes it, e.g. as the `out` of share generation; `row(j)` is where every rank reads it after gather())."""

    def __init__(self, k: int, n: int, template: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
        self.group, self.k, self.n = group, k, n
        self.world, self.rank = _world(group), _rank(group)
        self.rows_per_rank = -(-k // self.world)
        tail = tuple(template.shape[1:])
        self.block = torch.zeros((self.rows_per_rank, n) + tail, dtype=template.dtype, device=template.device)
        self.all = torch.empty((self.world * self.rows_per_rank, n) + tail, dtype=template.dtype, device=template.device)

    def block_row(self, j: int) -> torch.Tensor:
        if row_owner(j, self.world) != self.rank:
            raise ValueError(f'row {j} lives on rank {row_owner(j, self.world)}')
        return self.block[j // self.world]

    def gather(self):
        if self.world == 1:
            self.all.copy_(self.block)
            return
        if _host_staged(self.group) and self.block.is_cuda:
            host = torch.empty(self.all.shape, dtype=self.all.dtype)
            dist.all_gather_into_tensor(host, self.block.cpu(), group=self.group)
            self.all.copy_(host)
        else:
            dist.all_gather_into_tensor(self.all, self.block, group=self.group)

    def row(self, j: int) -> torch.Tensor:
        return self.all[row_owner(j, self.world) * self.rows_per_rank + j // self.world]

    @property
    def bytes_received(self) -> int:
        """bytes this rank receives from its peers per gather()"""
        return (self.world - 1) * self.block.numel() * self.block.element_size()


def scatter_party_major(slices: Sequence[torch.Tensor], row_ids: Sequence[int], n: int,
                        out_rows: Dict[int, torch.Tensor], group: Optional[dist.ProcessGroup] = None):
    """Inverse of exchange_party_major: every rank holds its column range [lo, hi) of all rows in row_ids
    (e.g. the share rows it has just generated for its shard of secrets) and row j must end up whole on
    rank j % world.  out_rows: {row id: full-length tensor} preallocated on the owner."""
    world = _world(group)
    rank = _rank(group)
    lo, hi = shard_range(n, rank, world)
    if _host_staged(group) and slices and slices[0].is_cuda:
        host_out = {j: torch.empty(r.shape, dtype=r.dtype) for j, r in out_rows.items()}
        scatter_party_major([t.cpu() for t in slices], row_ids, n, host_out, group)
        for j, r in out_rows.items():
            r.copy_(host_out[j])
        return
    ops = []
    for idx, j in enumerate(row_ids):
        owner = row_owner(j, world)
        if owner == rank:
            out_rows[j][lo:hi].copy_(slices[idx])
            for peer in range(world):
                plo, phi = shard_range(n, peer, world)
                if peer != rank and phi > plo:
                    ops.append(dist.P2POp(dist.irecv, out_rows[j][plo:phi], peer, group=group, tag=idx))
        elif hi > lo:
            ops.append(dist.P2POp(dist.isend, slices[idx].contiguous(), owner, group=group, tag=idx))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def recombine_party_major(ctx, local_rows: Dict[int, torch.Tensor], row_ids: Sequence[int],
                          lambdas: Sequence[int], n: int, group: Optional[dist.ProcessGroup] = None,
                          template: Optional[torch.Tensor] = None, chunks: int = 1, out=None,
                          recv: Optional[Sequence[torch.Tensor]] = None):
    """Exchange + local Lagrange recombination: returns this rank's slice of the secrets as a
    DevArray (element-sharded result).

    chunks > 1 PIPELINES the two (SURVEY 8e prices the party-major step as communication-bound at N = 8): the
    rank's column range is cut into `chunks` parts; the transfers of part c+1 are issued before the recombination
    kernel of part c is launched, so the kernel runs while the next slices are in flight on the RCCL stream
    (point-to-point transfers and kernels use different engines / streams).  Only the first transfer and the last
    kernel are exposed.  Results are bit-identical to chunks = 1 (the kernel is element-wise)."""
    from .engine import DevArray
    if len(row_ids) != len(lambdas):
        raise ValueError('one Lagrange coefficient per row')
    _check_rows(local_rows, n)
    world, rank = _world(group), _rank(group)
    lo, hi = shard_range(n, rank, world)
    any_row = next(iter(local_rows.values())) if local_rows else template
    if chunks <= 1 or (_host_staged(group) and any_row is not None and any_row.is_cuda):
        slices = exchange_party_major(local_rows, row_ids, n, group, template, recv)
        rows = [DevArray(ctx, t, t.shape[0]) for t in slices]
        if rows[0].n == 0:
            return ctx.empty(0)
        return ctx.recombine(rows, lambdas, out=out)
    out = out if out is not None else ctx.empty(hi - lo)
    if hi == lo:
        return out
    if recv is None:
        # two sets of receive buffers of one part each: part c+1 lands while part c is being read
        plen = max(part_range(lo, hi, c, chunks)[1] - part_range(lo, hi, c, chunks)[0] for c in range(chunks))
        bufs = [[torch.empty((plen,) + tuple(any_row.shape[1:]), dtype=any_row.dtype, device=any_row.device)
                 for _ in row_ids] for _ in range(2)]
    pending = None
    for c in range(chunks + 1):
        nxt = None
        if c < chunks:
            a, b = part_range(lo, hi, c, chunks)
            rbufs = [r[a - lo:b - lo] for r in recv] if recv is not None else bufs[c % 2]
            nxt = (_exchange_start(local_rows, row_ids, n, group, template, rbufs, c, chunks), a, b)
        if pending is not None:
            (slices, reqs), a, b = pending
            _exchange_finish(reqs)
            if b > a:
                ctx.recombine([DevArray(ctx, t[:b - a], b - a) for t in slices], lambdas,
                              out=DevArray(ctx, out.t[a - lo:b - lo], b - a))
        pending = nxt
    return out
