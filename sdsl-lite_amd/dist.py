"""Multi-GPU driver: the index is REPLICATED in every GPU's HBM, a query batch is sharded by rows.

Queries are independent and the index is read-only, so there is no exchange step
inside the hot path.  Two ways to run N ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests):

  * resident shards (bench.py): every rank already holds its slice of the batch in its own HBM and
    answers it — no data-path collective at all (weak scaling);
  * root-owned batch (`sharded_query`): rank `src` owns the whole batch; equal slices are scattered,
    answered locally and gathered back at the same offsets.  One scatter + one gather per batch, each
    peer on its own direct xGMI link; nothing else is communicated.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row ranges: the first n % world_size ranks get one extra row."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _equal_chunks(t: torch.Tensor, world_size: int, width: int) -> list[torch.Tensor]:
    """Pads every shard to the same number of rows (scatter/gather need equal shapes)."""
    n = t.shape[0] // width if width else 0
    per = (n + world_size - 1) // world_size
    chunks = []
    for r in range(world_size):
        lo, hi = shard_bounds(n, world_size, r)
        c = torch.zeros(per * width, dtype=t.dtype, device=t.device)
        c[: (hi - lo) * width] = t[lo * width: hi * width]
        chunks.append(c)
    return chunks


def sharded_query(fn: Callable[..., torch.Tensor], columns: Sequence[torch.Tensor | None], n: int,
                  widths: Sequence[int] | None = None, src: int = 0, group=None) -> torch.Tensor | None:
    """Answer a batch of `n` queries owned by rank `src` on all ranks of `group`.

    columns : the query columns on rank `src` (e.g. (idx,) for rank, (i, c) for wt.rank, (patterns,)
              for count); other ranks pass tensors of the right dtype/device and any length (used as
              dtype/device templates) — their content is ignored.
    widths  : elements per query in each column (1 for idx, m for m-byte patterns).
    fn      : local engine call, fn(*shard_columns) -> int64 tensor with one result per query.
    Returns the gathered results on `src` (None elsewhere).
    """
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    widths = list(widths) if widths is not None else [1] * len(columns)
    per = (n + ws - 1) // ws
    lo, hi = shard_bounds(n, ws, rank)
    local = []
    for col, w in zip(columns, widths):
        recv = torch.empty(per * w, dtype=col.dtype, device=col.device)
        dist.scatter(recv, _equal_chunks(col, ws, w) if rank == src else None, src=src, group=group)
        local.append(recv[: (hi - lo) * w])
    res = fn(*local)
    padded = torch.zeros(per, dtype=res.dtype, device=res.device)
    padded[: hi - lo] = res
    outs = [torch.empty_like(padded) for _ in range(ws)] if rank == src else None
    dist.gather(padded, outs, dst=src, group=group)
    if rank != src:
        return None
    full = torch.empty(n, dtype=res.dtype, device=res.device)
    for r in range(ws):
        a, b = shard_bounds(n, ws, r)
        full[a:b] = outs[r][: b - a]
    return full


def max_over_ranks(value: float, device, group=None) -> float:
    """The slowest rank defines a step's duration."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
