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


def replicate(t: torch.Tensor | None, like: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """The one collective at load time: rank `src` holds the raw input of an index (bit-vector words, a text);
    every rank receives a copy and lays out its own replica of the device structure from it.
    `like` is a tensor of the right dtype/device on every rank (its content is ignored)."""
    rank = dist.get_rank(group)
    n = torch.tensor([t.numel() if rank == src else 0], dtype=torch.int64, device=like.device)
    dist.broadcast(n, src=src, group=group)
    buf = t.contiguous().view(-1) if rank == src else torch.empty(int(n.item()), dtype=like.dtype, device=like.device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def sharded_query(fn: Callable[..., torch.Tensor], columns: Sequence[torch.Tensor | None], n: int,
                  widths: Sequence[int] | None = None, src: int = 0, group=None, chunks: int = 1) -> torch.Tensor | None:
    """Answer a batch of `n` queries owned by rank `src` on all ranks of `group`.

    columns : the query columns on rank `src` (e.g. (idx,) for rank, (i, c) for wt.rank, (patterns,)
              for count); other ranks pass tensors of the right dtype/device and any length (used as
              dtype/device templates) — their content is ignored.
    widths  : elements per query in each column (1 for idx, m for m-byte patterns).
    fn      : local engine call, fn(*shard_columns) -> int64 tensor with one result per query.
    chunks  : > 1 cuts the batch into that many pieces and pipelines them: while piece c is being answered the
              scatter of piece c+1 and the gather of piece c-1 are in flight (asynchronous collectives on the
              communicator's own stream), so transfers over xGMI overlap the kernels.
    Returns the gathered results on `src` (None elsewhere).
    """
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    widths = list(widths) if widths is not None else [1] * len(columns)
    chunks = max(1, min(int(chunks), max(1, n)))
    bounds = [shard_bounds(n, chunks, c) for c in range(chunks)]

    def post_scatter(c):
        c0, c1 = bounds[c]
        nc = c1 - c0
        per = (nc + ws - 1) // ws
        recvs, works = [], []
        for col, w in zip(columns, widths):
            recv = torch.empty(per * w, dtype=col.dtype, device=col.device)
            piece = col[c0 * w: c1 * w] if rank == src else None
            works.append(dist.scatter(recv, _equal_chunks(piece, ws, w) if rank == src else None, src=src, group=group,
                                      async_op=True))
            recvs.append(recv)
        return recvs, works

    full = None
    pending = []  # (work, outs, chunk) of gathers in flight
    nxt = post_scatter(0)
    for c in range(chunks):
        recvs, works = nxt
        for wk in works:
            wk.wait()
        if c + 1 < chunks:
            nxt = post_scatter(c + 1)
        c0, c1 = bounds[c]
        nc = c1 - c0
        per = (nc + ws - 1) // ws
        lo, hi = shard_bounds(nc, ws, rank)
        res = fn(*[r[: (hi - lo) * w] for r, w in zip(recvs, widths)])
        padded = torch.zeros(per, dtype=res.dtype, device=res.device)
        padded[: hi - lo] = res
        outs = [torch.empty_like(padded) for _ in range(ws)] if rank == src else None
        pending.append((dist.gather(padded, outs, dst=src, group=group, async_op=True), outs, c, padded))
        if rank == src and full is None:
            full = torch.empty(n, dtype=res.dtype, device=res.device)
    for wk, outs, c, _keep in pending:
        wk.wait()
        if rank == src:
            c0, c1 = bounds[c]
            for r in range(ws):
                a, b = shard_bounds(c1 - c0, ws, r)
                full[c0 + a: c0 + b] = outs[r][: b - a]
    return full if rank == src else None


def max_over_ranks(value: float, device, group=None) -> float:
    """The slowest rank defines a step's duration."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
