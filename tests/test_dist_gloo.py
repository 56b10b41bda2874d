"""CPU, world_size 2, gloo: the multi-GPU query sharding (sdsl-lite_amd/dist.py) — scatter of a
root-owned batch, local answering, gather at the same offsets.  The local engine is stood in for by the
CPU oracle (test infrastructure) because no GPU exists here; on the GPU box the same code path runs with
backend nccl (= RCCL) and the HIP engine."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_q, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("sdsl-lite_amd")
    import oracle_lib as ol
    import golden_data as gd
    # every rank holds a replica of the index
    n = 100_003
    w = ol.set_random_bits(n, 5)
    bv = ol.OBitVector(w, n)
    csa = ol.OCsa(gd.text("example01.txt") * 3)

    def rank_fn(idx):
        return torch.from_numpy(bv.rank(idx.numpy().astype(np.uint64)).astype(np.int64))

    m = 4

    def count_fn(p):
        return torch.from_numpy(csa.count_batch(p.numpy(), m).astype(np.int64))

    if rank == 0:
        idx = torch.from_numpy((ol.mt19937_64(n_q, 3) % np.uint64(n + 1)).astype(np.int64))
        t = np.frombuffer(gd.text("example01.txt") * 3, dtype=np.uint8)
        st = ol.mt19937_64(n_q, 4) % np.uint64(t.size - m)
        pats = torch.from_numpy(np.concatenate([t[int(s):int(s) + m] for s in st]))
    else:
        idx = torch.zeros(0, dtype=torch.int64)
        pats = torch.zeros(0, dtype=torch.uint8)
    r = pkg.dist.sharded_query(rank_fn, (idx,), n_q)
    c = pkg.dist.sharded_query(count_fn, (pats,), n_q, widths=(m,))
    # pipelined: the batch in three pieces, scatter / answer / gather of neighbouring pieces overlap
    r3 = pkg.dist.sharded_query(rank_fn, (idx,), n_q, chunks=3)
    c3 = pkg.dist.sharded_query(count_fn, (pats,), n_q, widths=(m,), chunks=3)
    # load-time replication of an index input: rank 0 owns the words, every rank ends up with the same tensor
    words = torch.from_numpy(w.astype(np.int64)) if rank == 0 else None
    got = pkg.dist.replicate(words, torch.zeros(0, dtype=torch.int64))
    assert np.array_equal(got.numpy().astype(np.uint64), w)
    lo, hi = pkg.dist.shard_bounds(n_q, world, rank)
    slow = pkg.dist.max_over_ranks(float(rank + 1), "cpu")
    assert slow == float(world)
    if rank == 0:
        ok_r = np.array_equal(r.numpy().astype(np.uint64), bv.rank(idx.numpy().astype(np.uint64))) and torch.equal(r, r3)
        ok_c = np.array_equal(c.numpy().astype(np.uint64), csa.count_batch(pats.numpy(), m)) and torch.equal(c, c3)
        open(out_path, "w").write(f"{int(ok_r)}{int(ok_c)} {lo} {hi}")
    else:
        assert r is None and c is None and r3 is None and c3 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_q", [1000, 1001, 1])
def test_sharded_query_two_ranks(tmp_path, n_q):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), n_q, str(out)), nprocs=2, join=True)
    flags, lo, hi = out.read_text().split()
    assert flags == "11"
    assert (int(lo), int(hi)) == (0, (n_q + 1) // 2)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    d = importlib.import_module("sdsl-lite_amd").dist
    for n in (0, 1, 7, 8, 9, 10**9):
        for ws in (1, 2, 3, 8):
            spans = [d.shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
