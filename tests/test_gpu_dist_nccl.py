"""GPU: the torch.distributed path of the multi-GPU driver (dist.py) on the backend the product uses — "nccl", which
is RCCL on ROCm — with the ranks the test box has (one): the load-time broadcast (dist.replicate) and the root-owned batch
(dist.sharded_query, single piece and four pipelined pieces: scatter -> kernels -> gather with asynchronous collectives on
RCCL's streams) run through the real communicator, so that branch is exercised on hardware before an 8-GPU node sees it."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    pkg = importlib.import_module("sdsl-lite_amd")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ["LOCAL_RANK"]) %% torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    n_bits = (1 << 24) + 13
    words = torch.from_numpy(pkg.set_random_bits(n_bits, 3).view(np.int64)).to(dev) if rank == 0 else None
    words = pkg.dist.replicate(words, torch.empty(0, dtype=torch.int64, device=dev))          # ncclBroadcast
    bv = pkg.bit_vector(words, n_bits, device=local)
    n = 3_000_017
    idx = torch.from_numpy(pkg.rnd_positions(5, n, n_bits + 1, 0).view(np.int64)).to(dev) if rank == 0 \\
        else torch.empty(1, dtype=torch.int64, device=dev)
    one = pkg.dist.sharded_query(lambda x: bv.rank(x, 1), (idx,), n)                                # scatter + gather
    four = pkg.dist.sharded_query(lambda x: bv.rank(x, 1), (idx,), n, chunks=4)                     # pipelined pieces
    t = pkg.dist.max_over_ranks(1.5 + rank, dev)
    if rank == 0:
        want = bv.rank(idx, 1)
        assert torch.equal(one, want) and torch.equal(four, want)
        assert t == 1.5 + world - 1
        print("NCCL_PATH_OK", world, int(want.sum()))
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()
''') % ROOT


def test_replicate_and_sharded_query_over_rccl(gpu, tmp_path):
    import torch
    script = tmp_path / "nccl_path.py"
    script.write_text(SCRIPT)
    ranks = max(1, torch.cuda.device_count())
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "NCCL_PATH_OK" in r.stdout
