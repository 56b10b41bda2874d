"""bench.py --gpus N must never report a device count that was not asked for (with WORLD_SIZE unset it used to run
one process and print n_gpus 1).  Without a launcher it starts its N ranks itself; with too few devices it refuses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_refuses_more_gpus_than_visible():
    """(CPU container: no device at all; GPU box: one device) --gpus 64 over RCCL cannot be served: exit code 2, no JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stderr[-500:]
    assert "device" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--mode", "group", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and not r.stdout.strip()


@pytest.mark.gpu
def test_launches_its_own_ranks(gpu):
    """--gpus 2 --backend gloo without torchrun: two ranks (sharing the GPU of a one-GPU box), one line, n_gpus 2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--log-n", "26", "--queries", "2e6",
                        "--steps", "2", "--warmup", "1", "--extras", "none", "--no-cpu"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0


@pytest.mark.gpu
def test_group_mode_line(gpu):
    """--mode group on whatever devices the box has (one: the group has one member and still goes through RCCL)."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--mode", "group", "--log-n", "28", "--queries", "4e6",
                        "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == n
    c = j["scaling_columns"]
    assert c["root_owned_batch"]["matches_single_gpu"] and c["kernel_only_resident_shards_Grank/s"] > 0
