"""GPU: bench.py honours its contract on a reduced workload — one JSON line with the required keys, the
roofline and cpu_baseline objects, and a multi-rank run (2 processes sharing the one GPU of the test box,
gloo for the control-plane collectives) that aggregates over ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _last_json(out):
    """the contract: ONE JSON line, the LAST line of stdout, compact (bench.py: LINE_LIMIT)"""
    all_lines = [l for l in out.splitlines() if l.strip()]
    lines = [l for l in all_lines if l.startswith("{")]
    assert len(lines) == 1 and all_lines[-1] == lines[0], out[-3000:]
    assert len(lines[0].encode()) < 4096, len(lines[0])
    return json.loads(lines[0])


def _extras(d):
    """the legs' blocks: the sidecar the line points to"""
    assert d["extras_file"], d
    side = json.load(open(os.path.join(ROOT, d["extras_file"])))
    return side["extras"]


def test_single_gpu_contract(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log-n", "26",
                        "--queries", "4e6", "--cpu-seconds", "0.5", "--extras", "select"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert all(k in d for k in REQUIRED)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["higher_is_better"] and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 4e6 * 3 / (d["ms_per_step"] * 3e-3) / 1e9) < 1e-6 * d["value"] + 1e-9
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] in ("reference", "port") and cb["matches_gpu"] is True
    assert rf["traffic"] is None or "committed PMC" in rf["traffic_source"]
    ex = _extras(d)
    assert "select_1" in ex and "select_1" in d["summary"] and d["extras_error"] is None
    assert "headline" in ex and ex["headline"]["cpu_baseline_all_cores"]["matches_gpu"] is True and ex["headline"]["roofline"]["phases_ms"] is None


def test_two_ranks_share_the_gpu(gpu):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "24", "--queries", "1e6",
                        "--extras", "none", "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 1e6 * 2 / (d["ms_per_step"] * 2e-3) / 1e9) < 1e-6 * d["value"] + 1e-9


def test_two_ranks_sharded_fm_count(gpu):
    """configs[4] path of the multi-GPU bench (default extras of an N > 1 run): replicated FM-index, patterns sharded
    over the ranks — resident shards and a root-owned batch through dist.sharded_query"""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "24", "--queries", "2e6",
                        "--text-mib", "16", "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    ex = _extras(d)
    fs = ex["fm_count_sharded"]
    assert "error" not in ex and d["extras_error"] is None, ex
    assert d["secondary"]["n_gpus"] == 2 and d["secondary"]["value"] == pytest.approx(fs["resident_shards"]["Mcount/s"], rel=1e-3)
    assert d["scaling_columns"]["end_to_end_root_owned_batch_Grank/s"] > 0
    assert fs["patterns_total"] == 200000 and fs["resident_shards"]["all_patterns_found"] is True
    assert fs["root_owned_batch_matches"] is True
    assert fs["resident_shards"]["Mcount/s"] > 0 and fs["root_owned_batch"]["Mcount/s"] > 0
    assert fs["root_owned_batch_pipelined"]["matches"] is True and fs["text_broadcast_s"] >= 0
    ro = ex["rank_root_owned_batch"]
    assert ro["queries"] == 2 * 2000000 and ro["matches_local"] is True and ro["Grank/s"] > 0



def test_device_generator_equals_the_host_generator(gpu):
    """bench.py draws its vector and its positions on the device from generator checkpoints (workload_dev.hip): the same
    numbers as the host's mt19937_64 walk, for every way a stretch can start and end"""
    import numpy as np
    for seed, count, mod, add, stride in ((42, 100_003, 0, 0, 1 << 10), (7, 1_000_000, (1 << 34) + 1, 0, 312 * 64), (11, 5, 1000, 1, 1 << 20),
                                          (9, 312 * 5, 100, 0, 312), (13, 70_001, (1 << 30) + 2, 3, 77_777), (5, 313, 0, 9, 100)):
        want = gpu.rnd_positions(seed, count, mod, add)
        got = gpu.rnd_positions_device(seed, count, mod, add, 0, stride).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), (seed, count, mod, add, stride)


def test_eight_ranks_on_one_gpu(gpu):
    """the launch the driver makes on an 8-GPU node, with all eight ranks sharing the test box's GPU (gloo for the control
    plane): eight replicas of the 2^34-bit index (2.45 GB each), every rank's positions drawn on the device, the root-owned batch
    and the sharded count() of configs[4] — inside five minutes"""
    import time
    env = dict(os.environ, OMP_NUM_THREADS="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--log-n", "34",
                        "--queries", "2e7", "--text-mib", "64", "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=env)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["reference_digest_match"] is True
    ex = _extras(d)
    assert "error" not in ex, ex
    fs = ex["fm_count_sharded"]
    assert fs["resident_shards"]["all_patterns_found"] is True and fs["root_owned_batch_matches"] is True
    assert ex["rank_root_owned_batch"]["matches_local"] is True
    assert took < 300, f"{took:.0f} s"


def test_goldens_and_bench_on_a_text_file_of_the_users(gpu, tmp_path):
    """The recipe for a maintainer who holds Pizza&Chili's english.1GB (not available offline): `make_golden_large.py c4 c4sel c4s
    --text-file FILE` makes the real library's digests for THAT text, `bench.py --text-file FILE` finds them and reports
    reference_digest_match for wt.rank / wt.select / count on it.  Here at 1 MiB with a text that holds zero bytes (they are dropped)."""
    import numpy as np
    import oracle_lib as ol
    if not ol.have_ref():
        pytest.skip("oracle/_ref (the real sdsl-lite) did not travel with the repo")
    text = gpu.english_text(1 << 20, 77).copy()
    text[::5000] = 0
    f = tmp_path / "mytext.1MB"
    text.tofile(str(f))
    gold = os.path.join(ROOT, "tests", "golden", "golden_large_mytext.1MB.json")
    env = dict(os.environ, GOLDEN_TEXT_LOG="20", GOLDEN_NQ_TEXT="20000", GOLDEN_NQ_WTSEL="20000", GOLDEN_NQ_TEXT_FULL="400000", GOLDEN_STRIDE="100")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden_large.py"), "c4", "c4sel", "c4s", "--text-file", str(f)],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and os.path.exists(gold), r.stdout[-2000:] + r.stderr[-2000:]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "24", "--queries", "4e5",
                            "--no-cpu", "--extras", "wt,fm", "--text-file", str(f), "--text-mib", "1"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _last_json(r.stdout)
        ex = _extras(d)
        assert d["extras_error"] is None, ex.get("error")
        assert ex["text"]["bytes"] == int((text != 0).sum()) and "mytext.1MB" in ex["text"]["kind"]
        assert ex["wt_huff_rank"]["reference_digest_match"] is True and ex["wt_huff_select"]["reference_digest_match"] is True
        assert ex["fm_count"]["reference_digest_match"] is True and d["secondary"]["reference_digest_match"] is True
        assert ex["fm_count_lean"]["reference_digest_match"] is True
    finally:
        if os.path.exists(gold):
            os.remove(gold)
