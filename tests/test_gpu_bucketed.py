"""The bucketed batch path (bv_sorted.hip) forced on vectors and batches of every shape: its answers are the direct kernel's.
Covers what the BASELINE-size tests cannot: few lines (one digit only, empty digit-1 groups, slices without keys), batches
smaller than a tile / than the number of blocks, all keys in one slice, sorted and windowed batches, arguments out of range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def words(n, d, seed):
    rng = np.random.default_rng(seed)
    bits = rng.random(n) < d
    pad = (-n) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)])
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


def batches(n_bits, rng, big):
    yield "one", np.array([n_bits // 2], dtype=np.uint64)
    yield "few", rng.integers(0, n_bits + 1, 100, dtype=np.uint64)
    for m in (8191, 8192, 8193, big):
        yield f"uniform{m}", rng.integers(0, n_bits + 1, m, dtype=np.uint64)
    yield "same", np.full(50_000, n_bits // 3, dtype=np.uint64)
    yield "sorted", np.sort(rng.integers(0, n_bits + 1, 200_000, dtype=np.uint64))
    yield "window", (n_bits // 2 + rng.integers(0, min(n_bits // 2, 1 << 12) + 1, 100_000, dtype=np.uint64))
    bad = rng.integers(0, n_bits + 1, 100_000, dtype=np.uint64)
    bad[::3] = n_bits + 1 + (bad[::3] << np.uint64(20))
    bad[5] = np.uint64(2 ** 64 - 1)
    yield "out_of_range", bad
    yield "ends", np.concatenate([np.arange(min(n_bits + 1, 3000), dtype=np.uint64), np.arange(max(0, n_bits - 3000), n_bits + 1, dtype=np.uint64)])


@pytest.mark.parametrize("n_bits", [1, 447, 448, 449, 448 * 1024, 448 * 1024 + 1, 448 * 1024 * 3 + 5, 448 * 1024 * 300, 40_000_003])
def test_bucketed_rank_equals_direct(gpu, n_bits):
    w = words(n_bits, 0.5 if n_bits % 2 else 0.07, n_bits % 997)
    bv = gpu.bit_vector(w, n_bits)
    rng = np.random.default_rng(n_bits)
    for name, idx in batches(n_bits, rng, 1_500_000):
        for bit in (0, 1):
            gpu.set_option("rank_sorted", 0)
            want = bv.rank(idx, bit)
            try:
                gpu.set_option("rank_sorted", 1)
                gpu.set_option("trace_phases", 1)
                got = bv.rank(idx, bit)
                assert gpu.last_phases().get("select") == 0, "the bucketed path was not taken"
            finally:
                gpu.set_option("rank_sorted", -1)
                gpu.set_option("trace_phases", 0)
            assert np.array_equal(got, want), f"{name}, bit {bit}"
    bv.release_scratch()


@pytest.mark.parametrize("n_bits,d", [(449, 0.5), (448 * 1024 + 1, 0.5), (448 * 1024 * 40, 0.5), (448 * 1024 * 40, 0.01), (40_000_003, 0.93)])
def test_bucketed_select_equals_direct(gpu, n_bits, d):
    w = words(n_bits, d, n_bits % 991)
    bv = gpu.bit_vector(w, n_bits)
    rng = np.random.default_rng(n_bits + 1)
    for bit in (0, 1):
        total = bv.ones() if bit else n_bits - bv.ones()
        if total < 2:
            continue
        cases = {"few": rng.integers(1, total + 1, 100, dtype=np.uint64),
                 "uniform": rng.integers(1, total + 1, 1_000_000, dtype=np.uint64),
                 "tile": rng.integers(1, total + 1, 8193, dtype=np.uint64),
                 "same": np.full(30_000, max(1, total // 2), dtype=np.uint64),
                 "sorted": np.sort(rng.integers(1, total + 1, 200_000, dtype=np.uint64)),
                 "out_of_range": np.concatenate([rng.integers(0, total + 3, 100_000, dtype=np.uint64), np.array([0, total, total + 1, 2 ** 64 - 1], dtype=np.uint64)])}
        for name, i in cases.items():
            gpu.set_option("select_sorted", 0)
            want = bv.select(i, bit)
            try:
                gpu.set_option("select_sorted", 1)
                got = bv.select(i, bit)
            finally:
                gpu.set_option("select_sorted", -1)
            assert np.array_equal(got, want), f"{name}, bit {bit}"
    bv.release_scratch()


def test_histogram_pass_2_still_works(gpu):
    """SDSL_HIP_SORTED_SWEEP=0 keeps the second histogram pass instead of the one-sweep partition (A/B measurements): the
    path is read from the environment when the library first runs a bucketed batch, so it is exercised in a child process."""
    import os, subprocess, sys
    code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("sdsl-lite_amd")
rng = np.random.default_rng(1)
n = 448 * 1024 * 37 + 11
w = rng.integers(0, 2**63, (n + 63) // 64, dtype=np.int64).view(np.uint64)
bv = pkg.bit_vector(w, n)
idx = rng.integers(0, n + 1, 700_001, dtype=np.uint64)
i1 = rng.integers(1, bv.ones() + 1, 700_001, dtype=np.uint64)
pkg.set_option("rank_sorted", 0); pkg.set_option("select_sorted", 0)
want = bv.rank(idx, 1), bv.rank(idx, 0), bv.select(i1, 1)
pkg.set_option("rank_sorted", 1); pkg.set_option("select_sorted", 1); pkg.set_option("trace_phases", 1)
got = bv.rank(idx, 1), bv.rank(idx, 0), bv.select(i1, 1)
ph = pkg.last_phases()
assert ph.get("hist2", 0) > 0.0005, ph
assert all(np.array_equal(a, b) for a, b in zip(got, want))
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDSL_HIP_SORTED_SWEEP="0", SDSL_HIP_SORTED_SWC="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_vectors_and_batches_for_ten_seconds(gpu):
    """tools/stress_bucketed.py: random vector sizes / densities / batch sizes / distributions, bucketed against direct."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_bucketed.py"), "10", "3"], capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("window_log2", [8, 20])
def test_heavily_skewed_large_batch(gpu, window_log2):
    """6 * 10^7 positions inside one line / inside three slices of a 2^30-bit vector, forced through the passes: every
    histogram block then sends > 2 * 10^5 keys to one 16-bit slice counter, which must hand them on to its global row without
    ever overflowing (the first form of that hand-over lost 65536 keys of a slice on exactly such a batch)."""
    import torch
    n_bits = (1 << 30) - 37
    g = torch.Generator(device="cuda").manual_seed(11)
    words = torch.randint(-2**63, 2**63 - 1, ((n_bits + 63) // 64,), device="cuda", dtype=torch.int64, generator=g)
    bv = gpu.bit_vector(words, n_bits)
    del words
    idx = (n_bits // 3) + torch.randint(0, 1 << window_log2, (60_000_000,), device="cuda", dtype=torch.int64, generator=g)
    try:
        gpu.set_option("rank_sorted", 0)
        want = bv.rank(idx, 1).clone()
        gpu.set_option("rank_sorted", 1)
        got = bv.rank(idx, 1)
        assert torch.equal(got, want)
    finally:
        gpu.set_option("rank_sorted", -1)
        bv.release_scratch()


@pytest.mark.parametrize("n_bits,d", [(1, 1.0), (2141, 0.3), (2142, 0.05), (2143, 0.5), (2142 * 128, 0.05), (2142 * 128 + 1, 0.95), (2142 * 128 * 3 + 5, 0.5),
                                      (40_000_003, 0.05), (2142 * 128 * 300, 0.02)])
def test_bucketed_rrr_rank_equals_direct(gpu, n_bits, d):
    """rank on rrr_vector<63> through the passes and the slice-wise decoder (rrr_sorted.hip) against the direct kernel: vectors
    that end inside / at the end of a record and of a slice, every density class (sparse decoder, complement, raw classes),
    batches of every shape."""
    w = words(n_bits, d, n_bits % 983)
    rv = gpu.rrr_vector(w, n_bits)
    rng = np.random.default_rng(n_bits + 3)
    for name, idx in batches(n_bits, rng, 1_200_000):
        for bit in (0, 1):
            gpu.set_option("rrr_sorted", 0)
            want = rv.rank(idx, bit)
            try:
                gpu.set_option("rrr_sorted", 1)
                got = rv.rank(idx, bit)
            finally:
                gpu.set_option("rrr_sorted", -1)
            assert np.array_equal(got, want), f"{name}, bit {bit}"


@pytest.mark.parametrize("n_bits,d", [(2143, 0.5), (2142 * 128 + 1, 0.05), (2142 * 128 * 5 + 77, 0.5), (40_000_003, 0.05), (40_000_003, 0.97),
                                      (2142 * 128 * 300, 0.003)])
def test_bucketed_rrr_select_equals_direct(gpu, n_bits, d):
    """select_1 / select_0 on rrr_vector<63> through the passes (buckets of argument ranks, records decoded slice-wise) against
    the direct kernel; sparse vectors put buckets wider than a slice into the fix-up pass."""
    w = words(n_bits, d, n_bits % 977)
    rv = gpu.rrr_vector(w, n_bits)
    rng = np.random.default_rng(n_bits + 5)
    for bit in (0, 1):
        total = rv.ones() if bit else n_bits - rv.ones()
        if total < 2:
            continue
        cases = {"few": rng.integers(1, total + 1, 100, dtype=np.uint64),
                 "uniform": rng.integers(1, total + 1, 700_000, dtype=np.uint64),
                 "same": np.full(30_000, max(1, total // 2), dtype=np.uint64),
                 "sorted": np.sort(rng.integers(1, total + 1, 150_000, dtype=np.uint64)),
                 "out_of_range": np.concatenate([rng.integers(0, total + 3, 50_000, dtype=np.uint64), np.array([0, total, total + 1, 2 ** 64 - 1], dtype=np.uint64)])}
        for name, i in cases.items():
            gpu.set_option("rrr_sorted", 0)
            want = rv.select(i, bit)
            try:
                gpu.set_option("rrr_sorted", 1)
                got = rv.select(i, bit)
            finally:
                gpu.set_option("rrr_sorted", -1)
            assert np.array_equal(got, want), f"{name}, bit {bit}"
