"""select_0 on sd_vector<> with the zero directory (sd.hip: quad_sd_select0's fast path) against numpy's list of zero positions and
against the same vector built without the directory: uniform sparse vectors (the fast path confirms its bucket), clustered ones
(more than a bucket's worth of entries in front of the answer: the general search takes over from exact counts), dense and
periodic ones, zeros at both ends."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = (1 << 24) + 12345


def make(name):
    rng = np.random.default_rng(5)
    bits = np.zeros(N, dtype=bool)
    if name == "sparse":
        bits[rng.integers(0, N, N // 4000)] = True
    elif name == "medium":
        bits[rng.integers(0, N, N // 40)] = True
    elif name == "dense":
        bits[:] = rng.random(N) < 0.5
    elif name == "clustered":  # runs of ones far longer than a bucket, separated by long gaps
        for s in rng.integers(0, N - 70_000, 40):
            bits[s:s + rng.integers(1, 60_000)] = True
    elif name == "clustered_sparse":  # a sparse vector with a few dense clumps: the buckets near a clump hold thousands of entries
        bits[rng.integers(0, N, N // 3000)] = True
        for s in rng.integers(0, N - 5000, 25):
            bits[s:s + 3000] = rng.random(3000) < 0.9
    elif name == "periodic":
        bits[::257] = True
    elif name == "ends":
        bits[1:N // 2] = True
    return bits


def pack(bits):
    pad = (-bits.size) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)])
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


@pytest.mark.parametrize("name", ["sparse", "medium", "dense", "clustered", "clustered_sparse", "periodic", "ends"])
def test_select0_with_the_zero_directory(gpu, name):
    bits = make(name)
    zeros = np.flatnonzero(~bits).astype(np.uint64)
    assert zeros.size >= 1 << 20
    w = pack(bits)
    with_dir = gpu.sd_vector(w, N)
    os.environ["SDSL_HIP_SD_NO_SEL0_DIR"] = "1"
    try:
        without = gpu.sd_vector(w, N)
    finally:
        del os.environ["SDSL_HIP_SD_NO_SEL0_DIR"]
    assert with_dir.device_bytes() > without.device_bytes()          # the directory is there ...
    assert with_dir.device_bytes() <= without.device_bytes() * 17 // 16 + 64  # ... within its budget
    rng = np.random.default_rng(6)
    i = np.concatenate([rng.integers(1, zeros.size + 1, 400_000, dtype=np.uint64), np.arange(1, 5000, dtype=np.uint64),
                        np.arange(zeros.size - 5000, zeros.size + 1, dtype=np.uint64)])
    want = zeros[(i - np.uint64(1)).astype(np.int64)]
    assert np.array_equal(with_dir.select(i, 0), want), name
    assert np.array_equal(without.select(i, 0), want), name
    # every zero of a stretch that crosses a clump / a run
    j = np.arange(max(1, zeros.size // 3), min(zeros.size, zeros.size // 3 + 600_000) + 1, dtype=np.uint64)
    assert np.array_equal(with_dir.select(j, 0), zeros[(j - np.uint64(1)).astype(np.int64)]), name
    bad = np.array([0, zeros.size + 1, 2 ** 64 - 1], dtype=np.uint64)
    assert np.all(with_dir.select(bad, 0) == np.uint64(2 ** 64 - 1))
    assert with_dir.serialize() == without.serialize()               # SDSL's bytes know nothing of it
