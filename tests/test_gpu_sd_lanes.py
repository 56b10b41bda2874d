"""sd_vector<> through the one-lane-per-query kernels (sd.hip: k_sd_rank_lane, k_sd_select0_lane, k_sd_redo) against numpy and against
the same vector on the quad kernels (SDSL_HIP_SD_NO_LANES): uniform vectors (the lanes answer everything), a uniform vector with a
few dense clumps or long runs (the lanes answer most queries, mark those near a clump, k_sd_redo answers the marked ones; a vector whose
probe batch leaves the short road too often stays on the quad kernels), a dense one, the ends of the universe, arguments outside the precondition.
Reference semantics: rank_support_sd / select_support_sd<0> / operator[] (sd_vector.hpp:328-349,553-575,633-664)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = (1 << 25) + 777


def make(name):
    rng = np.random.default_rng(15)
    bits = np.zeros(N, dtype=bool)
    if name == "sparse":
        bits[rng.integers(0, N, N // 3000)] = True
    elif name == "medium":
        bits[rng.integers(0, N, N // 50)] = True
    elif name == "clumps":  # sparse, with a few stretches whose buckets hold thousands of entries: runs that leave their line
        bits[rng.integers(0, N, N // 2000)] = True
        for s in rng.integers(0, N - 5000, 12):
            bits[s:s + 4000] = rng.random(4000) < 0.9
    elif name == "clustered":
        for s in rng.integers(0, N - 70_000, 60):
            bits[s:s + rng.integers(1, 60_000)] = True
    elif name == "dense":
        bits[:] = rng.random(N) < 0.5
    elif name == "all_ones":
        bits[:] = True
    elif name == "single":
        bits[N // 3] = True
    elif name == "first_half":  # one run: every entry in the buckets of the first half, none in the second
        bits[:N // 2] = True
    elif name == "ends":
        bits[rng.integers(0, N, N // 500)] = True
        bits[:3] = True
        bits[N - 3:] = True
    return bits


def pack(bits):
    pad = (-bits.size) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)])
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


@pytest.mark.parametrize("name,lanes", [("sparse", 3), ("medium", 3), ("clumps", 3), ("clustered", None), ("dense", None), ("ends", 3), ("all_ones", None), ("single", None), ("first_half", None)])
def test_lane_kernels_equal_numpy_and_the_quad_kernels(gpu, name, lanes):
    bits = make(name)
    w = pack(bits)
    v = gpu.sd_vector(w, N)
    os.environ["SDSL_HIP_SD_NO_LANES"] = "1"
    try:
        quad = gpu.sd_vector(w, N)
    finally:
        del os.environ["SDSL_HIP_SD_NO_LANES"]
    assert quad.lane_kernels() == 0
    assert lanes is None or v.lane_kernels() == lanes, "which kernels the probe batch chose"
    print(name, "lane kernels:", v.lane_kernels())
    cum = np.concatenate([[0], np.cumsum(bits, dtype=np.int64)]).astype(np.uint64)
    zeros = np.flatnonzero(~bits).astype(np.uint64)
    ones = np.flatnonzero(bits).astype(np.uint64)
    rng = np.random.default_rng(16)
    near = ones[rng.integers(0, ones.size, 150_000)].astype(np.int64) + rng.integers(-3, 4, 150_000)  # around the entries (and inside the clumps)
    x = np.concatenate([rng.integers(0, N + 1, 400_000), np.clip(near, 0, N), np.arange(0, 3000), np.arange(N - 3000, N + 1)]).astype(np.uint64)
    for bit in (1, 0):
        want = cum[x] if bit else x - cum[x]
        got = np.asarray(v.rank(x, bit))
        assert np.array_equal(got, want), f"rank_{bit}"
        assert np.array_equal(np.asarray(quad.rank(x, bit)), want)
    inside = x[x < N]
    assert np.array_equal(np.asarray(v.access(inside)).astype(bool), bits[inside.astype(np.int64)])
    zi = np.searchsorted(zeros, ones[rng.integers(0, ones.size, 150_000)]).astype(np.int64) + rng.integers(-2, 3, 150_000)  # zeros next to entries
    zs = max(zeros.size, 1)
    i = np.concatenate([rng.integers(1, zs + 1, 400_000), np.clip(zi, 1, zs), np.arange(1, min(3000, zs + 1)),
                        np.arange(max(1, zs - 3000), zs + 1)]).astype(np.uint64)
    if zeros.size:
        got = np.asarray(v.select(i, 0))
        assert np.array_equal(got, zeros[i - np.uint64(1)]), "select_0"
        assert np.array_equal(np.asarray(quad.select(i, 0)), got)
    # outside the precondition: NPOS for rank beyond size() and select_0(0) / beyond the number of zeros; 0xFF for operator[] beyond
    bad = np.array([N + 1, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64)
    assert (np.asarray(v.rank(bad, 1)) == np.uint64(2 ** 64 - 1)).all()
    assert (np.asarray(v.select(np.array([0, zeros.size + 1, 2 ** 64 - 2], dtype=np.uint64), 0)) == np.uint64(2 ** 64 - 1)).all()
    assert (np.asarray(v.access(np.array([N, 2 ** 64 - 1], dtype=np.uint64))) == 0xFF).all()


def test_universe_of_2_pow_40(gpu):
    """positions far beyond 2^32: 2^22 ones in a universe of 2^40, built from the position list"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(2)
    n = 1 << 40
    pos = torch.unique(torch.randint(0, n, (1 << 22,), device="cuda", dtype=torch.int64, generator=g))
    v = gpu.sd_vector(positions=pos, n_bits=n)
    assert v.lane_kernels() == 3
    x = torch.cat([torch.randint(0, n + 1, (1_000_000,), device="cuda", dtype=torch.int64, generator=g), pos[:200_000] + 1, pos[-200_000:]])
    want = torch.searchsorted(pos, x)                   # entries below x
    assert torch.equal(v.rank(x, 1).to(torch.int64), want)
    k = torch.randint(1, n - pos.numel() + 1, (1_000_000,), device="cuda", dtype=torch.int64, generator=g)
    z = v.select(k, 0).to(torch.int64)                  # the k-th zero: k - 1 zeros in front of it, and it is a zero
    r1 = torch.searchsorted(pos, z)
    assert torch.equal(z - r1, k - 1)
    assert not torch.isin(z[:100_000], pos).any()
