"""GPU: select under non-uniform density at 2^28 bits against the oracle's select_support_mcl —
ones clustered in 1 % of the range, alternating dense / empty 2^20-bit stripes, isolated ones (CRAFTED-SPARSE style: the
stretches that select_support_mcl stores as "long" blocks, select_support_mcl.hpp:242-252, and this library as fully
sampled intervals), a mixture of all of them; plain vector, rrr_vector<63> and sd_vector; both bit values."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
N = 1 << 28


def shape(name):
    rng = np.random.default_rng(hash(name) % 1000)
    nw = N // 64
    w = np.zeros(nw, dtype=np.uint64)
    dense = lambda k: rng.integers(0, 2**63, size=k, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=k).astype(np.uint64)
    if name == "clustered":
        lo, cnt = int(nw * 0.495), nw // 100
        w[lo:lo + cnt] = dense(cnt)
    elif name == "stripes":
        ws = (1 << 20) // 64
        v = dense(nw).reshape(-1, ws)
        v[1::2] = 0
        w = v.reshape(-1)
    elif name == "isolated":
        k = N >> 16
        pos = np.arange(k, dtype=np.int64) * (1 << 16) + rng.integers(0, 1 << 16, size=k)
        np.bitwise_or.at(w, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
    elif name == "uniform":
        w = dense(nw)
    elif name == "mixed":  # a dense quarter, an isolated quarter, a burst, emptiness, one bit at the very end
        q = nw // 4
        w[:q] = dense(q)
        k = (N // 4) >> 14
        pos = N // 4 + np.arange(k, dtype=np.int64) * (1 << 14) + rng.integers(0, 1 << 14, size=k)
        np.bitwise_or.at(w, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
        w[2 * q + 1000: 2 * q + 1100] = dense(100)
        w[-1] |= np.uint64(1) << np.uint64(63)
    return w


@pytest.mark.parametrize("mode", [0, 1], ids=["direct", "bucketed"])
@pytest.mark.parametrize("name", ["clustered", "stripes", "isolated", "mixed", "uniform"])
def test_plain_select_matches_oracle_on_shape(gpu, name, mode):
    """both ways of answering a batch: the direct kernel, and the bucketed path (bv_sorted.hip) whose buckets of 2^r
    consecutive ranks are staged in LDS when they fit a slice and left to its fix-up pass when they do not (the sparse
    stretches of these shapes), on device-resident batches"""
    import torch
    w = shape(name)
    o = ol.OBitVector(w, N)
    bv = gpu.bit_vector(w, N)
    rng = np.random.default_rng(1)
    gpu.set_option("select_sorted", mode)
    try:
        for b in (1, 0):
            ac = o.arg_cnt(b)
            i = np.concatenate([rng.integers(1, ac + 1, size=300_000, dtype=np.uint64),
                                np.array([1, 2, ac - 1, ac, ac + 1, 0], dtype=np.uint64),
                                np.arange(1, min(ac, 5000) + 1, dtype=np.uint64)])
            want = o.select(i[(i >= 1) & (i <= ac)], b)
            got = bv.select(torch.from_numpy(i.view(np.int64)).cuda(), b).cpu().numpy().view(np.uint64)
            ok = (i >= 1) & (i <= ac)
            assert np.array_equal(got[ok], want), (name, b, mode)
            assert (got[~ok] == np.uint64(2**64 - 1)).all(), "arguments outside [1, #args] answer NPOS"
    finally:
        gpu.set_option("select_sorted", -1)


@pytest.mark.parametrize("name", ["clustered", "isolated", "mixed"])
def test_rrr_and_sd_select_match_oracle_on_shape(gpu, name):
    w = shape(name)
    o = ol.OBitVector(w, N)
    rng = np.random.default_rng(2)
    ac = o.arg_cnt(1)
    i = np.concatenate([rng.integers(1, ac + 1, size=200_000, dtype=np.uint64), np.array([1, ac], dtype=np.uint64)])
    want = o.select(i, 1)
    rv = gpu.rrr_vector(w, N)
    assert np.array_equal(rv.select(i, 1), want), name
    z = rng.integers(1, N - ac + 1, size=100_000, dtype=np.uint64)
    assert np.array_equal(rv.select(z, 0), o.select(z, 0)), name
    rv.close()
    sd = gpu.sd_vector(words=w, n_bits=N)
    assert np.array_equal(sd.select(i, 1), want), name
    sd.close()


def test_long_interval_tables_are_built_only_for_sparse_stretches(gpu):
    dense = gpu.bit_vector(shape("stripes"), N)
    sparse = gpu.bit_vector(shape("isolated"), N)
    # the dense vector's directories stay small; the isolated one keeps every position of its 4096 ones (and of no zero)
    assert sparse.device_bytes() - dense.device_bytes() < (1 << 20)
