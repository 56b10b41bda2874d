"""Vectors of more than 2^26 rank lines (2^34.8 bits): the bucketed rank keeps 2^16 slices and makes them wider (2^11 .. 2^13
lines), the answering kernel stages a slice in rounds of 2^10 lines (bv_sorted.hip: k_sr_rank_lds<MULTI>).  Both bit values
against the direct kernel on the same batch, and — at 2^36 bits on SURVEY 8(d)'s seeded vector — against digests of the real
sdsl-lite's rank_support_v5 (tests/golden/golden_large.json: c2w, made by make_golden_large.py c2w)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_large.json")))


def both_routes(gpu, bv, idx, bit):
    import torch
    gpu.set_option("rank_sorted", 0)
    want = bv.rank(idx, bit).clone()
    gpu.set_option("rank_sorted", 1)
    gpu.set_option("trace_phases", 1)
    try:
        got = bv.rank(idx, bit)
        torch.cuda.synchronize()
        ph = gpu.last_phases()
    finally:
        gpu.set_option("trace_phases", 0)
        gpu.set_option("rank_sorted", -1)
    assert ph.get("part1", 0) > 0, f"the batch did not take the passes: {ph}"
    return got, want


@pytest.mark.parametrize("n_bits", [448 * ((1 << 26) + 12345) + 77, (1 << 35) + (1 << 34) + 4242, 448 * ((1 << 28) + 3) + 5])
def test_wide_slices_answer_like_the_direct_kernel(gpu, n_bits):
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n_bits % 1000)
    words = torch.randint(-2**63, 2**63 - 1, ((n_bits + 63) // 64,), device=dev, dtype=torch.int64, generator=g)
    bv = gpu.bit_vector(words, n_bits, select1=False, select0=False)
    del words
    nq = 60_000_000
    idx = torch.randint(0, n_bits + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    idx[:1000] = n_bits                      # the very end
    idx[1000:2000] = torch.arange(1000, device=dev)   # the very start
    idx[2000:3000] = n_bits + 7              # beyond: NPOS
    for bit in (1, 0):
        got, want = both_routes(gpu, bv, idx, bit)
        bad = (got != want).nonzero().flatten()
        assert bad.numel() == 0, f"n_bits {n_bits}, bit {bit}: query {int(bad[0])} at {int(idx[bad[0]])}: {int(got[bad[0]])} != {int(want[bad[0]])}"
    # a batch that lives in ONE round of one slice, and one spread over two slices only
    L = 1 << 13
    local = (n_bits // 3) // (448 * L) * (448 * L) + torch.randint(0, 448 * 1024, (5_000_000,), device=dev, dtype=torch.int64, generator=g)
    got, want = both_routes(gpu, bv, local, 1)
    assert torch.equal(got, want)
    bv.release_scratch()
    bv.close()


@pytest.mark.skipif("c2w" not in G, reason="golden_large.json has no c2w section (tests/golden/make_golden_large.py c2w)")
def test_2_pow_36_bits_match_the_real_rank_support_v5(gpu):
    import hashlib
    import torch
    c = G["c2w"]
    n = 1 << c["log_n"]
    dev = torch.device("cuda:0")
    words = gpu.rnd_positions_device(c["words_seed"], n // 64, 0, 0, 0)
    bv = gpu.bit_vector(words, n, select1=False, select0=False)
    del words
    assert bv.ones() == c["ones"]
    nq = c["rank_1"]["n"]
    idx = gpu.rnd_positions_device(c["rank_seed"], nq, n + 1, 0, 0)
    for bit, key in ((1, "rank_1"), (0, "rank_0")):
        want = c[key]
        for route in (1, 0):
            gpu.set_option("rank_sorted", route)
            try:
                a = bv.rank(idx, bit).cpu().numpy().view(np.uint64)
            finally:
                gpu.set_option("rank_sorted", -1)
            first = np.array(want["first"], dtype=np.uint64)
            assert np.array_equal(a[: first.size], first), f"{key}, route {route}: first answers differ from the reference's"
            assert int(np.add.reduce(a, dtype=np.uint64)) == want["sum"] and hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"], (key, route)
    bv.release_scratch()
    bv.close()
