"""Option "rrr_sparse_limit" (default 20 since round 6; 10 before): an rrr_vector<63> of 10-30 % density keeps the classes up to 20 (instead of 10) enumerative — the
space of SDSL's vector plus the record overhead instead of 63 raw bits for every block of eleven or more ones (rrr.hip:
choose_sparse_max; the reference decodes every class from its offset, rrr_vector.hpp:158-270 / rrr_helper.hpp:480-534).  The
decoder then walks up to eighteen bisections per block; every query kernel, the device encoder, the loader and the writer of
SDSL's stream have to agree with the default vector, with numpy, and with the real library's bytes."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def words_of(bits):
    pad = (-bits.size) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)])
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


def build(gpu, w, n_bits, limit, **kw):
    gpu.set_option("rrr_sparse_limit", limit)
    try:
        return gpu.rrr_vector(w, n_bits) if w is not None else gpu.rrr_vector(**kw)
    finally:
        gpu.set_option("rrr_sparse_limit", 20)  # (the default since round 6)


@pytest.mark.parametrize("n_bits,d", [(700_001, 0.10), (700_001, 0.15), (900_000, 0.20), (500_000, 0.30), (600_000, 0.85),
                                      (2142 * 300 + 7, 0.25), (63 * 34, 0.2), (40, 0.3)])
def test_limit_20_equals_default_numpy_and_sdsl_bytes(gpu, n_bits, d):
    rng = np.random.default_rng(int(d * 1000) + n_bits % 97)
    bits = rng.random(n_bits) < d
    if n_bits > 100_000:
        bits[5000:9000] = rng.random(4000) < 0.5       # a stretch of middle classes: raw under every limit
        bits[20_000:23_000] = rng.random(3000) < 0.03  # and a sparse one
    w = words_of(bits)
    ref, cmp = build(gpu, w, n_bits, 10), build(gpu, w, n_bits, 20)
    assert cmp.ones() == ref.ones() == int(bits.sum())
    if n_bits > 100_000:
        assert cmp.device_bytes() < ref.device_bytes(), "classes 11..20 as offsets are smaller than 63 raw bits"
    cum = np.concatenate([[0], np.cumsum(bits, dtype=np.int64)]).astype(np.uint64)
    idx = np.concatenate([rng.integers(0, n_bits + 1, 80_000, dtype=np.uint64), np.arange(min(n_bits + 1, 5000), dtype=np.uint64),
                          np.arange(max(0, n_bits - 5000), n_bits + 1, dtype=np.uint64)])
    for route in (0, 1):                                # 1: "whenever possible" — the bucketed route, whose slice decoder holds the columns 3..20 (round 6)
        gpu.set_option("rrr_sorted", route)
        try:
            assert np.array_equal(np.asarray(cmp.rank(idx, 1)), cum[idx]), "rank_1 against numpy"
            assert np.array_equal(np.asarray(cmp.rank(idx, 0)), idx - cum[idx]), "rank_0 against numpy"
            for bit in (0, 1):
                pos = np.flatnonzero(bits == bool(bit)).astype(np.uint64)
                if pos.size == 0:
                    continue
                i = np.concatenate([rng.integers(1, pos.size + 1, 60_000, dtype=np.uint64), np.arange(1, min(pos.size, 4000) + 1, dtype=np.uint64),
                                    np.arange(max(1, pos.size - 4000), pos.size + 1, dtype=np.uint64)])
                assert np.array_equal(np.asarray(cmp.select(i, bit)), pos[i - np.uint64(1)]), f"select_{bit} against numpy"
                assert np.array_equal(np.asarray(cmp.select(i, bit)), np.asarray(ref.select(i, bit)))
        finally:
            gpu.set_option("rrr_sorted", -1)
    inside = idx[idx < n_bits]
    assert np.array_equal(np.asarray(cmp.access(inside)).astype(bool), bits[inside.astype(np.int64)])
    for length in (1, 17, 63, 64):
        ok = idx[idx + np.uint64(length) <= n_bits] if n_bits >= length else idx[:0]
        if len(ok):
            assert np.array_equal(np.asarray(cmp.get_int(ok, length)), np.asarray(ref.get_int(ok, length))), f"get_int {length}"
    # SDSL's stream: the same bytes from both (the default's are the real library's: tests/test_gpu_parity.py), the oracle's
    # writer agrees, and a vector loaded from them under the limit answers the same and writes them again
    sb = cmp.serialize()
    assert sb == ref.serialize()
    assert sb == ol.ORrr(w, n_bits).serialize()
    again = build(gpu, None, None, 20, sdsl_bytes=sb)
    assert again.serialize() == sb
    assert abs(again.device_bytes() - cmp.device_bytes()) <= 4096  # (a handle that has answered a bucketed batch keeps its verdict word and bucket plans)
    assert np.array_equal(np.asarray(again.rank(idx, 1)), cum[idx])


def test_space_against_sdsl_between_10_and_30_percent(gpu):
    """bits per bit of the device vector over bits per bit of SDSL's serialised rrr_vector<63>: the default pays 63 raw bits
    for every block of class 11..52, the limit 20 only for 21..42."""
    n_bits = 1 << 27  # (the device's fixed part — binomial table, select directories: 0.5 MB — is 5 % of SDSL's bytes here, 1 % at 2^30)
    rows = []
    for d in (0.10, 0.15, 0.20, 0.30):
        bits = np.random.default_rng(int(d * 100)).random(n_bits) < d
        w = words_of(bits)
        a, b = build(gpu, w, n_bits, 10), build(gpu, w, n_bits, 20)
        sdsl = len(b.serialize())
        rows.append((d, a.device_bytes() / sdsl, b.device_bytes() / sdsl))
        idx = np.random.default_rng(1).integers(0, n_bits + 1, 500_000, dtype=np.uint64)
        assert np.array_equal(np.asarray(a.rank(idx, 1)), np.asarray(b.rank(idx, 1)))
    print("density, default / SDSL, limit 20 / SDSL:", rows)
    for d, r10, r20 in rows:
        assert r20 <= 1.25, rows
        assert r20 <= r10 - (0.0 if d <= 0.10 else 0.10), rows


def test_the_default_is_20_and_large_batches_stay_bucketed(gpu):
    """a stand-alone vector of 20 % density built with NO option set keeps the classes up to 20 enumerative (smaller than with the limit at
    10) and a large batch takes the bucketed route on it (last_phases names its passes) with the direct kernels' answers; the vectors
    inside a wavelet tree keep the limit 10 (count() decodes a block per level and lane)."""
    import torch
    n_bits = 1 << 28
    w = gpu.density_bits(n_bits, 27, 20)
    v = gpu.rrr_vector(w, n_bits)
    gpu.set_option("rrr_sparse_limit", 10)
    try:
        old = gpu.rrr_vector(w, n_bits)
    finally:
        gpu.set_option("rrr_sparse_limit", 20)
    assert v.device_bytes() < 0.92 * old.device_bytes()
    idx = torch.randint(0, n_bits + 1, (30_000_000,), device="cuda", dtype=torch.int64)
    gpu.set_option("trace_phases", 1)
    gpu.set_option("rrr_sorted", 1)
    try:
        got = v.rank(idx, 1)
        phases = gpu.last_phases()
        sel_i = torch.randint(1, v.ones() + 1, (20_000_000,), device="cuda", dtype=torch.int64)
        got_s = v.select(sel_i, 1)
    finally:
        gpu.set_option("rrr_sorted", 0)
        gpu.set_option("trace_phases", 0)
    try:
        assert phases, "the batch did not take the bucketed route"
        assert torch.equal(got, v.rank(idx, 1)) and torch.equal(got, old.rank(idx, 1))
        assert torch.equal(got_s, v.select(sel_i, 1)) and torch.equal(got_s, old.select(sel_i, 1))
    finally:
        gpu.set_option("rrr_sorted", -1)
    v.close()
    old.close()
