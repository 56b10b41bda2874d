"""GPU parity of batched rank/select/access on plain bit vectors against a naive scan — the
assertion style of the reference's own tests (test/rank_support_test.cpp:109-128 compares every
rank(j) with a running count; test/select_support_test.cpp:85-104 checks select(++k)==j for every
set bit j).  Oracle-based and golden-vector tests live in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import unpack_bits

pytestmark = pytest.mark.gpu

EDGE_SIZES = [0, 1, 63, 64, 65, 383, 384, 385, 447, 448, 449, 895, 896, 897, 2047, 2048, 2049, 4096, 100001]


def make_words(n_bits, density, seed, stray=True):
    rng = np.random.default_rng(seed)
    nw = (n_bits + 63) // 64
    if density == 0.5:
        w = rng.integers(0, 2**64, size=nw, dtype=np.uint64)
    else:
        bits = (rng.random(nw * 64) < density).astype(np.uint8)
        w = np.packbits(bits, bitorder="little").view(np.uint64).copy()
    if not stray and n_bits % 64:
        w[-1] &= np.uint64((1 << (n_bits % 64)) - 1)
    return w


def check_all(pkg, words, n_bits):
    bits = unpack_bits(words, n_bits).astype(np.int64)
    bv = pkg.bit_vector(words, n_bits)
    assert bv.size() == n_bits
    assert bv.ones() == int(bits.sum())
    pref1 = np.concatenate([[0], np.cumsum(bits)]).astype(np.uint64)
    idx = np.arange(n_bits + 1, dtype=np.uint64)
    r1 = bv.rank(idx, 1)
    assert np.array_equal(r1, pref1)
    r0 = bv.rank(idx, 0)
    assert np.array_equal(r0, idx - pref1)
    for b in (1, 0):
        pos = np.nonzero(bits == b)[0].astype(np.uint64)
        if pos.size:
            got = bv.select(np.arange(1, pos.size + 1, dtype=np.uint64), b)
            assert np.array_equal(got, pos), f"select_{b} mismatch n={n_bits}"
        # outside SDSL's precondition: defined here as NPOS
        bad = bv.select(np.array([0, pos.size + 1], dtype=np.uint64), b)
        assert np.all(bad == np.uint64(pkg.capi.NPOS))
    if n_bits:
        acc = bv.access(np.arange(n_bits, dtype=np.uint64))
        assert np.array_equal(acc, bits.astype(np.uint8))
    assert np.all(bv.rank(np.array([n_bits + 1], dtype=np.uint64)) == np.uint64(pkg.capi.NPOS))
    back = bv.export_words()
    ref = words.copy()
    if n_bits % 64:
        ref[-1] &= np.uint64((1 << (n_bits % 64)) - 1)
    assert np.array_equal(back, ref[: (n_bits + 63) // 64])


@pytest.mark.parametrize("n_bits", EDGE_SIZES)
@pytest.mark.parametrize("density", [0.5, 0.0, 1.0, 0.01])
def test_edge_sizes(gpu, n_bits, density):
    check_all(gpu, make_words(n_bits, density, seed=n_bits + 7), n_bits)


@pytest.mark.parametrize("density", [0.5, 0.05, 0.95, 0.001])
def test_one_mbit(gpu, density):
    check_all(gpu, make_words(1 << 20, density, seed=3), 1 << 20)


def test_clustered_vector(gpu):
    # adversarial for the interpolated select probe: long empty stretches between dense clusters
    n = 3_000_000
    bits = np.zeros(n, dtype=np.uint8)
    rng = np.random.default_rng(5)
    for start in rng.integers(0, n - 5000, size=40):
        bits[start:start + 5000] = rng.random(5000) < 0.9
    bits[-1] = 1
    w = np.packbits(np.pad(bits, (0, (-n) % 64)), bitorder="little").view(np.uint64).copy()
    check_all(gpu, w, n)


def test_set_random_bits_known_answers(gpu):
    # bit_vector(2^20) + util::set_random_bits(bv, 815)
    w = gpu.set_random_bits(1 << 20, 815)
    bv = gpu.bit_vector(w)
    assert int(bv.rank(np.array([1 << 20], dtype=np.uint64))[0]) == 524053
    assert int(bv.select(np.array([1000], dtype=np.uint64))[0]) == 1961


def test_device_tensors_and_stream(gpu):
    import torch
    n = 1 << 22
    w = make_words(n, 0.5, 11)
    bits = unpack_bits(w, n).astype(np.int64)
    pref = np.concatenate([[0], np.cumsum(bits)])
    dw = torch.from_numpy(w.view(np.int64)).cuda()
    bv = gpu.bit_vector(dw, n)
    g = torch.Generator(device="cuda").manual_seed(1)
    idx = torch.randint(0, n + 1, (1_000_003,), device="cuda", generator=g, dtype=torch.int64)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = bv.rank(idx)
    s.synchronize()
    assert np.array_equal(out.cpu().numpy(), pref[idx.cpu().numpy()])
    ones = int(bits.sum())
    i = torch.randint(1, ones + 1, (500_001,), device="cuda", generator=g, dtype=torch.int64)
    pos = np.nonzero(bits)[0]
    got = bv.select(i).cpu().numpy()
    assert np.array_equal(got, pos[i.cpu().numpy() - 1])


def test_host_arrays_take_the_pipelined_path(gpu):
    """>= 2^23 queries in host arrays: chunks on two streams (upload / kernel / download overlapped) — same answers as
    the device-resident call, including a ragged last chunk and out-of-range arguments"""
    import torch
    n = (1 << 26) + 77
    w = gpu.set_random_bits(n, 3)
    bv = gpu.bit_vector(w, n)
    rng = np.random.default_rng(2)
    nq = (1 << 23) + (1 << 22) + 12345
    idx = rng.integers(0, n + 1, nq).astype(np.uint64)
    idx[5] = n + 9  # outside: NPOS
    out = np.empty(nq, dtype=np.uint64)
    bv.rank(idx, 1, out)
    dev = bv.rank(torch.from_numpy(idx.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
    assert np.array_equal(out, dev) and out[5] == np.uint64(2**64 - 1)
    i = rng.integers(1, bv.ones() + 1, nq).astype(np.uint64)
    i[7] = 0
    bv.select(i, 1, out)
    dev = bv.select(torch.from_numpy(i.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
    assert np.array_equal(out, dev) and out[7] == np.uint64(2**64 - 1)
    # the compressed vectors take the same path
    import oracle_lib as ol
    for vec in (gpu.rrr_vector(w, n), gpu.sd_vector(w, n)):
        idx[5] = n  # in range for both
        vec.rank(idx, 1, out)
        assert np.array_equal(out[: 1 << 12], ol.OBitVector(w, n).rank(idx[: 1 << 12], 1))
        d1 = vec.rank(torch.from_numpy(idx.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
        assert np.array_equal(out, d1)
        i[7] = 1
        vec.select(i, 1, out)
        d1 = vec.select(torch.from_numpy(i.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
        assert np.array_equal(out, d1)
    nosel = gpu.bit_vector(w, n, select1=False, select0=False)
    with pytest.raises(gpu.capi.SdslHipError):  # an error inside a pipeline worker reaches the caller
        nosel.select(i, 1, out)

