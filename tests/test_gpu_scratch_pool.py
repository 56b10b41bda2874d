"""The bucketed batch paths work in ONE scratch pool per device (bv_host.hpp: DeviceScratch): handles share it, it grows to the
largest pass, release_scratch frees it, and the next batch brings it back — answers unchanged throughout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_handles_share_one_pool(gpu):
    rng = np.random.default_rng(3)
    n_bits = 448 * 1024 * 64 + 17
    w = rng.integers(0, 2**63, (n_bits + 63) // 64, dtype=np.int64).view(np.uint64)
    a, b = gpu.bit_vector(w, n_bits), gpu.bit_vector(w[::-1].copy(), n_bits)
    rv = gpu.rrr_vector(w & (w >> np.uint64(1)) & (w >> np.uint64(2)) & (w >> np.uint64(3)), n_bits)
    idx = rng.integers(0, n_bits + 1, 2_000_000, dtype=np.uint64)
    small = idx[:300_000]
    a.release_scratch()
    assert gpu.device_scratch_bytes(0) == 0
    base_a, base_rv = a.device_bytes(), rv.device_bytes()
    try:
        gpu.set_option("rank_sorted", 0)
        gpu.set_option("rrr_sorted", 0)
        want_a, want_b, want_rv = a.rank(idx, 1), b.rank(small, 1), rv.rank(idx, 1)
        gpu.set_option("rank_sorted", 1)
        gpu.set_option("rrr_sorted", 1)
        assert np.array_equal(b.rank(small, 1), want_b)
        first = gpu.device_scratch_bytes(0)
        assert first > 0
        assert np.array_equal(a.rank(idx, 1), want_a)          # a larger pass: the pool grows, once, for everybody
        grown = gpu.device_scratch_bytes(0)
        assert grown > first
        assert np.array_equal(b.rank(small, 1), want_b) and np.array_equal(rv.rank(idx, 1), want_rv)
        assert gpu.device_scratch_bytes(0) == grown            # ... and is what the other handles use, rrr vectors included
        assert a.device_bytes() <= base_a + 4096 and rv.device_bytes() <= base_rv + 4096  # the handles themselves hold none of it
        b.release_scratch()                                    # any handle frees the device's pool
        assert gpu.device_scratch_bytes(0) == 0
        assert np.array_equal(rv.rank(idx, 1), want_rv) and gpu.device_scratch_bytes(0) > 0
    finally:
        gpu.set_option("rank_sorted", -1)
        gpu.set_option("rrr_sorted", -1)
        a.release_scratch()
