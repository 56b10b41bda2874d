"""(run by tests/test_gpu_group.py in a child process)  GPU: the multi-device group of the C ABI on whatever devices the box has — on a one-GPU box the group
has one member, which still drives RCCL communicator creation, the load-time broadcast, the loop-back send/recv through
both communicators and the chunked three-stream driver; with more GPUs the same test shards for real."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["rccl", "copy"])
def group(pkg, request):
    """rccl: the devices the box has, RCCL transport.  copy: THREE members on device 0 over the copy transport
    (SDSL_HIP_GROUP_TRANSPORT=copy: device-to-device copies ordered by events; the only transport that accepts a device twice) —
    uneven shards, peers' staging buffers, the chunk pipeline and its event chains for G > 1 run on a one-GPU box this way."""
    import os
    import torch
    L = pkg.capi.lib()
    g = C.c_void_p(None)
    if request.param == "copy":
        n = 3
        devs = (C.c_int32 * n)(0, 0, 0)
        os.environ["SDSL_HIP_GROUP_TRANSPORT"] = "copy"
        try:
            pkg.capi.check(L.sdsl_hip_group_create(devs, n, C.byref(g)))
        finally:
            del os.environ["SDSL_HIP_GROUP_TRANSPORT"]
    else:
        n = torch.cuda.device_count()
        devs = (C.c_int32 * n)(*range(n))
        pkg.capi.check(L.sdsl_hip_group_create(devs, n, C.byref(g)))
    assert L.sdsl_hip_group_size(g) == n and L.sdsl_hip_group_device(g, 0) == 0
    yield g, n
    pkg.capi.check(L.sdsl_hip_group_destroy(g))


def test_group_loopback_moves_data_around_the_group(pkg, group):
    g, n = group
    ms = C.c_float(0)
    pkg.capi.check(pkg.capi.lib().sdsl_hip_group_loopback(g, 1 << 20, C.byref(ms)))
    assert ms.value > 0


def test_group_rejects_bad_arguments(pkg):
    L = pkg.capi.lib()
    g = C.c_void_p(None)
    two = (C.c_int32 * 2)(0, 0)
    assert L.sdsl_hip_group_create(two, 2, C.byref(g)) == pkg.capi.ERR_INVALID  # the same device twice
    assert L.sdsl_hip_group_create(None, 1, C.byref(g)) == pkg.capi.ERR_INVALID
    far = (C.c_int32 * 1)(99)
    assert L.sdsl_hip_group_create(far, 1, C.byref(g)) != 0


@pytest.mark.parametrize("chunks", [1, 4])
def test_group_bv_rank_select_equal_single_gpu_answers(pkg, group, chunks):
    import torch
    g, n = group
    L = pkg.capi.lib()
    rng = np.random.default_rng(5)
    n_bits = (1 << 22) + 77
    words = rng.integers(0, 2**63, size=(n_bits + 63) // 64, dtype=np.int64).astype(np.uint64)
    bv = pkg.bit_vector(words, n_bits, device=0)
    reps = (C.c_void_p * n)()
    pkg.capi.check(L.sdsl_hip_group_bv_replicate(g, bv._h, reps))
    try:
        nq = 1_000_003
        idx = rng.integers(0, n_bits + 1, size=nq).astype(np.uint64)
        want = bv.rank(idx, 1)
        # host-resident batch
        got = np.empty(nq, dtype=np.uint64)
        pkg.capi.check(L.sdsl_hip_group_bv_rank_batch(g, reps, 1, idx.ctypes.data, nq, got.ctypes.data, chunks))
        assert np.array_equal(got, want)
        # root-device-resident batch
        d_idx = torch.from_numpy(idx.view(np.int64)).cuda()
        d_out = torch.empty_like(d_idx)
        pkg.capi.check(L.sdsl_hip_group_bv_rank_batch(g, reps, 0, d_idx.data_ptr(), nq, d_out.data_ptr(), chunks))
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), bv.rank(idx, 0))
        ones = bv.ones()
        si = rng.integers(1, ones + 1, size=nq).astype(np.uint64)
        pkg.capi.check(L.sdsl_hip_group_bv_select_batch(g, reps, 1, si.ctypes.data, nq, got.ctypes.data, chunks))
        assert np.array_equal(got, bv.select(si, 1))
    finally:
        for r in range(1, n):
            L.sdsl_hip_bv_destroy(reps[r])


def test_group_fm_count_equals_single_gpu(pkg, group):
    g, n = group
    L = pkg.capi.lib()
    text = pkg.english_text(1 << 20, 77)
    csa = pkg.csa_wt(text=text, device=0)
    reps = (C.c_void_p * n)()
    pkg.capi.check(L.sdsl_hip_group_fm_create_from_text(g, text.ctypes.data, text.size, 0, reps))
    try:
        rng = np.random.default_rng(9)
        m, npat = 12, 200_003
        st = rng.integers(0, text.size - m, size=npat)
        pats = np.ascontiguousarray(text[st[:, None] + np.arange(m)[None, :]].reshape(-1))
        want = csa.count(pats, m)
        got = np.empty(npat, dtype=np.uint64)
        pkg.capi.check(L.sdsl_hip_group_fm_count_batch(g, reps, pats.ctypes.data, m, npat, got.ctypes.data, 3))
        assert np.array_equal(got, want) and int(got.min()) >= 1
    finally:
        for r in range(n):
            L.sdsl_hip_fm_destroy(reps[r])


def test_a_peer_that_never_turns_up_is_an_error_not_a_hang(pkg):
    """Round 6: a batch has a deadline (option group_timeout_ms).  Three members on device 0 over the copy transport; member 1's scatter stream
    is held by the test hook (`group_test_stall`: a kernel that spins on a host word — in the copy transport a receive that is never posted
    IS a scatter stream that never reaches its copy).  The call must come back inside ten seconds with SDSL_HIP_ERR_HIP, name the member and
    the stage, refuse further batches on the group — and a fresh group works once the peer is released."""
    import os
    import time
    L = pkg.capi.lib()
    os.environ["SDSL_HIP_GROUP_TRANSPORT"] = "copy"
    try:
        g = C.c_void_p(None)
        pkg.capi.check(L.sdsl_hip_group_create((C.c_int32 * 3)(0, 0, 0), 3, C.byref(g)))
        rng = np.random.default_rng(3)
        n_bits = (1 << 20) + 5
        words = rng.integers(0, 2**63, size=(n_bits + 63) // 64, dtype=np.int64).astype(np.uint64)
        bv = pkg.bit_vector(words, n_bits, device=0)
        reps = (C.c_void_p * 3)()
        pkg.capi.check(L.sdsl_hip_group_bv_replicate(g, bv._h, reps))
        nq = 100_000
        idx = rng.integers(0, n_bits + 1, size=nq).astype(np.uint64)
        got = np.empty(nq, dtype=np.uint64)
        pkg.capi.check(L.sdsl_hip_group_bv_rank_batch(g, reps, 1, idx.ctypes.data, nq, got.ctypes.data, 2))
        assert np.array_equal(got, bv.rank(idx, 1))
        pkg.set_option("group_timeout_ms", 2000)
        pkg.set_option("group_test_stall", 1)
        t0 = time.time()
        st = L.sdsl_hip_group_bv_rank_batch(g, reps, 1, idx.ctypes.data, nq, got.ctypes.data, 2)
        took = time.time() - t0
        msg = L.sdsl_hip_last_error().decode()
        pkg.set_option("group_test_stall", -1)            # the peer turns up after all: the stuck stream drains
        assert st == pkg.capi.ERR_HIP, (st, msg)
        assert took < 10.0, took
        # (every unfinished stage is listed; with all three members on ONE device a stream of another member may share the stuck one's
        # hardware queue and be listed too — member 1's own three stages always are)
        assert all(f"member 1 (device 0) {st_}" in msg for st_ in ("scatter", "kernels", "gather")) and "group_timeout_ms" in msg, msg
        st2 = L.sdsl_hip_group_bv_rank_batch(g, reps, 1, idx.ctypes.data, nq, got.ctypes.data, 2)
        assert st2 == pkg.capi.ERR_HIP and "destroy" in L.sdsl_hip_last_error().decode()
        import torch
        torch.cuda.synchronize()
        for r in range(1, 3):
            L.sdsl_hip_bv_destroy(reps[r])
        pkg.capi.check(L.sdsl_hip_group_destroy(g))
        # a fresh group on the same device answers
        pkg.set_option("group_timeout_ms", 120000)
        g2 = C.c_void_p(None)
        pkg.capi.check(L.sdsl_hip_group_create((C.c_int32 * 3)(0, 0, 0), 3, C.byref(g2)))
        pkg.capi.check(L.sdsl_hip_group_bv_replicate(g2, bv._h, reps))
        pkg.capi.check(L.sdsl_hip_group_bv_rank_batch(g2, reps, 1, idx.ctypes.data, nq, got.ctypes.data, 2))
        assert np.array_equal(got, bv.rank(idx, 1))
        for r in range(1, 3):
            L.sdsl_hip_bv_destroy(reps[r])
        pkg.capi.check(L.sdsl_hip_group_destroy(g2))
    finally:
        pkg.set_option("group_test_stall", -1)
        pkg.set_option("group_timeout_ms", 120000)
        del os.environ["SDSL_HIP_GROUP_TRANSPORT"]
