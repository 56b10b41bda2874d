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
