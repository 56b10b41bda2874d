"""GPU: handles give back every byte of HBM — after normal use, after queries through the host-array path (its
staging buffers and streams), and after loaders rejected damaged streams half-way through their uploads."""
import gc
import os

import numpy as np
import pytest

import golden_data as gd

pytestmark = pytest.mark.gpu
SLACK = 1  # the library's own accounting is exact


def _held_bytes():
    """device memory the library holds for this process (its own accounting: other processes on the GPU do not disturb
    it, unlike the device-wide free-memory figure)"""
    import importlib
    import torch
    torch.cuda.synchronize()
    return importlib.import_module("sdsl-lite_amd").capi.lib().sdsl_hip_allocated_bytes()


def _settled(fn, rounds):
    """run fn `rounds` times after one warm-up call (which lets the runtime create its pools), return the change of
    free device memory across the timed rounds"""
    fn()
    gc.collect()
    before = _held_bytes()
    for _ in range(rounds):
        fn()
    gc.collect()
    return _held_bytes() - before


def test_build_query_destroy_cycles_do_not_leak(gpu):
    rng = np.random.default_rng(5)
    n = 1 << 26
    w = gpu.set_random_bits(n, 7)
    text = rng.integers(97, 101, size=4_000_000, dtype=np.uint8).tobytes()
    idx = rng.integers(0, n, size=10_000).astype(np.uint64)

    def once():
        bv = gpu.bit_vector(w, n)
        bv.rank(idx)
        bv.select(idx[:100] % np.uint64(bv.ones()) + np.uint64(1))
        rv = gpu.rrr_vector(w, n)
        rv.rank(idx)
        rv.serialize()
        sd = gpu.sd_vector(w, n)
        sd.rank(idx)
        pv = gpu.bit_vector(w, n, pattern=(10, 2))
        pv.rank(idx)
        for kw in ({}, {"rrr": True}, {"hutu": True}, {"balanced": True}):
            csa = gpu.csa_wt(text=text, **kw)
            pats = np.frombuffer(text[1000:1000 + 8 * 64], dtype=np.uint8)
            csa.count(pats, 8)
            csa.locate(pats, 8)
            csa.sa(idx[:100] % np.uint64(csa.size()))
            csa.serialize(32, 64)
            csa.drop_sa()
            csa.sa(idx[:100] % np.uint64(csa.size()))
            csa.extract(np.array([5], dtype=np.uint64), np.array([500], dtype=np.uint64))
            csa.close()
        for h in (bv, rv, sd, pv):
            h.close()

    leaked = _settled(once, 4)
    assert leaked < SLACK, f"{leaked} bytes of HBM did not come back"


def test_host_array_pipeline_releases_its_staging(gpu):
    n = 1 << 24
    bv = gpu.bit_vector(gpu.set_random_bits(n, 3), n)
    idx = np.random.default_rng(1).integers(0, n, size=(1 << 23) + 12345).astype(np.uint64)  # above the pipelining threshold

    def once():
        bv.rank(idx)

    leaked = _settled(once, 3)
    assert leaked < SLACK, f"{leaked} bytes of staging were kept"
    bv.close()


def test_rejected_streams_do_not_leak(gpu):
    """every truncation point of valid streams: the loader fails somewhere between its uploads and must undo them"""
    cases = [
        ("faust.txt.csa_wt_huff_v5.sdsl", lambda b: gpu.csa_wt(sdsl_bytes=b, select_is_mcl=True)),
        ("faust.txt.csa_wt_huff_rrr63.sdsl", lambda b: gpu.csa_wt(sdsl_bytes=b, rrr=True)),
        ("example01.txt.wt_huff_v5_mcl.sdsl", lambda b: gpu.wt_huff(sdsl_bytes=b, select_is_mcl=True)),
        ("faust.txt.wt_huff_rrr63.sdsl", lambda b: gpu.wt_huff(sdsl_bytes=b, rrr=True)),
        ("rnd.8192.1043.rrr63.sdsl", lambda b: gpu.rrr_vector(sdsl_bytes=b)),
        ("CRAFTED-32.sd_vector.sdsl", lambda b: gpu.sd_vector(sdsl_bytes=b)),
    ]
    blobs = [(gd.sdsl_file(name), ld) for name, ld in cases]
    refused = [0]

    def once():
        for blob, ld in blobs:
            for cut in np.linspace(8, len(blob) - 1, 16).astype(int):
                try:
                    h = ld(blob[:cut])
                except gpu.capi.SdslHipError:
                    refused[0] += 1
                    continue
                h.close()  # a prefix that happens to parse is fine too

    leaked = _settled(once, 3)
    assert refused[0] > 0
    assert leaked < SLACK, f"{leaked} bytes leaked on error paths"


def test_concurrent_queries_from_host_threads(gpu):
    """include/sdsl_hip.h: query calls are const on the handle and may run concurrently from several host threads
    (SDSL's own rule for its const members).  Eight threads hammer shared handles; every answer must equal the
    single-threaded one."""
    import threading
    rng = np.random.default_rng(11)
    n = 1 << 24
    w = gpu.set_random_bits(n, 9)
    bv = gpu.bit_vector(w, n)
    rv = gpu.rrr_vector(w, n)
    sd = gpu.sd_vector(w, n)
    text = rng.integers(97, 105, size=1_000_000, dtype=np.uint8).tobytes()
    csa = gpu.csa_wt(text=text)
    crrr = gpu.csa_wt(text=text, rrr=True)
    crrr.drop_sa()
    arr = np.frombuffer(text, dtype=np.uint8)
    jobs = []
    for t in range(8):
        r = np.random.default_rng(100 + t)
        idx = r.integers(0, n + 1, size=200_000).astype(np.uint64)
        k = r.integers(1, bv.ones() + 1, size=100_000).astype(np.uint64)
        st = r.integers(0, len(text) - 12, size=3000)
        pats = np.concatenate([arr[s:s + 12] for s in st])
        si = r.integers(0, csa.size(), size=2000).astype(np.uint64)
        jobs.append((idx, k, pats, si))

    def answers(job):
        idx, k, pats, si = job
        off, pos = csa.locate(pats, 12)
        eo, et = crrr.extract(si[:50], np.minimum(si[:50] + np.uint64(40), np.uint64(csa.size() - 1)))
        return [bv.rank(idx), bv.select(k), rv.rank(idx), rv.select(k), sd.rank(idx), sd.select(k), csa.count(pats, 12),
                crrr.count(pats, 12), off, pos, csa.sa(si), crrr.sa(si[:300]), crrr.isa(si[:300]), eo, et,
                csa.wavelet_tree.rank(si, arr[:si.size])]

    expect = [answers(j) for j in jobs]
    got = [None] * len(jobs)
    errors = []

    def worker(t):
        try:
            for _ in range(3):
                got[t] = answers(jobs[t])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(len(jobs)):
        for a, b in zip(expect[t], got[t]):
            assert np.array_equal(a, b), f"thread {t} got a different answer"
