"""GPU: the stream loaders under byte-level corruption.  Every mutated stream must either be refused with an
SdslHipError or load into a structure whose queries terminate and stay inside their documented value ranges — never
crash, hang or fault.  (Seeded, a few hundred mutations per stream type; the streams are the real library's.)"""
import numpy as np
import pytest

import golden_data as gd

pytestmark = pytest.mark.gpu
NPOS = np.uint64(2**64 - 1)


def _mutations(blob: bytes, count: int, seed: int):
    rng = np.random.default_rng(seed)
    n = len(blob)
    for k in range(count):
        b = bytearray(blob)
        kind = k % 4
        if kind == 0:      # flip one bit
            p = int(rng.integers(0, n))
            b[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:    # overwrite an aligned 8-byte word (headers, counters, pointers)
            p = int(rng.integers(0, max(1, n // 8))) * 8
            b[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        elif kind == 2:    # a burst of random bytes
            p = int(rng.integers(0, n))
            ln = int(rng.integers(1, 33))
            b[p:p + ln] = rng.integers(0, 256, min(ln, n - p), dtype=np.uint8).tobytes()
        else:              # truncate or extend
            cut = int(rng.integers(0, n))
            b = b[:cut] if k % 8 == 3 else b + bytearray(rng.integers(0, 256, 16, dtype=np.uint8).tobytes())
        yield bytes(b)


def test_fuzz_rrr_stream(gpu):
    blob = gd.sdsl_file("rnd.8192.1043.rrr63.sdsl")
    accepted = 0
    for mut in _mutations(blob, 240, 1):
        try:
            rv = gpu.rrr_vector(sdsl_bytes=mut)
        except gpu.capi.SdslHipError:
            continue
        accepted += 1
        n = rv.size()
        idx = (np.arange(0, 64, dtype=np.uint64) * np.uint64(max(1, n // 64)))
        r = rv.rank(idx, 1)
        assert np.all((r <= np.uint64(n)) | (r == NPOS))
        ones = rv.ones()
        if ones:
            s = rv.select(np.arange(1, min(ones, 64) + 1, dtype=np.uint64), 1)
            assert np.all(s <= np.uint64(n))
    assert accepted < 240  # header / pointer damage must be noticed


def test_fuzz_wt_and_csa_streams(gpu):
    for name, rrr, mcl in (("example01.txt.wt_huff_v5_mcl.sdsl", False, True), ("example01.txt.wt_huff_rrr63.sdsl", True, True)):
        blob = gd.sdsl_file(name)
        for mut in _mutations(blob, 160, 2):
            try:
                wt = gpu.wt_huff(sdsl_bytes=mut, rrr=rrr, select_is_mcl=mcl)
            except gpu.capi.SdslHipError:
                continue
            n = wt.size()
            i = np.arange(0, 32, dtype=np.uint64) * np.uint64(max(1, n // 32))
            c = (np.arange(32) * 7 % 256).astype(np.uint8)
            r = wt.rank(i, c)
            assert np.all((r <= np.uint64(n)) | (r == NPOS))
            if n:
                wt.access(np.minimum(i, np.uint64(n - 1)))
    for name, rrr, mcl in (("example01.txt.csa_wt_huff_v5.sdsl", False, True), ("example01.txt.csa_wt_huff_rrr63.sdsl", True, True)):
        blob = gd.sdsl_file(name)
        for mut in _mutations(blob, 160, 3):
            try:
                csa = gpu.csa_wt(sdsl_bytes=mut, rrr=rrr, select_is_mcl=mcl, sa_dens=32, isa_dens=64)
            except gpu.capi.SdslHipError:
                continue
            N = csa.size()
            pats = np.frombuffer(b"the and of to a in is ", dtype=np.uint8)[:20]
            cnt = csa.count(np.tile(pats, 4), 4)
            assert np.all(cnt <= np.uint64(N))
            idx = np.arange(0, 16, dtype=np.uint64) * np.uint64(max(1, N // 16))
            sa = csa.sa(idx)  # a walk is bounded by N steps: corrupt samples give wrong values or NPOS, never a hang
            assert np.all((sa < np.uint64(N)) | (sa == NPOS))
            csa.extract(np.array([0], dtype=np.uint64), np.array([min(N - 1, 40)], dtype=np.uint64))


def test_fuzz_sd_stream(gpu):
    blob = gd.sdsl_file("rnd.8192.1043.sd_vector.sdsl")
    for mut in _mutations(blob, 240, 4):
        try:
            sd = gpu.sd_vector(sdsl_bytes=mut)
        except gpu.capi.SdslHipError:
            continue
        n, m = sd.size(), sd.ones()
        if n > (1 << 62):
            continue  # a damaged size word: nothing sensible to ask
        idx = np.arange(0, 64, dtype=np.uint64) * np.uint64(max(1, n // 64))
        r = sd.rank(idx, 1)
        assert np.all((r <= np.uint64(m)) | (r == NPOS))
        if m:
            sd.select(np.arange(1, min(m, 64) + 1, dtype=np.uint64), 1)
