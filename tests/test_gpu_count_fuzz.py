"""GPU: a seeded fuzz of count() around the point where a search is handed over to the text (round 6: at up to eight suffixes).  Texts with
duplicated passages, tiny alphabets (wide intervals for long) and tiny texts (the whole index a few suffixes); patterns cut from the text,
mutated, longer than the text, with the sentinel byte 0 at the end / inside; large batches (flat kernels, fm_count2.hip), small ones (the
lock-step kernel, fm.hip), the rrr index (wt_rrr.hip); with suffix array and text resident and dropped — every answer against the oracle's
backward search (suffix_array_algorithm.hpp:464-471).  SDSL_HIP_FUZZ_CASES=<n> runs more cases than the default."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("SDSL_HIP_FUZZ_CASES", "24"))


def make_text(rng):
    kind = rng.integers(0, 5)
    if kind == 0:                                   # tiny
        n = int(rng.integers(1, 12))
        return rng.integers(1, 4, n, dtype=np.uint8)
    sigma = int(rng.choice([2, 3, 4, 20, 90, 250]))
    n = int(rng.integers(200, 30_000))
    t = rng.integers(1, sigma + 1, n, dtype=np.uint8)
    if kind >= 2:                                   # duplicated passages, some of them many times
        for _ in range(int(rng.integers(1, 30))):
            ln = int(rng.integers(5, max(6, n // 8)))
            a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
            t[b:b + ln] = t[a:a + ln]
    if kind == 4:                                   # a run
        a = int(rng.integers(0, n // 2))
        t[a:a + n // 4] = t[a]
    return t


@pytest.mark.parametrize("case", range(CASES))
def test_count_against_the_oracle(gpu, case):
    rng = np.random.default_rng(9000 + case)
    text = make_text(rng)
    n = text.size
    o = ol.OCsa(bytes(text))
    rrr = bool(case % 3 == 2)
    csa = gpu.csa_wt(text=text, rrr=rrr)
    for m in sorted(set(int(x) for x in (1, 2, 3, rng.integers(4, 9), rng.integers(9, 20), 20, rng.integers(21, 40)))):
        npat = 5000
        st = rng.integers(0, max(1, n - m + 1), npat) if n >= m else np.zeros(npat, dtype=np.int64)
        idx = (st[:, None] + np.arange(m)[None, :]) % n          # (wraps for m > n: patterns longer than the text)
        pats = text[idx].copy()
        k = rng.random(npat)
        mut = k < 0.25
        pats[mut, rng.integers(0, m, int(mut.sum()))] = rng.integers(1, 255, int(mut.sum()), dtype=np.uint8)
        z_end = (k >= 0.25) & (k < 0.32)                          # the text's tail + the sentinel
        for q in np.flatnonzero(z_end):
            tail = np.concatenate([text[max(0, n - (m - 1)):], np.zeros(1, dtype=np.uint8)])[-m:]
            pats[q, m - tail.size:] = tail
        z_in = (k >= 0.32) & (k < 0.37)
        pats[z_in, rng.integers(0, m, int(z_in.sum()))] = 0
        flat = np.ascontiguousarray(pats.reshape(-1))
        want = o.count_batch(flat, m)
        got = np.asarray(csa.count(flat, m)).astype(np.uint64)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"case {case} rrr {rrr} m {m} large batch: {bytes(pats[bad[0]])!r} got {got[bad[0]]} want {want[bad[0]]} (n {n})"
        got = np.asarray(csa.count(flat[: 60 * m], m)).astype(np.uint64)
        assert np.array_equal(got, want[:60]), f"case {case} rrr {rrr} m {m} small batch"
    if n > 40:
        csa.drop_sa()                                             # no text to hand over to: every character an LF step
        m = 12
        st = rng.integers(0, n - m, 5000)
        pats = text[st[:, None] + np.arange(m)[None, :]].copy()
        pats[::5, 3] = 0
        flat = np.ascontiguousarray(pats.reshape(-1))
        assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), o.count_batch(flat, m)), f"case {case}: samples only"
    csa.close()
