"""GPU: a seeded campaign of random wavelet trees / FM-indexes against the oracle.  Alphabet sizes sit around the
boundaries of the fused layout's 8-ary tree (1 + 7k leaves fill it exactly; 8, 9, 15, 16, 57, 64, 65 ... do not), symbol
frequencies go from uniform to steeply skewed (deep Huffman paths), lengths from a handful of symbols to a few lines of
the fused layout — every query type on every tree."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
NPOS = np.uint64(2**64 - 1)

SIGMAS = [2, 3, 7, 8, 9, 15, 16, 17, 50, 57, 58, 64, 65, 128, 200, 255]


def _random_text(rng, sigma, n, skew):
    alphabet = rng.choice(np.arange(1, 256), size=sigma, replace=False).astype(np.uint8)
    w = np.arange(1, sigma + 1, dtype=np.float64) ** (-skew)
    t = alphabet[rng.choice(sigma, size=n, p=w / w.sum())]
    t[rng.choice(n, size=min(n, sigma), replace=False)] = alphabet[: min(n, sigma)]  # every symbol at least once (n >= sigma)
    return t.tobytes()


@pytest.mark.parametrize("sigma", SIGMAS)
@pytest.mark.parametrize("skew", [0.0, 1.0, 3.0])
def test_random_wavelet_trees(gpu, sigma, skew):
    rng = np.random.default_rng(1000 * sigma + int(10 * skew))
    n = int(rng.integers(max(sigma, 300), 6000))
    text = _random_text(rng, sigma, n, skew)
    arr = np.frombuffer(text, dtype=np.uint8)
    wt = gpu.wt_huff(text=text)
    o = ol.OWt(text)
    i = rng.integers(0, n + 1, 3000).astype(np.uint64)
    c = np.concatenate([arr[rng.integers(0, n, 2900)], rng.integers(0, 256, 100).astype(np.uint8)])
    assert np.array_equal(wt.rank(i, c), o.rank(i, c))
    j = rng.integers(0, n, 2000).astype(np.uint64)
    r, ch = wt.inverse_select(j)
    orr, och = o.inverse_select(j)
    assert np.array_equal(r, orr) and np.array_equal(ch, och)
    order = np.argsort(arr, kind="stable")
    sym = arr[order]
    kk = (np.arange(n) - np.searchsorted(sym, sym, side="left") + 1).astype(np.uint64)
    assert np.array_equal(wt.select(kk, sym), order.astype(np.uint64))  # every occurrence of every symbol
    steps, lens = wt.fused_steps(), wt.code_lengths()
    assert np.all((steps > 0) == (lens > 0))
    assert wt.serialize() == gpu.wt_huff(sdsl_bytes=wt.serialize(), select_is_mcl=False).serialize()


@pytest.mark.parametrize("sigma", [2, 5, 9, 16, 40, 64, 120])
def test_random_fm_indexes(gpu, sigma):
    rng = np.random.default_rng(77 + sigma)
    n = int(rng.integers(500, 4000))
    text = _random_text(rng, sigma, n, 1.2)
    arr = np.frombuffer(text, dtype=np.uint8)
    csa = gpu.csa_wt(text=text)
    o = ol.OCsa(text)
    N = csa.size()
    idx = np.arange(N, dtype=np.uint64)
    for m in (1, 2, 4, 9):
        st = rng.integers(0, n - m, 400)
        pats = np.concatenate([arr[s:s + m] for s in st] + [rng.integers(1, 256, m * 30).astype(np.uint8)])
        assert np.array_equal(csa.count(pats, m), o.count_batch(pats, m)), m
    assert np.array_equal(csa.lf(idx), o.lf(idx)) and np.array_equal(csa.psi(idx), o.psi(idx))
    assert np.array_equal(csa.sa(idx), o.sa(idx))
    csa.drop_sa()
    assert np.array_equal(csa.sa(idx), o.sa(idx)) and np.array_equal(csa.isa(idx), o.isa(idx))
    off, t = csa.extract(np.array([0], dtype=np.uint64), np.array([N - 1], dtype=np.uint64))
    assert t.tobytes() == text + b"\x00"
    p3 = arr[5:8]
    off, pos = csa.locate(p3, 3)
    assert np.array_equal(pos, o.locate(p3.tobytes()))
