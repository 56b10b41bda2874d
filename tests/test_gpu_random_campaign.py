"""GPU: a seeded campaign of random wavelet trees / FM-indexes against the oracle.  Alphabet sizes sit around the
boundaries of the fused layout's 16-ary tree (1 + 15k leaves fill it exactly: 16, 31, 46, 61, 241, 256; their neighbours do
not) and of its 8-ary form (1 + 7k), symbol frequencies go from uniform to steeply skewed (deep Huffman paths), lengths
from a handful of symbols to a few lines of the fused layout — every query type on every tree."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
NPOS = np.uint64(2**64 - 1)

SIGMAS = [2, 3, 7, 8, 9, 15, 16, 17, 30, 31, 32, 46, 50, 57, 58, 61, 64, 65, 128, 200, 240, 241, 242, 255]


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


@pytest.mark.parametrize("seed", range(64))
def test_random_walks_through_the_states_of_an_index(gpu, seed):
    """The state machine round 5 added to csa_wt — set_footprint (binary levels released, suffix array and text -> packed samples, k-mer
    table resized), drop_sa at the caller's densities, restore_suffix_array, set_kmer_table / set_jump_depth — driven by a seeded random
    sequence of calls on a random text; after every call (accepted or refused) count of large and small batches, ragged count, intervals,
    csa[i], isa[i], psi, lf, extract, locate and the serialised stream are what the oracle / the first state gave."""
    rng = np.random.default_rng(4242 + seed)
    sigma = int(rng.choice([2, 3, 8, 9, 30, 64, 200]))
    n = int(rng.integers(3000, 60_000))
    text = _random_text(rng, sigma, n, float(rng.choice([0.0, 1.0, 2.5])))
    arr = np.frombuffer(text, dtype=np.uint8)
    o = ol.OCsa(text)
    csa = gpu.csa_wt(text=text)
    N = csa.size()
    m = int(rng.choice([3, 8, 20]))
    npat = 6000
    st = rng.integers(0, n - m, npat)
    pats = arr[st[:, None] + np.arange(m)[None, :]].copy()
    mut = rng.random(npat) < 0.3
    pats[mut, rng.integers(0, m, int(mut.sum()))] = rng.integers(1, 256, int(mut.sum())).astype(np.uint8)
    flat = np.ascontiguousarray(pats.reshape(-1))
    want_cnt = np.asarray(o.count_batch(flat, m)).astype(np.uint64)
    idx = rng.integers(0, N, 1500).astype(np.uint64)
    want = {"sa": np.asarray(o.sa(idx)), "isa": np.asarray(o.isa(idx)), "lf": np.asarray(o.lf(idx)), "psi": np.asarray(o.psi(idx))}
    blob = csa.serialize(32, 64)
    lo0, hi0 = (np.asarray(a) for a in csa.interval(flat[: 500 * m], m))
    dens = [32, 64]

    def check(what):
        assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), want_cnt), what
        assert np.array_equal(np.asarray(csa.count(flat[: 50 * m], m)).astype(np.uint64), want_cnt[:50]), what
        assert np.array_equal(np.asarray(csa.count_ragged([bytes(r) for r in pats[:40]])).astype(np.uint64), want_cnt[:40]), what
        lo, hi = csa.interval(flat[: 500 * m], m)
        assert np.array_equal(np.asarray(lo), lo0) and np.array_equal(np.asarray(hi), hi0), what
        for k, fn in (("sa", csa.sa), ("isa", csa.isa), ("lf", csa.lf), ("psi", csa.psi)):
            assert np.array_equal(np.asarray(fn(idx)), want[k]), (what, k)
        b = rng.integers(0, n - 40, 20).astype(np.uint64)
        off, t = csa.extract(b, b + np.uint64(39))
        t = np.asarray(t)
        assert all(bytes(t[i * 40:(i + 1) * 40]) == text[int(b[i]):int(b[i]) + 40] for i in range(20)), what
        q = int(np.flatnonzero(want_cnt > 0)[0])
        off, pos = csa.locate(np.ascontiguousarray(pats[q]), m)
        assert np.array_equal(np.sort(np.asarray(pos)), np.sort(o.locate(bytes(pats[q])))), what
        has_sa = csa.sampling()[2]
        if has_sa or tuple(csa.sampling()[:2]) == (32, 64):
            assert csa.serialize(32, 64) == blob, what

    check("as created")
    for step in range(7):
        op = int(rng.integers(0, 6))
        total = csa.device_bytes()
        try:
            if op == 0:
                csa.set_footprint(int(total * rng.uniform(0.15, 1.05)))
                what = "set_footprint"
            elif op == 1:
                d = (int(rng.choice([4, 8, 32, 64])), int(rng.choice([8, 16, 64])))
                csa.drop_sa(*d)
                dens[:] = d
                what = f"drop_sa{d}"
            elif op == 2:
                csa.restore_suffix_array()
                what = "restore_suffix_array"
            elif op == 3:
                csa.set_kmer_table(int(rng.integers(0, 9)), int(rng.integers(1, 1 << 24)))
                what = "set_kmer_table"
            elif op == 4:
                csa.set_jump_depth(int(rng.integers(0, 3)))
                what = "set_jump_depth"
            else:
                csa.drop_sa()
                what = "drop_sa()"
        except gpu.capi.SdslHipError as e:
            assert e.status in (gpu.capi.ERR_INVALID, gpu.capi.ERR_UNSUPPORTED), (op, e)
            what = f"refused op {op}: {e}"
        check(f"seed {seed}, step {step}: {what}")
    csa.close()
