"""GPU: the plain wavelet tree keeps two layouts in HBM — SDSL's binary levels (select, export) and the fused 16-ary
layout (four levels per fetch; three in a SDSL_HIP_FUSED_K=3 build: rank, backward search, inverse_select, LF walks).
SDSL_HIP_WT_FUSED=0 at creation time leaves the fused layout out, so every traversal takes the binary path; both must give
the oracle's answers on the same inputs, for trees whose depth is not a multiple of the fused step, with leaves at every
depth below it, and for every shape."""
import numpy as np
import pytest

import golden_data as gd
import oracle_lib as ol

pytestmark = pytest.mark.gpu

TEXTS = {
    "two_symbols": b"ab" * 300 + b"a" * 77,                                   # depth 1
    "dna": bytes(np.random.default_rng(1).choice(list(b"ACGT"), 5000)),       # depth 2-3 with the sentinel
    "skewed": b"a" * 4000 + b"b" * 500 + b"c" * 60 + b"d" * 7 + b"e",         # a chain: leaves at depth 1, 2, 3, 4
    "bytes256": bytes(np.random.default_rng(2).integers(1, 256, 30000, dtype=np.uint8)),  # depth 8-9
    "zipf": bytes((np.random.default_rng(3).zipf(1.3, 40000) % 200 + 1).astype(np.uint8)),  # deep Huffman tree
    "faust": None,
}


def _text(name):
    return gd.text("faust.txt") if name == "faust" else TEXTS[name]


@pytest.mark.parametrize("fused", ["1", "derived_shape", "0", "no_select_directory"])
@pytest.mark.parametrize("name", list(TEXTS))
def test_wavelet_tree_queries_on_both_layouts(gpu, monkeypatch, name, fused):
    """"1": fused layout with its own 16-ary Huffman shape (the default); "derived_shape": fused layout cut out of SDSL's
    binary tree; "0": binary levels only; "no_select_directory": fused layout, select on the binary levels"""
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "0" if fused == "0" else "1")
    monkeypatch.setenv("SDSL_HIP_WT_FUSED_SELECT", "0" if fused == "no_select_directory" else "1")
    monkeypatch.setenv("SDSL_HIP_WT_FUSED_SHAPE", "binary" if fused == "derived_shape" else "own")
    text = _text(name)
    wt = gpu.wt_huff(text=text)
    o = ol.OWt(text)
    n = len(text)
    rng = np.random.default_rng(7)
    arr = np.frombuffer(text, dtype=np.uint8)
    i = np.concatenate([rng.integers(0, n + 1, 4000), [0, 1, n - 1, n, 255, 256, 257, 511, 512]]).astype(np.uint64)
    i = np.minimum(i, np.uint64(n))
    c = np.concatenate([arr[rng.integers(0, n, 4000)], rng.integers(0, 256, 9).astype(np.uint8)])
    assert np.array_equal(wt.rank(i, c), o.rank(i, c))
    j = np.concatenate([rng.integers(0, n, 3000), [0, n - 1, min(255, n - 1), min(256, n - 1)]]).astype(np.uint64)
    r, ch = wt.inverse_select(j)
    orr, och = o.inverse_select(j)
    assert np.array_equal(r, orr) and np.array_equal(ch, och)
    assert np.array_equal(wt.access(j), arr[j.astype(np.int64)])
    # select: EVERY occurrence of every symbol that occurs (and the overflow / absent-symbol answers)
    order = np.argsort(arr, kind="stable")
    sym = arr[order]
    first = np.searchsorted(sym, sym, side="left")
    kk = (np.arange(n) - first + 1).astype(np.uint64)
    if n > 40000:
        pick = rng.choice(n, 40000, replace=False)
        order, sym, kk = order[pick], sym[pick], kk[pick]
    assert np.array_equal(wt.select(kk, sym), order.astype(np.uint64))
    absent = np.setdiff1d(np.arange(256), np.unique(arr))
    if absent.size:
        a = absent[:4].astype(np.uint8)
        assert np.all(wt.select(np.ones(a.size, dtype=np.uint64), a) == np.uint64(n))
    cnt = np.bincount(arr, minlength=256)
    present = np.unique(arr)[:8]
    assert np.all(wt.select(cnt[present].astype(np.uint64) + np.uint64(1), present) == np.uint64(2**64 - 1))


@pytest.mark.parametrize("fused", ["1", "derived_shape", "0"])
@pytest.mark.parametrize("kw", [{}, {"balanced": True}, {"hutu": True}])
def test_fm_index_queries_on_both_layouts(gpu, monkeypatch, kw, fused):
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "0" if fused == "0" else "1")
    monkeypatch.setenv("SDSL_HIP_WT_FUSED_SHAPE", "binary" if fused == "derived_shape" else "own")
    text = gd.text("faust.txt")
    csa = gpu.csa_wt(text=text, **kw)
    o = ol.OCsa(text)
    arr = np.frombuffer(text, dtype=np.uint8)
    rng = np.random.default_rng(9)
    for m in (1, 2, 3, 7, 20):
        st = rng.integers(0, len(text) - m, 1500)
        pats = np.concatenate([arr[s:s + m] for s in st] + [rng.integers(1, 256, m * 20).astype(np.uint8)])
        assert np.array_equal(csa.count(pats, m), o.count_batch(pats, m)), m
    idx = rng.integers(0, csa.size(), 3000).astype(np.uint64)
    assert np.array_equal(csa.lf(idx), o.lf(idx)) and np.array_equal(csa.psi(idx), o.psi(idx))
    assert np.array_equal(csa.psi(csa.lf(idx)), idx)
    csa.drop_sa()
    assert np.array_equal(csa.sa(idx), o.sa(idx)) and np.array_equal(csa.isa(idx), o.isa(idx))
    off, t = csa.extract(np.array([0, 1000, 200000], dtype=np.uint64), np.array([99, 1900, 200300], dtype=np.uint64))
    assert t.tobytes() == text[0:100] + text[1000:1901] + text[200000:200301]
    pats = np.frombuffer(b"und", dtype=np.uint8)
    off, pos = csa.locate(pats, 3)
    assert np.array_equal(pos, o.locate(b"und"))


def test_the_knob_decides_what_is_resident(gpu, monkeypatch):
    text = gd.text("faust.txt")
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "0")
    plain = gpu.wt_huff(text=text).device_bytes()
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "1")
    both = gpu.wt_huff(text=text).device_bytes()
    monkeypatch.delenv("SDSL_HIP_WT_FUSED")
    default = gpu.wt_huff(text=text).device_bytes()
    assert both > plain and default == both
    wt = gpu.wt_huff(text=text)
    steps, lens = wt.fused_steps(), wt.code_lengths()
    k = gpu.fused_geometry()["levels_per_fetch"]
    assert k in (3, 4)
    assert np.all((steps > 0) == (lens > 0)) and np.all(steps[lens > 0] <= (lens[lens > 0] + k - 1) // k + 1)
    cnt = np.bincount(np.frombuffer(text, dtype=np.uint8), minlength=256)
    # the layout's own shape is a 2^k-ary Huffman tree: never more expected steps than SDSL's tree cut into groups of k levels
    assert (cnt * steps).sum() <= (cnt * ((lens.astype(np.int64) + k - 1) // k)).sum()
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "0")
    assert not gpu.wt_huff(text=text).fused_steps().any()
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "1")
    # the compressed flavour has no fused layout
    with_knob = gpu.wt_huff(text=text, rrr=True).device_bytes()
    monkeypatch.setenv("SDSL_HIP_WT_FUSED", "0")
    assert gpu.wt_huff(text=text, rrr=True).device_bytes() == with_knob


@pytest.mark.parametrize("extra", [0, 1, -1])
def test_sequences_that_end_on_a_line_or_a_superblock(gpu, extra):
    """n a multiple of the positions of a line / of a superblock (and one more, one less), two symbols: the root node is the whole
    sequence, position n is addressable (rank(n, c)), the last line is empty or holds one position"""
    geo = gpu.fused_geometry()
    ppl, lps = geo["positions_per_line"], max(1, geo["lines_per_superblock"])
    rng = np.random.default_rng(11 + extra)
    for n in (ppl * 7 + extra, ppl * lps + extra, 2 * ppl * lps + extra):
        arr = rng.choice(np.array([65, 66], dtype=np.uint8), size=n, p=[0.7, 0.3])
        wt = gpu.wt_huff(text=arr.tobytes())
        i = np.unique(np.clip(np.concatenate([[0, 1, n - 1, n], np.arange(0, n + 1, ppl), np.arange(0, n + 1, ppl) - 1,
                                              rng.integers(0, n + 1, 5000)]), 0, n)).astype(np.uint64)
        for c in (65, 66, 67):
            pre = np.concatenate([[0], np.cumsum(arr == c)]).astype(np.uint64)
            assert np.array_equal(wt.rank(i, np.full(i.size, c, dtype=np.uint8)), pre[i.astype(np.int64)]), (n, c)
        for c in (65, 66):
            occ = np.flatnonzero(arr == c)
            kk = np.unique(np.concatenate([[1, occ.size], rng.integers(1, occ.size + 1, 3000)])).astype(np.uint64)
            assert np.array_equal(wt.select(kk, np.full(kk.size, c, dtype=np.uint8)), occ[kk.astype(np.int64) - 1].astype(np.uint64)), (n, c)
        j = i[i < n]
        assert np.array_equal(wt.access(j), arr[j.astype(np.int64)])
        wt.close()


@pytest.mark.parametrize("sigma,skew", [(3, 0.0), (20, 1.0), (200, 1.5)])
def test_lines_superblocks_and_node_starts(gpu, sigma, skew):
    """A sequence of several superblocks per fused node (16-ary lines: 1024 lines of 184 positions each), nodes that start in the
    middle of a superblock, positions on both sides of every line and superblock boundary of the root and around the places where
    the deeper nodes change superblocks: rank, inverse_select and select against prefix counts computed here."""
    geo = gpu.fused_geometry()
    ppl, lps = geo["positions_per_line"], max(1, geo["lines_per_superblock"])
    rng = np.random.default_rng(sigma)
    n = 3 * ppl * lps + 12345
    alphabet = np.sort(rng.choice(np.arange(1, 256), size=sigma, replace=False)).astype(np.uint8)
    w = np.arange(1, sigma + 1, dtype=np.float64) ** (-skew)
    arr = alphabet[rng.choice(sigma, size=n, p=w / w.sum())]
    wt = gpu.wt_huff(text=arr.tobytes())
    edges = np.concatenate([np.arange(0, n, ppl * lps), np.arange(0, n, ppl)[:: max(1, lps // 7)], [n]])
    i = np.unique(np.clip(np.concatenate([edges - 1, edges, edges + 1, rng.integers(0, n + 1, 20000)]), 0, n)).astype(np.uint64)
    # prefix counts per symbol at the probed positions
    for c in alphabet[:: max(1, sigma // 12)]:
        pre = np.concatenate([[0], np.cumsum(arr == c)]).astype(np.uint64)
        assert np.array_equal(wt.rank(i, np.full(i.size, c, dtype=np.uint8)), pre[i.astype(np.int64)]), int(c)
        occ = np.flatnonzero(arr == c)
        # select around the occurrences whose rank inside deeper nodes crosses lines / superblocks: every ppl-th and its neighbours
        kk = np.unique(np.clip(np.concatenate([np.arange(1, occ.size + 1, ppl), np.arange(1, occ.size + 1, ppl) + 1,
                                               np.arange(ppl, occ.size + 1, ppl * lps), [1, occ.size]]), 1, occ.size)).astype(np.uint64)
        assert np.array_equal(wt.select(kk, np.full(kk.size, c, dtype=np.uint8)), occ[kk.astype(np.int64) - 1].astype(np.uint64)), int(c)
    j = i[i < n]
    r, ch = wt.inverse_select(j)
    assert np.array_equal(ch, arr[j.astype(np.int64)])
    order = np.argsort(arr, kind="stable")
    rank_of = np.empty(n, dtype=np.int64)
    first = np.searchsorted(arr[order], arr[order], side="left")
    rank_of[order] = np.arange(n) - first
    assert np.array_equal(r, rank_of[j.astype(np.int64)].astype(np.uint64))


def test_loaded_streams_get_the_fused_layout_too(gpu):
    blob = gd.sdsl_file("example01.txt.wt_huff_v5_mcl.sdsl")
    text = gd.text("example01.txt")
    wt = gpu.wt_huff(sdsl_bytes=blob, select_is_mcl=True)
    assert wt.device_bytes() == gpu.wt_huff(text=text).device_bytes()
    o = ol.OWt(text)
    rng = np.random.default_rng(4)
    i = rng.integers(0, len(text) + 1, 5000).astype(np.uint64)
    c = np.frombuffer(text, dtype=np.uint8)[rng.integers(0, len(text), 5000)]
    assert np.array_equal(wt.rank(i, c), o.rank(i, c))


def test_both_layouts_agree_at_full_size(gpu, monkeypatch):
    """BASELINE.json's index size (1 GiB text): 10^7 random rank / count / LF queries answered on the fused layout and on
    the binary levels must be identical, and obey what the domain guarantees at any size."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", 0)
    nt = 1 << 30
    text = bench.synthetic_text(nt, 1234, dev)
    g = torch.Generator(device=dev).manual_seed(77)
    nq = 10_000_000
    i = torch.randint(0, nt + 2, (nq,), device=dev, dtype=torch.int64, generator=g)
    c = text[torch.randint(0, nt, (nq,), device=dev, generator=g)]
    m = 20
    st = torch.randint(0, nt - m, (2_000_000,), device=dev, generator=g)
    pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
    idx = torch.randint(0, nt + 1, (2_000_000,), device=dev, dtype=torch.int64, generator=g)
    got = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SDSL_HIP_WT_FUSED", fused)
        csa = gpu.csa_wt(text=text)
        wt = csa.wavelet_tree
        r = wt.rank(i, c)
        cnt = csa.count(pats, m)
        lf = csa.lf(idx)
        ir, ic = wt.inverse_select(idx)
        got[fused] = (r, cnt, lf, ir, ic)
        if fused == "1":
            # domain properties: every pattern was cut from the text; LF is a permutation step consistent with
            # inverse_select + C; rank(n + 1, c) over all symbols sums to n + 1
            assert bool((cnt >= 1).all())
            tot = wt.rank(torch.full((256,), nt + 1, device=dev, dtype=torch.int64),
                          torch.arange(256, device=dev, dtype=torch.uint8))
            assert int(tot.sum()) == nt + 1
            assert torch.equal(csa.psi(lf[:200_000]), idx[:200_000])
        csa.close()
        del csa, wt
        torch.cuda.empty_cache()
    for a, b in zip(got["1"], got["0"]):
        assert torch.equal(a, b)
