"""wt.select of a large batch through the bucketed passes (wt_sorted.hip) against the direct kernel: uniform and heavily skewed
alphabets, queries outside select's precondition, absent symbols, batches concentrated on one symbol / one stretch of ranks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both_ways(gpu, wt, k, c):
    gpu.set_option("wt_select_sorted", 0)
    want = np.asarray(wt.select(k, c)).astype(np.uint64)
    gpu.set_option("wt_select_sorted", 1)
    try:
        got = np.asarray(wt.select(k, c)).astype(np.uint64)
    finally:
        gpu.set_option("wt_select_sorted", -1)
    return got, want


@pytest.mark.parametrize("kind", ["english", "skewed", "binary", "uniform200"])
def test_bucketed_wt_select_equals_the_direct_kernel(gpu, kind):
    rng = np.random.default_rng(len(kind))
    n = 6 << 20
    if kind == "english":
        text = gpu.english_text(n, 5)
    elif kind == "skewed":
        text = np.where(rng.random(n) < 0.97, 65, rng.integers(1, 120, n)).astype(np.uint8)
    elif kind == "binary":
        text = rng.integers(1, 3, n).astype(np.uint8)
    else:
        text = rng.integers(1, 201, n).astype(np.uint8)
    wt = gpu.wt_huff(text=text)
    occ = np.bincount(text, minlength=256).astype(np.uint64)
    nq = 1_200_000
    c = text[rng.integers(0, n, nq)].copy()
    k = (1 + rng.integers(0, 1 << 62, nq).astype(np.uint64) % occ[c]).astype(np.uint64)
    # outside the precondition / absent symbols / the extremes of every symbol
    k[:1000] = 0
    k[1000:2000] = occ[c[1000:2000]] + np.uint64(1)
    c[2000:3000] = 255 if occ[255] == 0 else c[2000:3000]
    k[3000:4000] = 1
    k[4000:5000] = occ[c[4000:5000]]
    # a stretch of the batch on ONE symbol and a narrow range of ranks: one bucket gets tens of thousands of keys
    top = int(np.argmax(occ))
    c[10_000:110_000] = top
    k[10_000:110_000] = 1 + rng.integers(0, min(int(occ[top]), 5000), 100_000).astype(np.uint64)
    got, want = both_ways(gpu, wt, k, c)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{kind}: query {bad[0]} (k {k[bad[0]]}, c {c[bad[0]]}): {got[bad[0]]} != {want[bad[0]]}"
    ok = (k >= 1) & (k <= occ[c]) & (occ[c] > 0)
    pos = got[ok]
    assert np.array_equal(text[pos.astype(np.int64)], c[ok]), "the symbol at select(k, c) is c"
    assert np.array_equal(np.asarray(wt.rank(pos, c[ok])).astype(np.uint64), k[ok] - np.uint64(1)), "rank(select(k, c), c) == k - 1"
    wt.close()
