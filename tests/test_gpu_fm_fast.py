"""count() of large fixed-length batches (fm_count2.hip): k-mer hash table -> flat search kernel -> text comparison.
Every combination of table depth (none, 1..8), suffix array kept / dropped, pattern length around the table depth and around the
16-byte pattern window, and the patterns that leave the common road: absent k-mers, absent characters further in, 0 bytes (the
sentinel's character), patterns at the very start of the text and of the batch.  References: the CPU oracle and the lock-step
kernel of fm.hip (through count_ragged, which never takes the new road)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def make_patterns(rng, text, sigma, m, npat):
    n = text.size
    st = rng.integers(0, n - m, npat)
    st[:300] = np.arange(300) % 11                     # cut at the very start of the text
    pats = text[st[:, None] + np.arange(m)[None, :]].copy()
    kind = rng.integers(0, 8, npat)
    where = rng.integers(0, m, npat)
    mut = kind == 0                                    # one character replaced by another symbol of the alphabet
    pats[mut, where[mut]] = rng.integers(1, sigma + 1, int(mut.sum()), dtype=np.uint8)
    absent = kind == 1                                 # ... by a byte that does not occur in the text
    pats[absent, where[absent]] = 255
    zero = kind == 2                                   # ... by the sentinel's byte
    pats[zero, where[zero]] = 0
    rnd = kind == 3                                    # random strings over the alphabet
    pats[rnd] = rng.integers(1, sigma + 1, (int(rnd.sum()), m), dtype=np.uint8)
    pats[300:330] = 0                                  # all sentinel bytes
    if m >= 2:
        pats[330:360, :-1] = text[n - (m - 1):]        # the text's last characters followed by the sentinel: occurs once
        pats[330:360, -1] = 0
    return np.ascontiguousarray(pats.reshape(-1))


@pytest.mark.parametrize("sigma,n", [(3, 40_000), (20, 150_000), (120, 300_000)])
def test_fast_count_equals_oracle_for_every_table_depth(gpu, sigma, n):
    rng = np.random.default_rng(sigma * 31 + 5)
    text = rng.integers(1, sigma + 1, n, dtype=np.uint8)
    text[2000:2400] = text[9000:9400]                  # long repeats: intervals stay wider than one suffix deep into a pattern
    text[20000:20064] = text[2100:2164]
    ocsa = ol.OCsa(bytes(text))
    csa = gpu.csa_wt(text=text)
    assert csa.kmer_table_depth() >= 1, "an index created from text carries the k-mer table"
    npat = 12_000
    want = {}
    for depth in (None, 0, 1, 2, 5, 8, "dropped"):
        if depth == "dropped":
            csa.set_kmer_table(6, 1 << 30)
            csa.drop_sa()                              # no text comparison any more; the table stays
            assert csa.kmer_table_depth() >= 1
        elif depth is not None:
            csa.set_kmer_table(depth, 1 << 30)
            assert csa.kmer_table_depth() == (min(depth, 8) if depth else 0)
        for m in (1, 2, 5, 8, 9, 20, 24, 25, 41):
            flat = make_patterns(np.random.default_rng(m), text, sigma, m, npat)
            if m not in want:
                want[m] = np.asarray(ocsa.count_batch(flat, m)).astype(np.uint64)
            got = np.asarray(csa.count(flat, m)).astype(np.uint64)
            bad = np.flatnonzero(got != want[m])
            assert bad.size == 0, (f"sigma {sigma}, table {depth}, m {m}: pattern {bad[0]} = "
                                   f"{bytes(flat[bad[0] * m:(bad[0] + 1) * m])!r} got {got[bad[0]]} want {want[m][bad[0]]}")
    csa.close()


def test_fast_count_equals_lock_step_kernel_on_a_larger_text(gpu):
    n = 24 << 20
    text = gpu.english_text(n, 77)
    csa = gpu.csa_wt(text=text)
    k = csa.kmer_table_depth()
    assert 3 <= k <= 8, k
    rng = np.random.default_rng(3)
    npat, m = 1_500_000, 20
    st = rng.integers(0, n - m, npat)
    pats = text[st[:, None] + np.arange(m)[None, :]].copy()
    mut = rng.random(npat) < 0.3
    pats[mut, rng.integers(0, m, int(mut.sum()))] = text[rng.integers(0, n, int(mut.sum()))]
    flat = np.ascontiguousarray(pats.reshape(-1))
    got = np.asarray(csa.count(flat, m)).astype(np.uint64)
    ref = np.asarray(csa.count_ragged([bytes(r) for r in pats[:200_000]])).astype(np.uint64)
    assert np.array_equal(got[:200_000], ref)
    assert int((got[~mut] >= 1).all()), "a pattern cut from the text occurs"
    csa.set_kmer_table(0, 0)                           # the flat kernel from the whole interval
    assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), got)
    csa.set_kmer_table(8, 64 << 30)                    # as deep as it gets
    assert csa.kmer_table_depth() == 8
    assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), got)
    csa.drop_sa()
    assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), got)
    csa.close()


def test_an_index_loaded_from_an_sdsl_stream_gets_text_suffix_array_and_table_back(gpu):
    rng = np.random.default_rng(11)
    n = 200_000
    text = rng.integers(1, 30, n, dtype=np.uint8)
    text[500:900] = text[7000:7400]
    built = gpu.csa_wt(text=text)
    blob = built.serialize(32, 64)
    m, npat = 12, 20_000
    st = rng.integers(0, n - m, npat)
    pats = text[st[:, None] + np.arange(m)[None, :]].copy()
    pats[::3, rng.integers(0, m)] = 7
    flat = np.ascontiguousarray(pats.reshape(-1))
    want = np.asarray(built.count(flat, m)).astype(np.uint64)
    built.close()
    loaded = gpu.csa_wt(sdsl_bytes=blob, select_is_mcl=False, sa_dens=32, isa_dens=64)
    assert loaded.kmer_table_depth() == 0 and not loaded.sampling()[2]
    assert np.array_equal(np.asarray(loaded.count(flat, m)).astype(np.uint64), want)  # dense table + flat kernel, every character walked
    loaded.restore_suffix_array()
    assert loaded.kmer_table_depth() >= 1 and loaded.sampling()[2], "text, suffix array and k-mer table are back"
    assert np.array_equal(np.asarray(loaded.count(flat, m)).astype(np.uint64), want)
    sidx = rng.integers(0, n + 1, 5000).astype(np.uint64)
    assert np.array_equal(np.asarray(loaded.sa(sidx)), np.asarray(ol.OCsa(bytes(text)).sa(sidx)))
    loaded.close()
    bare = gpu.csa_wt(sdsl_bytes=blob, select_is_mcl=False)  # no densities: nothing to read the text back with
    with pytest.raises(Exception):
        bare.restore_suffix_array()
    bare.close()
