"""GPU: degenerate inputs of the wider API — empty batches, empty and one-symbol structures, patterns longer than the
text, all-zero / all-one sparse vectors — against the oracle or first principles."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
NPOS = np.uint64(2**64 - 1)
E = np.zeros(0, dtype=np.uint64)


def test_empty_batches_everywhere(gpu):
    text = b"abracadabra" * 20
    csa = gpu.csa_wt(text=text)
    assert csa.sa(E).size == 0 and csa.isa(E).size == 0 and csa.lf(E).size == 0 and csa.psi(E).size == 0
    off, t = csa.extract(E, E)
    assert list(off) == [0] and t.size == 0
    off, p = csa.locate(np.zeros(0, np.uint8), 3)
    assert list(off) == [0] and p.size == 0
    assert csa.count(np.zeros(0, np.uint8), 4).size == 0
    wt = csa.wavelet_tree
    assert wt.rank(E, np.zeros(0, np.uint8)).size == 0 and wt.select(E, np.zeros(0, np.uint8)).size == 0
    bv = gpu.bit_vector(gpu.set_random_bits(1000, 1), 1000)
    assert bv.rank(E).size == 0 and bv.select(E).size == 0 and bv.access(E).size == 0
    sd = gpu.sd_vector(gpu.set_random_bits(1000, 1), 1000)
    assert sd.rank(E).size == 0 and sd.select(E).size == 0 and sd.access(E).size == 0


@pytest.mark.parametrize("text", [b"a", b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", b"ab", b"abababababababababababab", b"z" * 5000])
@pytest.mark.parametrize("rrr", [False, True])
def test_tiny_and_unary_texts(gpu, text, rrr):
    """sigma 2 (one symbol + sentinel) and 3: the jump table covers patterns containing the sentinel byte, trees of one or
    two levels, SA / ISA / extract over the whole index"""
    csa = gpu.csa_wt(text=text, rrr=rrr)
    o = ol.OCsa(text)
    N = csa.size()
    assert N == len(text) + 1 and csa.sigma() == o.sigma()
    idx = np.arange(N, dtype=np.uint64)
    for k in (None, 0, 1, 3):
        if k is not None:
            csa.set_jump_depth(min(k, 3))
        for pat in (b"a", b"b", b"aa", b"ab", b"ba", b"aaa", b"abab", text, text + b"a", b"\x00", b"a\x00", b""):
            assert int(csa.count_ragged([pat])[0]) == o.count(pat), (pat, k)
    assert np.array_equal(csa.sa(idx), o.sa(idx)) and np.array_equal(csa.isa(idx), o.isa(idx))
    assert np.array_equal(csa.lf(idx), o.lf(idx)) and np.array_equal(csa.psi(idx), o.psi(idx))
    off, t = csa.extract(np.array([0], dtype=np.uint64), np.array([N - 1], dtype=np.uint64))
    assert t.tobytes() == text + b"\x00"
    csa.drop_sa()
    assert np.array_equal(csa.sa(idx), o.sa(idx))
    m = min(2, len(text))
    pats = np.frombuffer(text[:m] * 3, dtype=np.uint8)
    off, pos = csa.locate(pats, m)
    assert np.array_equal(pos[: int(off[1])], o.locate(text[:m]))


def test_pattern_longer_than_text_and_absent_symbols(gpu):
    text = b"mississippi"
    csa = gpu.csa_wt(text=text)
    long = np.frombuffer(b"mississippimississippi", dtype=np.uint8)
    assert int(csa.count(long, long.size)[0]) == 0
    off, pos = csa.locate(long, long.size)
    assert list(off) == [0, 0] and pos.size == 0
    l, r = csa.interval(np.frombuffer(b"ssix", dtype=np.uint8), 4)
    ol_l, ol_r = ol.OCsa(text).interval(b"ssix")
    assert (int(l[0]), int(r[0])) == (ol_l, ol_r)  # the empty interval as the reference leaves it


@pytest.mark.parametrize("n,ones", [(0, 0), (1, 0), (1, 1), (64, 0), (64, 64), (1000, 0), (1000, 1000), (100000, 1), (4096, 4096)])
def test_sd_and_pattern_vectors_at_the_extremes(gpu, n, ones):
    bits = np.zeros(n, dtype=np.uint8)
    bits[:ones] = 1
    if 0 < ones < n:
        bits = np.roll(bits, n // 2)
    words = np.packbits(np.concatenate([bits, np.zeros((-n) % 64 + 64, np.uint8)]), bitorder="little").view(np.uint64)
    sd = gpu.sd_vector(words, n)
    assert (sd.size(), sd.ones()) == (n, ones)
    idx = np.arange(0, n + 1, max(1, n // 50), dtype=np.uint64)
    cum = np.concatenate([[0], np.cumsum(bits)]).astype(np.uint64)
    assert np.array_equal(sd.rank(idx, 1), cum[idx.astype(np.int64)])
    assert np.array_equal(sd.rank(idx, 0), idx - cum[idx.astype(np.int64)])
    if ones:
        i = np.arange(1, ones + 1, max(1, ones // 50), dtype=np.uint64)
        assert np.array_equal(sd.select(i, 1), np.flatnonzero(bits)[(i - np.uint64(1)).astype(np.int64)].astype(np.uint64))
    if n - ones:
        i = np.arange(1, n - ones + 1, max(1, (n - ones) // 20), dtype=np.uint64)
        assert np.array_equal(sd.select(i, 0), np.flatnonzero(bits == 0)[(i - np.uint64(1)).astype(np.int64)].astype(np.uint64))
    if ol.have_ref():
        assert sd.serialize() == ol.RSd(words, n).serialize()
    for pat, targ in enumerate([(10, 2), (1, 2), (0, 2), (11, 2)]):
        pv = gpu.bit_vector(words, n, pattern=targ)
        d = ol.pattern_bits(bits, pat)
        c = np.concatenate([[0], np.cumsum(d)]).astype(np.uint64)
        assert np.array_equal(pv.rank(idx, 1), c[idx.astype(np.int64)])


def test_single_symbol_trees_of_every_shape(gpu):
    data = b"q" * 777
    for kw in ({}, {"balanced": True}, {"hutu": True}, {"rrr": True}, {"hutu": True, "rrr": True}):
        wt = gpu.wt_huff(data, **kw)
        assert (wt.size(), wt.sigma(), wt.bv_size()) == (777, 1, 0)
        i = np.array([0, 1, 500, 777], dtype=np.uint64)
        assert list(wt.rank(i, np.full(4, ord("q"), np.uint8))) == [0, 1, 500, 777]
        assert list(wt.rank(i, np.full(4, ord("x"), np.uint8))) == [0, 0, 0, 0]
        assert list(wt.select(np.array([1, 777], dtype=np.uint64), np.full(2, ord("q"), np.uint8))) == [0, 776]
        assert list(wt.access(np.array([0, 776], dtype=np.uint64))) == [ord("q")] * 2


@pytest.mark.parametrize("shape", ["cluster_front", "cluster_back", "dense_half", "two_clusters", "every_4096th", "all_but_one"])
def test_sd_select0_on_skewed_vectors(gpu, shape):
    """select_0 interpolates over buckets (sd.hip): vectors whose zeros are anything but evenly spread must still give the
    position a scan gives — every zero of the vector is asked for"""
    n = 50_000 + 37
    bits = np.zeros(n, dtype=np.uint8)
    if shape == "cluster_front":
        bits[:20_000] = 1
    elif shape == "cluster_back":
        bits[-20_000:] = 1
    elif shape == "dense_half":
        bits[np.random.default_rng(1).random(n) < 0.5] = 1
    elif shape == "two_clusters":
        bits[1000:9000] = 1
        bits[30_000:30_700] = 1
    elif shape == "every_4096th":
        bits[::4096] = 1
    else:
        bits[:] = 1
        bits[12345] = 0
    pos = np.flatnonzero(bits).astype(np.uint64)
    sd = gpu.sd_vector(positions=pos, n_bits=n)
    zeros = np.flatnonzero(bits == 0).astype(np.uint64)
    i = np.arange(1, zeros.size + 1, dtype=np.uint64)
    assert np.array_equal(sd.select(i, 0), zeros)
    assert np.all(sd.select(np.array([0, zeros.size + 1], dtype=np.uint64), 0) == NPOS)
    ones_i = np.arange(1, pos.size + 1, dtype=np.uint64)
    assert np.array_equal(sd.select(ones_i, 1), pos)


@pytest.mark.parametrize("rrr", [False, True])
def test_patterns_that_end_with_the_sentinel_on_texts_of_a_few_suffixes(gpu, rrr):
    """count() hands a search over to the text when the interval is down to a few suffixes (round 6: up to eight) — but the sentinel byte 0 is
    in the index and not in the text buffer, so a pattern that ENDS with it must go through the index for at least one character.  On a text
    of fewer than eight suffixes the WHOLE interval is that narrow before anything has been matched: large batches (the flat kernels) and
    small ones (the lock-step kernel), patterns with the sentinel at the end / in the middle / nowhere, against the oracle
    (suffix_array_algorithm.hpp:464-471)."""
    for text in (b"a", b"ab", b"abcab", b"aaaaaa", b"abababa"):
        csa = gpu.csa_wt(text=text, rrr=rrr)
        o = ol.OCsa(text)
        alphabet = sorted(set(text)) + [0]
        for m in (2, 3, 4):
            rng = np.random.default_rng(m + len(text))
            pats = np.array(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), (6000, m))]
            pats[:len(text), :] = 0
            for i in range(min(len(text), 6000)):                    # every suffix of the text followed by the sentinel, cut to m bytes
                s = (text[i:] + b"\x00")[-m:]
                pats[i, m - len(s):] = np.frombuffer(s, dtype=np.uint8)
                pats[i, :m - len(s)] = text[0]
            want = np.array([o.count(bytes(r)) for r in pats], dtype=np.uint64)
            flat = np.ascontiguousarray(pats.reshape(-1))
            assert np.array_equal(np.asarray(csa.count(flat, m)).astype(np.uint64), want), (text, m, "large batch")
            assert np.array_equal(np.asarray(csa.count(flat[: 50 * m], m)).astype(np.uint64), want[:50]), (text, m, "small batch")
            if m <= len(text) + 1:
                assert want.max() >= 1, "the text's last m - 1 bytes + the sentinel occur"
        csa.close()
