"""CPU: the C restatement (oracle/) against the committed golden vectors produced by the real
sdsl-lite — answers AND serialised bytes (sha256 of SDSL's own streams).  This is what pins the
oracle; the GPU parity tests then compare the HIP path with the oracle and with the same vectors."""
import hashlib
import os

import numpy as np
import pytest

import golden_data as gd
import oracle_lib as ol


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("name", gd.bv_case_names())
def test_bitvector_rank_select_rrr(name):
    g = gd.bv_golden()
    w, n = gd.bv_case(name)
    assert int(g[f"{name}/n"][0]) == n
    o = ol.OBitVector(w, n)
    r = ol.ORrr(w, n)
    idx = g[f"{name}/idx"]
    for b in (0, 1):
        assert np.array_equal(o.rank(idx, b), g[f"{name}/rank{b}"])
        assert np.array_equal(r.rank(idx, b), g[f"{name}/rank{b}"])
        assert o.arg_cnt(b) == int(g[f"{name}/total{b}"][0])
        if o.arg_cnt(b):
            assert np.array_equal(o.select(g[f"{name}/sel{b}_i"], b), g[f"{name}/sel{b}"])
        assert np.array_equal(r.select(g[f"{name}/rrr_sel{b}_i"], b), g[f"{name}/rrr_sel{b}"])
    shas = list(g[f"{name}/sha"])
    assert [sha(o.serialize_rank(1)), sha(o.serialize_rank(0)), sha(o.serialize_select(1)),
            sha(o.serialize_select(0)), sha(r.serialize())] == shas


def test_known_answers_from_baseline_md():
    # BASELINE.md §2 / SURVEY.md §8(c)
    w = ol.set_random_bits(1 << 20, 815)
    o = ol.OBitVector(w, 1 << 20)
    assert int(o.rank([1 << 20])[0]) == 524053
    assert int(o.select([1000])[0]) == 1961
    r = ol.ORrr(w, 1 << 20)
    assert int(r.rank([1 << 20])[0]) == 524053 and int(r.select([1000])[0]) == 1961
    w, n = gd.bv_case("CRAFTED-MAT-SELECT")
    assert n == 1000000 and int(ol.OBitVector(w, n).rank([n])[0]) == 4061


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 383, 384, 385, 2047, 2048, 2049, 4096])
def test_rank_v5_edges_with_stray_bits(n):
    # every idx in [0,n] on vectors whose last word has stray bits above n;
    # serialised directory size = 8 + 8*(2*((n+63)>>11)+2)
    w = ol.mt19937_64((n + 63) // 64, 99)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(np.uint64)
    pref = np.concatenate([[0], np.cumsum(bits)]).astype(np.uint64)
    o = ol.OBitVector(w, n)
    idx = np.arange(n + 1, dtype=np.uint64)
    assert np.array_equal(o.rank(idx, 1), pref)
    assert np.array_equal(o.rank(idx, 0), idx - pref)
    assert len(o.serialize_rank(1)) == 8 + 8 * (2 * ((n + 63) >> 11) + 2)


def test_select0_padding_clamp():
    # select_support_mcl.hpp:299-300: 100001-bit vector with one set bit, select_0(100000) = 100000
    n = 100001
    w = np.zeros((n + 63) // 64, dtype=np.uint64)
    w[0] = 1
    o = ol.OBitVector(w, n)
    assert int(o.select([100000], 0)[0]) == 100000


def test_rrr_dummy_block_and_overflow():
    n = 630  # 63 | n -> dummy block, bt.size() == 11 (rrr_vector.hpp:163)
    w = ol.mt19937_64((n + 63) // 64, 5)
    r = ol.ORrr(w, n)
    ser = r.serialize()
    bt_hdr = int.from_bytes(ser[8:16], "little")
    assert bt_hdr >> 56 == 6 and (bt_hdr & ((1 << 56) - 1)) == 11 * 6
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n]
    ones = int(bits.sum())
    assert int(r.rank([n])[0]) == ones
    assert int(r.select([ones + 1])[0]) == n


@pytest.mark.parametrize("name", gd.TEXTS)
def test_wavelet_tree(name):
    g = gd.text_golden()
    data = gd.text(name)
    wt = ol.OWt(data)
    n, sigma, bvs = (int(x) for x in g[f"{name}/meta"])
    assert (wt.size(), wt.sigma(), wt.bv_size()) == (n, sigma, bvs)
    assert np.array_equal(wt.rank(g[f"{name}/rank_i"], g[f"{name}/rank_c"]), g[f"{name}/rank"])
    assert np.array_equal(wt.rank(np.full(256, n, dtype=np.uint64), np.arange(256, dtype=np.uint8)),
                          g[f"{name}/rank_full"])
    if n:
        ai = g[f"{name}/acc_i"]
        assert np.array_equal(wt.access(ai), g[f"{name}/acc"])
        r, c = wt.inverse_select(ai)
        assert np.array_equal(r, g[f"{name}/invsel_rank"]) and np.array_equal(c, g[f"{name}/acc"])
        assert np.array_equal(wt.select(g[f"{name}/sel_i"], g[f"{name}/sel_c"]), g[f"{name}/sel"])
        # serialised wt_huff streams are byte-identical to SDSL's (both select flavours)
        assert [sha(wt.serialize(1)), sha(wt.serialize(0))] == list(g[f"{name}/sha"])


def test_wt_reference_test_semantics():
    # test/wt_byte_test.cpp:134-168: rank(j+1, text[j]) equals the running count for every position
    data = gd.text("example01.txt")
    arr = np.frombuffer(data, dtype=np.uint8)
    wt = ol.OWt(data)
    cnt = np.zeros(256, dtype=np.uint64)
    exp = np.empty(arr.size, dtype=np.uint64)
    for j, c in enumerate(arr):
        cnt[c] += 1
        exp[j] = cnt[c]
    assert np.array_equal(wt.rank(np.arange(1, arr.size + 1, dtype=np.uint64), arr), exp)


@pytest.mark.parametrize("name", [t for t in gd.TEXTS if t not in ("empty.txt", "all_symbols.txt")])
def test_fm_index(name):
    g = gd.text_golden()
    data = gd.text(name)
    if f"{name}/csa_meta" not in g.files:
        pytest.skip("text contains a 0 byte: not indexable (construct.hpp:41)")
    csa = ol.OCsa(data)
    assert [csa.size(), csa.sigma()] == [int(x) for x in g[f"{name}/csa_meta"]]
    assert sha(csa.bwt().tobytes()) == str(g[f"{name}/bwt_sha"][0])
    c2c, Cc = csa.alphabet()
    assert np.array_equal(c2c, g[f"{name}/char2comp"]) and np.array_equal(Cc, g[f"{name}/C"])
    for m in (1, 2, 4, 20):
        if f"{name}/pat{m}" not in g.files:
            continue
        pats = g[f"{name}/pat{m}"]
        assert np.array_equal(csa.count_batch(pats, m), g[f"{name}/count{m}"])


def test_fm_known_answers():
    # SURVEY §8(c)
    f = ol.OCsa(gd.text("faust.txt"))
    assert (f.size(), f.sigma(), f.count(b"und"), f.wt().bv_size()) == (226836, 92, 690, 1096825)
    a = ol.OCsa(b"abracadabra")
    assert bytes(a.bwt()) == b"ard\x00rcaaaabb"
    assert list(a.alphabet()[1][:7]) == [0, 1, 6, 8, 9, 10, 12]
    assert [a.count(b"abra"), a.count(b"xyz"), a.count(b"a" * 22), a.count(b"")] == [2, 0, 0, 12]
    # test/csa_byte_test.cpp:81-106: whole text -> interval of size 1; empty pattern -> [0, size-1]
    t = gd.text("example01.txt")
    c = ol.OCsa(t)
    l, r = c.interval(t)
    assert r + 1 - l == 1
    assert c.interval(b"") == (0, c.size() - 1)


@pytest.mark.parametrize("name", [t for t in gd.TEXTS if t not in ("empty.txt", "all_symbols.txt")])
def test_fm_locate_extract(name):
    """SA / ISA / LF / psi access, extract and locate against the real library's answers"""
    g = gd.text_golden()
    if f"{name}/csa_idx" not in g.files:
        pytest.skip("text contains a 0 byte: not indexable (construct.hpp:41)")
    data = gd.text(name)
    csa = ol.OCsa(data)
    idx = g[f"{name}/csa_idx"][:400]
    assert np.array_equal(csa.sa(idx), g[f"{name}/csa_sa"][:400])
    assert np.array_equal(csa.isa(idx), g[f"{name}/csa_isa"][:400])
    assert np.array_equal(csa.lf(idx), g[f"{name}/csa_lf"][:400])
    assert np.array_equal(csa.psi(idx), g[f"{name}/csa_psi"][:400])
    eb, ee = g[f"{name}/ext_b"], g[f"{name}/ext_e"]
    want = g[f"{name}/ext_text"].tobytes()
    got = b"".join(csa.extract(int(b), int(e)) for b, e in zip(eb, ee))
    assert got == want
    full = data + b"\0"  # the indexed sequence (construct.hpp:100-108)
    assert all(csa.extract(int(b), int(e)) == full[int(b):int(e) + 1] for b, e in zip(eb[:50], ee[:50]))
    for m in (2, 4, 20):
        if f"{name}/loc_n{m}" not in g.files:
            continue
        k = min(int(g[f"{name}/loc_n{m}"][0]), 60)
        off, pos, pats = g[f"{name}/loc_off{m}"], g[f"{name}/loc_pos{m}"], g[f"{name}/pat{m}"]
        for i in range(k):
            assert np.array_equal(csa.locate(pats[i * m:(i + 1) * m].tobytes()), pos[int(off[i]):int(off[i + 1])])


def test_fm_other_densities_same_answers():
    """the sampling densities change the walks, never the answers"""
    data = gd.text("faust.txt")[:20000]
    a, b = ol.OCsa(data), ol.OCsa(data, sa_dens=7, isa_dens=1000)
    idx = np.arange(0, a.size(), 97, dtype=np.uint64)
    assert np.array_equal(a.sa(idx), b.sa(idx)) and np.array_equal(a.isa(idx), b.isa(idx))
    assert a.extract(19000, 20000) == b.extract(19000, 20000) == data[19000:] + b"\0"
    assert np.array_equal(a.sa(a.isa(idx)), idx)


@pytest.mark.parametrize("name", [n for n in gd.bv_case_names()])
def test_two_bit_pattern_model(name):
    """the occurrence-vector model of rank_support_v5<pat,2> / select_support_mcl<pat,2> against the real library"""
    g = gd.bv_golden()
    if f"{name}/pat0_rank" not in g.files:
        pytest.skip("no pattern vectors for this case")
    words, n = gd.bv_case(name)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
    idx = g[f"{name}/idx"]
    for pat in range(4):
        d = ol.pattern_bits(bits, pat)
        cum = np.concatenate([[0], np.cumsum(d)]).astype(np.uint64)
        assert np.array_equal(cum[idx.astype(np.int64)], g[f"{name}/pat{pat}_rank"])
        si = g[f"{name}/pat{pat}_sel_i"]
        if si.size:
            assert np.array_equal(np.flatnonzero(d)[(si - np.uint64(1)).astype(np.int64)].astype(np.uint64),
                                  g[f"{name}/pat{pat}_sel"])


@pytest.mark.parametrize("name", gd.bv_case_names())
def test_sd_vector(name):
    """sd_vector<> restatement against the real library's answers and serialised bytes"""
    g = gd.bv_golden()
    if f"{name}/sd_rank1" not in g.files:
        pytest.skip("no sd_vector vectors for this case")
    words, n = gd.bv_case(name)
    sd = ol.OSd(words, n)
    idx = g[f"{name}/idx"]
    assert np.array_equal(sd.rank(idx[:500], 1), g[f"{name}/sd_rank1"][:500])
    assert np.array_equal(sd.rank(idx[:200], 0), idx[:200] - g[f"{name}/sd_rank1"][:200])
    for b in (0, 1):
        si = g[f"{name}/sel{b}_i"][:600]
        if si.size:
            assert np.array_equal(sd.select(si[:300], b), g[f"{name}/sd_sel{b}"][:300])
            assert np.array_equal(g[f"{name}/sd_sel{b}"], g[f"{name}/sel{b}"][:600])  # == the plain vector's select
    ai = idx[idx < n]
    assert np.array_equal(sd.access(ai[:500]), g[f"{name}/sd_acc"][:500])
    if os.path.exists(os.path.join(gd.GOLDEN, "sdsl", f"{name}.sd_vector.sdsl")):
        assert sd.serialize() == gd.sdsl_file(f"{name}.sd_vector.sdsl")


def test_sd_vector_large_universe():
    g = gd.bv_golden()
    pos = g["sdpos/pos"]
    sd = ol.OSd(positions=pos)
    assert sd.size() == int(g["sdpos/n"][0]) and sd.ones() == pos.size
    assert np.array_equal(sd.rank(g["sdpos/idx"][:800], 1), g["sdpos/rank1"][:800])
    assert np.array_equal(sd.select(g["sdpos/sel1_i"][:500], 1), g["sdpos/sel1"][:500])
    assert np.array_equal(sd.select(g["sdpos/sel0_i"][:100], 0), g["sdpos/sel0"][:100])
    assert np.array_equal(pos[(g["sdpos/sel1_i"] - np.uint64(1)).astype(np.int64)], g["sdpos/sel1"])
    assert sd.serialize() == gd.sdsl_file("sdpos.sd_vector.sdsl")

