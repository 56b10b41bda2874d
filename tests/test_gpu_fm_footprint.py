"""sdsl_hip_fm_set_footprint: the index gives HBM back (binary tree levels, then suffix array and text -> SDSL's default samples packed to
32 bits, k-mer table as deep as the budget allows) and every query keeps its answers — count of large and small batches, intervals,
csa[i], isa[i], lf, psi, locate, extract, the tree's rank / select / access — and the serialised stream stays byte-identical to the one
the real library loads (the binary levels are rebuilt from the fused lines for the time of the call).  Oracle: the C restatement."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def answers(csa, text, rng_seed=5):
    rng = np.random.default_rng(rng_seed)
    n = text.size
    m, npat = 12, 30_000
    st = rng.integers(0, n - m, npat)
    pats = text[st[:, None] + np.arange(m)[None, :]].copy()
    pats[::4, rng.integers(0, m)] = text[rng.integers(0, n)]
    flat = np.ascontiguousarray(pats.reshape(-1))
    out = {"count": np.asarray(csa.count(flat, m)).astype(np.uint64),
           "count_small": np.asarray(csa.count(flat[: 100 * m], m)).astype(np.uint64)}
    lo, hi = csa.interval(flat[: 2000 * m], m)
    out["l"], out["r"] = np.asarray(lo), np.asarray(hi)
    idx = rng.integers(0, n + 1, 3000).astype(np.uint64)
    out["sa"] = np.asarray(csa.sa(idx))
    out["isa"] = np.asarray(csa.isa(idx))
    out["lf"] = np.asarray(csa.lf(idx))
    out["psi"] = np.asarray(csa.psi(idx))
    off, pos = csa.locate(flat[: 300 * m], m)
    out["loc_off"] = np.asarray(off)
    out["loc"] = np.concatenate([np.sort(np.asarray(pos)[int(off[i]):int(off[i + 1])]) for i in range(300)]) if len(pos) else np.zeros(0)
    b = rng.integers(0, n - 200, 200).astype(np.uint64)
    eo, et = csa.extract(b, b + np.uint64(150))
    out["ext"] = np.asarray(et)
    wt = csa.wavelet_tree
    qi = rng.integers(0, n + 2, 5000).astype(np.uint64)
    qc = text[rng.integers(0, n, 5000)]
    out["wt_rank"] = np.asarray(wt.rank(qi, qc))
    out["wt_access"] = np.asarray(wt.access(qi[qi <= n]))
    occ = np.bincount(text, minlength=256)
    k = (1 + rng.integers(0, 1 << 40, 5000) % np.maximum(occ[qc], 1)).astype(np.uint64)
    out["wt_select"] = np.asarray(wt.select(k, qc))
    return out


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("shape", ["english", "sigma3"])
def test_every_stage_keeps_every_answer(gpu, shape):
    if shape == "english":
        text = gpu.english_text(3 << 20, 21)
    else:
        text = np.random.default_rng(2).integers(1, 4, 400_000, dtype=np.uint8)
    ocsa = ol.OCsa(bytes(text))
    csa = gpu.csa_wt(text=text)
    before = answers(csa, text)
    rng = np.random.default_rng(5)
    assert np.array_equal(before["count"][:500], np.asarray(ocsa.count_batch(np.ascontiguousarray(
        _pats(text, 5)[: 500 * 12]), 12)).astype(np.uint64))
    blob = csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL)
    full = csa.device_bytes()
    parts = csa.footprint_parts()
    assert sum(parts.values()) == full and parts["wt_binary_levels"] > 0 and parts["suffix_array"] == 4 * (text.size + 1)
    # stage 1: only the binary levels have to go
    csa.set_footprint(full - parts["wt_binary_levels"] // 2)
    p1 = csa.footprint_parts()
    assert p1["wt_binary_levels"] == 0 and p1["suffix_array"] == parts["suffix_array"] and csa.device_bytes() <= full - parts["wt_binary_levels"] // 2
    assert same(before, answers(csa, text))
    assert csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL) == blob, "the stream must not change"
    assert csa.footprint_parts()["wt_binary_levels"] == 0, "the levels rebuilt for serialize are released again"
    # stage 2: 1.5 x the reference's own stream (sigma 3: 2.2 x — a fused line spends 4 bits per symbol and fused level whatever the
    # alphabet, the reference's Huffman levels 1.7 bits on three symbols)
    budget = int((1.5 if shape == "english" else 2.2) * len(blob))
    csa.set_footprint(budget)
    p2 = csa.footprint_parts()
    assert csa.device_bytes() <= budget and p2["suffix_array"] == 0 and p2["text"] == 0 and p2["wt_binary_levels"] == 0
    assert p2["sa_isa_samples"] == 4 * ((text.size + 32) // 32 + (text.size + 64) // 64)
    assert csa.sampling() == (32, 64, False)
    assert same(before, answers(csa, text))
    assert csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL) == blob
    # the k-mer table cannot be rebuilt without suffix array and text: the call fails and the table the index has stays
    k_now = csa.kmer_table_depth()
    with pytest.raises(Exception):
        csa.set_kmer_table(8, 1 << 30)
    assert csa.kmer_table_depth() == k_now
    # stage 3: the floor (no k-mer table at all)
    floor = sum(v for k, v in p2.items() if k != "kmer_table")
    csa.set_footprint(floor)
    assert csa.device_bytes() <= floor and csa.kmer_table_depth() == 0
    assert same(before, answers(csa, text))
    with pytest.raises(Exception) as e:
        csa.set_footprint(floor // 2)
    assert "smallest form" in str(e.value)
    # and back: suffix array, text and the default table return, answers unchanged
    csa.restore_suffix_array()
    assert csa.footprint_parts()["suffix_array"] > 0 and csa.kmer_table_depth() >= 1
    assert same(before, answers(csa, text))
    csa.close()


def _pats(text, seed):
    rng = np.random.default_rng(seed)
    n, m, npat = text.size, 12, 30_000
    st = rng.integers(0, n - m, npat)
    pats = text[st[:, None] + np.arange(m)[None, :]].copy()
    pats[::4, rng.integers(0, m)] = text[rng.integers(0, n)]
    return pats.reshape(-1)


def test_a_loaded_stream_can_be_shrunk_too(gpu):
    """an index loaded from the real library's stream (no suffix array, no text): the binary levels go, the samples are packed, and a
    k-mer table within the budget is built from suffix array and text that exist only for the time of the call"""
    text = gpu.english_text(2 << 20, 4)
    built = gpu.csa_wt(text=text)
    blob = built.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL)
    want = answers(built, text)
    built.close()
    csa = gpu.csa_wt(sdsl_bytes=blob, select_is_mcl=True, sa_dens=32, isa_dens=64)
    assert csa.kmer_table_depth() == 0
    budget = int(1.5 * len(blob))
    csa.set_footprint(budget)
    p = csa.footprint_parts()
    assert csa.device_bytes() <= budget and p["wt_binary_levels"] == 0 and p["suffix_array"] == 0 and p["text"] == 0
    assert csa.kmer_table_depth() >= 2, "the budget leaves room for a table"
    assert same(want, answers(csa, text))
    assert csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL) == blob
    csa.close()


def test_what_cannot_be_shrunk_says_so(gpu):
    text = gpu.english_text(1 << 20, 9)
    crrr = gpu.csa_wt(text=text, rrr=True)
    crrr.set_footprint(1 << 40)  # already below: nothing to do
    before = crrr.footprint_parts()
    with pytest.raises(Exception) as e:
        crrr.set_footprint(1000)
    assert "smallest form" in str(e.value) and "rrr-compressed tree" in str(e.value)
    assert crrr.footprint_parts() == before, "a refused call leaves the index as it was"
    crrr.close()
    # created from a BWT: no suffix array and no samples to fall back to
    bw = gpu.csa_wt(bwt=np.asarray(ol.OCsa(bytes(text[:50_000])).bwt()))
    with pytest.raises(Exception) as e:
        bw.set_footprint(1000)
    assert "nothing smaller" in str(e.value) or "smallest form" in str(e.value)
    bw.close()


@pytest.mark.parametrize("source", ["text", "stream"])
def test_the_compressed_index_at_a_compressed_size(gpu, source):
    """csa_wt<wt_huff<rrr_vector<63>>> (csa_wt.hpp:389-402, rrr_vector.hpp:366-378): created from text the device image holds suffix array,
    text and a k-mer table — 11 x the type's own stream; set_footprint takes it down to 1.5 x that stream (rrr tree, 32-bit samples, the
    k-mer table the rest of the budget holds) with every answer unchanged, and the stream it serialises to stays the real library's."""
    text = gpu.english_text(3 << 20, 17)
    built = gpu.csa_wt(text=text, rrr=True)
    blob = built.serialize(32, 64, gpu.capi.LAYOUT_RRR63)
    want = answers(built, text)
    if ol.have_ref():
        assert blob == ol.ref_csa_rrr_bytes(bytes(text)), "the stream is the real library's csa_wt<wt_huff<rrr_vector<63>>, 32, 64>"
    if source == "text":
        csa = built
        assert csa.device_bytes() > 5 * len(blob)
    else:
        built.close()
        csa = gpu.csa_wt(sdsl_bytes=blob, rrr=True, sa_dens=32, isa_dens=64)
    # (on the 1 GiB bench text the floor is 1.27 x the stream and 1.5 x leaves room for a k-mer table of depth 4: bench.py's row
    # fm_count_rrr63_lean; on three MiB the 128-byte records of the device's rrr layout weigh more: 1.58 x)
    budget = int(1.8 * len(blob))
    csa.set_footprint(budget)
    p = csa.footprint_parts()
    assert csa.device_bytes() <= budget and p["suffix_array"] == 0 and p["text"] == 0 and p["wt_fused_lines"] == 0
    assert csa.kmer_table_depth() >= 1, "the budget leaves room for a table"
    assert p["sa_isa_samples"] == 4 * ((text.size + 1 + 31) // 32 + (text.size + 1 + 63) // 64), "SDSL's samples at 32 / 64, 32 bits each"
    assert csa.sampling() == (32, 64, False)
    assert same(want, answers(csa, text))
    assert csa.serialize(32, 64, gpu.capi.LAYOUT_RRR63) == blob
    # a second, smaller budget: the k-mer table goes, nothing else can
    floor = sum(v for k, v in p.items() if k != "kmer_table")
    csa.set_footprint(floor)
    assert csa.device_bytes() <= floor and csa.kmer_table_depth() == 0
    assert same(want, answers(csa, text))
    with pytest.raises(Exception):
        csa.set_footprint(floor - (1 << 16))
    assert same(want, answers(csa, text)), "a refused call changes nothing"
    csa.restore_suffix_array()
    assert csa.sampling() == (32, 64, True) and same(want, answers(csa, text))
    csa.close()


def test_fused_header_kernels_stride_over_what_the_grid_does_not_cover(gpu, monkeypatch):
    """k_wt8_counts / k_wt8_cross are launched with a capped grid (2^20 blocks) and stride over the rest — a node of 2^33 symbols and
    more has more lines than that.  With the cap at 8 blocks a 2^21-symbol tree is built entirely by the striding."""
    rng = np.random.default_rng(8)
    text = rng.integers(1, 60, 1 << 21, dtype=np.uint8)
    qi = rng.integers(0, text.size + 1, 200_000).astype(np.uint64)
    qc = text[rng.integers(0, text.size, 200_000)]
    wt = gpu.wt_huff(text=text)
    want = np.asarray(wt.rank(qi, qc))
    wt.close()
    monkeypatch.setenv("SDSL_HIP_WT8_GRID_CAP", "8")
    wt2 = gpu.wt_huff(text=text)
    assert wt2.fused_steps().any()
    assert np.array_equal(np.asarray(wt2.rank(qi, qc)), want)
    wt2.close()


def test_samples_at_the_densities_of_the_callers_type(gpu):
    """csa_wt<wt_huff<>, t_dens, t_inv_dens>: drop_sa(8, 16) keeps SA samples every 8th suffix and ISA samples every 16th position; the walks
    and the stream written from those samples equal what the whole suffix array gives"""
    text = gpu.english_text(1 << 20, 13)
    ocsa = ol.OCsa(bytes(text))
    full = gpu.csa_wt(text=text)
    blob_8_16 = full.serialize(8, 16)
    full.close()
    csa = gpu.csa_wt(text=text)
    csa.drop_sa(8, 16)
    assert csa.sampling() == (8, 16, False)
    idx = np.random.default_rng(1).integers(0, text.size + 1, 20_000).astype(np.uint64)
    assert np.array_equal(np.asarray(csa.sa(idx)), np.asarray(ocsa.sa(idx)))
    assert np.array_equal(np.asarray(csa.isa(idx)), np.asarray(ocsa.isa(idx)))
    off, txt = csa.extract(np.array([3, 1000, text.size - 300], dtype=np.uint64), np.array([259, 1001, text.size - 1], dtype=np.uint64))
    assert bytes(np.asarray(txt)) == bytes(text[3:260]) + bytes(text[1000:1002]) + bytes(text[text.size - 300:])
    assert csa.serialize(8, 16) == blob_8_16
    with pytest.raises(Exception):
        csa.drop_sa(32, 64)  # other densities need the suffix array back
    csa.restore_suffix_array()
    csa.drop_sa(32, 64)
    assert csa.sampling() == (32, 64, False)
    assert np.array_equal(np.asarray(csa.sa(idx)), np.asarray(ocsa.sa(idx)))
    csa.close()


@pytest.mark.parametrize("mode", ["text resident: a copy", "samples: LF walks"])
def test_extract_of_every_alignment_and_length(gpu, mode):
    """the walk collects its bytes into aligned words (locate.hip: emit), the copy of a resident text fills 16 output bytes per thread across
    range borders: every start alignment x every length around the word size, neighbouring snippets that share a word, the sentinel"""
    text = gpu.english_text(1 << 18, 3)
    csa = gpu.csa_wt(text=text)
    if mode.startswith("samples"):
        csa.drop_sa()
    off, txt = csa.extract(np.array([text.size - 5, 7, 9], dtype=np.uint64), np.array([text.size, 6, 9], dtype=np.uint64))  # to the sentinel; b > e: empty
    assert bytes(np.asarray(txt)) == bytes(text[-5:]) + b"\0" + bytes(text[9:10]) and list(np.asarray(off)) == [0, 6, 6, 7]
    b = np.array([s0 + 100 * k for k, s0 in enumerate(range(1000, 1064))] * 1, dtype=np.uint64)
    for ln in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 200):
        e = b + np.uint64(ln - 1)
        off, txt = csa.extract(b, e)
        txt = np.asarray(txt)
        want = np.concatenate([text[int(x):int(x) + ln] for x in b])
        assert np.array_equal(txt, want), ln
    csa.close()


@pytest.mark.parametrize("shape", ["huff", "blcd", "hutu"])
def test_a_wavelet_tree_without_its_binary_levels(gpu, shape):
    """sdsl_hip_wt_release_binary_levels on a stand-alone tree of every byte shape: same rank / select / access / inverse_select, fewer
    bytes, and the serialised stream — whose levels are rebuilt from the fused lines with the tree's own node table — stays what it was"""
    text = gpu.english_text(1 << 20, 31)
    wt = gpu.wt_huff(text=text, balanced=(shape == "blcd"), hutu=(shape == "hutu"))
    rng = np.random.default_rng(2)
    qi = rng.integers(0, text.size + 1, 50_000).astype(np.uint64)
    qc = text[rng.integers(0, text.size, 50_000)]
    occ = np.bincount(text, minlength=256)
    k = (1 + rng.integers(0, 1 << 40, 50_000) % occ[qc]).astype(np.uint64)
    before = (np.asarray(wt.rank(qi, qc)), np.asarray(wt.select(k, qc)), np.asarray(wt.access(qi[qi < text.size])), wt.serialize(gpu.capi.LAYOUT_BV_MCL))
    full = wt.device_bytes()
    wt.release_binary_levels()
    assert wt.device_bytes() < 0.62 * full
    after = (np.asarray(wt.rank(qi, qc)), np.asarray(wt.select(k, qc)), np.asarray(wt.access(qi[qi < text.size])), wt.serialize(gpu.capi.LAYOUT_BV_MCL))
    assert all(np.array_equal(a, b) if not isinstance(a, bytes) else a == b for a, b in zip(before, after))
    assert wt.device_bytes() < 0.62 * full, "the levels rebuilt for serialize are released again"
    wt.close()


import golden_data as gd  # noqa: E402


@pytest.mark.parametrize("name", [t for t in gd.TEXTS if t not in ("empty.txt", "all_symbols.txt")])
def test_the_references_own_fixtures_at_every_footprint(gpu, name):
    """test/test_cases of the reference (a 100-fold 'a': sigma 1; one byte; tiny periodic texts; faust.txt): the golden count / interval
    vectors of the real library on the index as created, with the binary levels released, at the floor, and restored — or a clean refusal
    where a structure does not exist for the text (no fused layout for a one-symbol alphabet), never a changed answer"""
    g = gd.text_golden()
    if f"{name}/csa_meta" not in g.files:
        pytest.skip("text contains a 0 byte: not indexable (construct.hpp:41)")
    data = gd.text(name)
    csa = gpu.csa_wt(text=data)
    oc = ol.OCsa(data)
    n = len(data)

    def check(stage):
        for m in (1, 2, 4, 20):
            if f"{name}/pat{m}" in g.files:
                pats = g[f"{name}/pat{m}"]
                assert np.array_equal(csa.count(pats, m), g[f"{name}/count{m}"]), (stage, m)
                l, r = csa.interval(pats, m)
                assert np.array_equal(l, g[f"{name}/ival_l{m}"]) and np.array_equal(r, g[f"{name}/ival_r{m}"]), (stage, m)
        idx = np.arange(n + 1, dtype=np.uint64) if n < 5000 else np.random.default_rng(1).integers(0, n + 1, 5000).astype(np.uint64)
        assert np.array_equal(np.asarray(csa.sa(idx)), np.asarray(oc.sa(idx))), stage
        assert np.array_equal(np.asarray(csa.isa(idx)), np.asarray(oc.isa(idx))), stage
        if n:
            off, txt = csa.extract(np.array([0], dtype=np.uint64), np.array([n - 1], dtype=np.uint64))
            assert bytes(np.asarray(txt)) == bytes(data), stage

    check("as created")
    blob = csa.serialize(32, 64)
    parts = csa.footprint_parts()
    try:
        csa.set_footprint(csa.device_bytes() - max(1, parts["wt_binary_levels"] // 2))
    except gpu.capi.SdslHipError as e:
        assert e.status in (gpu.capi.ERR_UNSUPPORTED, gpu.capi.ERR_INVALID), e
        check("after a refused set_footprint")
        assert csa.serialize(32, 64) == blob
        csa.close()
        return
    check("binary levels released")
    p = csa.footprint_parts()
    floor = sum(v for k, v in p.items() if k not in ("kmer_table", "suffix_array", "text", "wt_binary_levels", "sa_isa_samples", "jump_table"))
    floor += 4 * ((n + 32) // 32 + (n + 64) // 64) + 4096
    was = csa.device_bytes()
    csa.set_footprint(floor + (1 << 16))
    assert csa.device_bytes() <= floor + (1 << 16)
    if was > floor + (1 << 16):  # (a text of 100 bytes is below any budget as it stands)
        assert csa.footprint_parts()["suffix_array"] == 0
    check("at the floor")
    assert csa.serialize(32, 64) == blob
    csa.restore_suffix_array()
    check("restored")
    csa.close()


def test_an_index_whose_suffix_array_is_wider_than_its_intervals_shrinks_too(gpu, monkeypatch):
    """an index that holds the 64-BIT suffix array although its intervals are 32 bits wide (2^32 - 2 or 2^32 - 1 symbols; or any
    text sent through the 64-bit sorter by SDSL_HIP_SA64) cannot build a k-mer table — set_footprint failed there AFTER it had changed the
    index.  Now: the call is transactional, and such an index goes without a table instead of failing."""
    text = gpu.english_text(1 << 20, 31)
    ref = gpu.csa_wt(text=text)
    want = answers(ref, text)
    blob = ref.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL)
    ref.close()
    monkeypatch.setenv("SDSL_HIP_SA64", "1")
    csa = gpu.csa_wt(text=text)
    monkeypatch.delenv("SDSL_HIP_SA64")
    assert csa.sampling() == (32, 64, True) and csa.footprint_parts()["suffix_array"] == 8 * (text.size + 1)
    assert same(want, answers(csa, text))
    budget = int(1.6 * len(blob))
    csa.set_footprint(budget)
    p = csa.footprint_parts()
    assert csa.device_bytes() <= budget and p["suffix_array"] == 0 and p["text"] == 0 and p["wt_binary_levels"] == 0
    assert same(want, answers(csa, text))
    assert csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL) == blob
    csa.close()
