"""GPU parity tests proper: every call goes through the C ABI of libsdsl_hip.so and is compared
bit-exactly with (a) the committed golden vectors produced by the real sdsl-lite, (b) the CPU
restatement (oracle/) on seeded inputs, and (c) the real library itself when oracle/_ref travelled
with the snapshot.  Integer work only: the bar is equality, no tolerances anywhere."""
import os

import numpy as np
import pytest

import golden_data as gd
import oracle_lib as ol
from conftest import unpack_bits

pytestmark = pytest.mark.gpu
NPOS = np.uint64(0xFFFFFFFFFFFFFFFF)


def mk(n, d, seed):
    r = np.random.default_rng(seed)
    nw = (n + 63) // 64
    if d == 0.5:
        return r.integers(0, 2**64, size=nw, dtype=np.uint64)
    bits = (r.random(nw * 64) < d).astype(np.uint8)
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


# ---------------------------------------------------------------------------------------------------
# plain bit vector: rank_support_v5 / select_support_mcl
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gd.bv_case_names())
def test_bv_golden(gpu, name):
    g = gd.bv_golden()
    w, n = gd.bv_case(name)
    bv = gpu.bit_vector(w, n)
    idx = g[f"{name}/idx"]
    for b in (0, 1):
        assert np.array_equal(bv.rank(idx, b), g[f"{name}/rank{b}"])
        tot = int(g[f"{name}/total{b}"][0])
        assert (bv.ones() if b else n - bv.ones()) == tot
        if tot:
            assert np.array_equal(bv.select(g[f"{name}/sel{b}_i"], b), g[f"{name}/sel{b}"])


@pytest.mark.parametrize("n,d", [(1 << 20, 0.5), (3_000_000, 0.02), (999_999, 0.97), (200_000, 0.5)])
def test_bv_vs_oracle(gpu, n, d):
    rng = np.random.default_rng(n)
    w = mk(n, d, 21)
    o = ol.OBitVector(w, n)
    bv = gpu.bit_vector(w, n)
    idx = rng.integers(0, n + 1, size=200_000, dtype=np.uint64)
    for b in (0, 1):
        assert np.array_equal(bv.rank(idx, b), o.rank(idx, b))
        ac = o.arg_cnt(b)
        i = rng.integers(1, ac + 1, size=200_000, dtype=np.uint64)
        assert np.array_equal(bv.select(i, b), o.select(i, b))


def test_bv_vs_real_sdsl(gpu):
    if not ol.have_ref():
        pytest.skip("oracle/_ref not present")
    n = 2_500_000
    w = mk(n, 0.3, 77)
    r = ol.RBitVector(w, n)
    bv = gpu.bit_vector(w, n)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, n + 1, size=100_000, dtype=np.uint64)
    assert np.array_equal(bv.rank(idx, 1), r.rank(idx, 1))
    assert np.array_equal(bv.rank(idx, 1), r.rank_v(idx))  # rank_support_v gives the same answers
    i = rng.integers(1, bv.ones() + 1, size=100_000, dtype=np.uint64)
    assert np.array_equal(bv.select(i, 1), r.select(i, 1))


def test_bv_reference_test_semantics(gpu):
    """test/rank_support_test.cpp:109-128 and test/select_support_test.cpp:85-104 on the fixture every
    int-vec.* entry of the reference's config resolves to: all positions, all set bits."""
    w, n = gd.bv_case("CRAFTED-MAT-SELECT")
    bits = unpack_bits(w, n).astype(np.uint64)
    bv = gpu.bit_vector(w, n)
    pref = np.concatenate([[0], np.cumsum(bits)]).astype(np.uint64)
    assert np.array_equal(bv.rank(np.arange(n + 1, dtype=np.uint64), 1), pref)
    pos = np.nonzero(bits)[0].astype(np.uint64)
    assert pos.size == 4061
    assert np.array_equal(bv.select(np.arange(1, pos.size + 1, dtype=np.uint64), 1), pos)


# ---------------------------------------------------------------------------------------------------
# rrr_vector<63>
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gd.bv_case_names())
def test_rrr_golden(gpu, name):
    g = gd.bv_golden()
    w, n = gd.bv_case(name)
    v = gpu.rrr_vector(w, n)
    assert v.size() == n
    idx = g[f"{name}/idx"]
    for b in (0, 1):
        assert np.array_equal(v.rank(idx, b), g[f"{name}/rank{b}"])
        assert np.array_equal(v.select(g[f"{name}/rrr_sel{b}_i"], b), g[f"{name}/rrr_sel{b}"])
    assert v.ones() == int(g[f"{name}/total1"][0])


@pytest.mark.parametrize("n", [0, 1, 62, 63, 64, 126, 630, 2015, 2016, 2017, 63 * 32 * 3, 100_000, 1_000_003])
@pytest.mark.parametrize("d", [0.5, 0.05, 0.95, 0.0, 1.0])
def test_rrr_vs_oracle(gpu, n, d):
    rng = np.random.default_rng(n + 1)
    w = mk(n, d, n + 5)
    o = ol.ORrr(w, n)
    v = gpu.rrr_vector(w, n)
    idx = np.arange(n + 1, dtype=np.uint64) if n <= 5000 else rng.integers(0, n + 1, size=50_000, dtype=np.uint64)
    for b in (0, 1):
        assert np.array_equal(v.rank(idx, b), o.rank(idx, b))
        tot = int(o.rank([n], b)[0])
        i = (np.arange(1, tot + 2, dtype=np.uint64) if tot <= 5000 else
             rng.integers(1, tot + 2, size=50_000, dtype=np.uint64))  # includes the overflow value tot+1 -> size()
        assert np.array_equal(v.select(i, b), o.select(i, b))
        assert int(v.select(np.array([0], dtype=np.uint64), b)[0]) == int(NPOS)
    if n:
        ia = idx[idx < n][:20000]
        assert np.array_equal(v.access(ia), unpack_bits(w, n)[ia.astype(np.int64)])


@pytest.mark.parametrize("name", ["CRAFTED-32", "rnd.8192.1043", "rnd.200000.7", "rnd.10080.9"])
def test_rrr_loads_sdsl_stream(gpu, name):
    """the bytes written by the real rrr_vector<63>::serialize are accepted as they are"""
    g = gd.bv_golden()
    v = gpu.rrr_vector(sdsl_bytes=gd.sdsl_file(f"{name}.rrr63.sdsl"))
    idx = g[f"{name}/idx"]
    for b in (0, 1):
        assert np.array_equal(v.rank(idx, b), g[f"{name}/rank{b}"])
        assert np.array_equal(v.select(g[f"{name}/rrr_sel{b}_i"], b), g[f"{name}/rrr_sel{b}"])
    with pytest.raises(gpu.capi.SdslHipError) as e:
        gpu.rrr_vector(sdsl_bytes=gd.sdsl_file(f"{name}.rrr63.sdsl")[:-9])
    assert e.value.status == gpu.capi.ERR_FORMAT


def test_rrr_inline_overflow_path(gpu):
    # dense superblocks have ~1950 offset bits: fields beyond the 640 inline bits come from the stream
    n = 63 * 32 * 50 + 17
    w = mk(n, 0.5, 3)
    o = ol.ORrr(w, n)
    v = gpu.rrr_vector(w, n)
    idx = np.arange(n + 1, dtype=np.uint64)
    assert np.array_equal(v.rank(idx, 1), o.rank(idx, 1))
    tot = v.ones()
    i = np.arange(1, tot + 1, dtype=np.uint64)
    assert np.array_equal(v.select(i, 1), o.select(i, 1))


@pytest.mark.parametrize("d", [0.02, 0.1, 0.3, 0.5, 0.85, 0.97])
def test_rrr_raw_class_budget_changes_neither_answers_nor_sdsl_bytes(gpu, d):
    """Which classes the device stores raw (option rrr_raw_budget: none beyond 11..52, the default 2 %, everything) is a
    layout choice: rank / select / access / get_int and the serialised rrr_vector<63> bytes are the same for every budget,
    whether the vector is built from the bits or loaded from SDSL's stream."""
    n = 2142 * 700 + 29
    w = mk(n, d, int(d * 1000))
    o = ol.ORrr(w, n)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, n + 1, 100_000, dtype=np.uint64)
    blobs = []
    try:
        for budget in (0, 20, 200, 1000):
            gpu.set_option("rrr_raw_budget", budget)
            v = gpu.rrr_vector(w, n)
            blob = v.serialize()
            blobs.append(blob)
            for vv in (v, gpu.rrr_vector(sdsl_bytes=blob)):
                assert np.array_equal(vv.rank(idx, 1), o.rank(idx, 1))
                tot = vv.ones()
                i = rng.integers(1, tot + 1, 50_000, dtype=np.uint64)
                assert np.array_equal(vv.select(i, 1), o.select(i, 1))
                i0 = rng.integers(1, n - tot + 1, 50_000, dtype=np.uint64)
                assert np.array_equal(vv.select(i0, 0), o.select(i0, 0))
                pos = idx[idx + 64 <= n]
                assert np.array_equal(vv.get_int(pos, 64) & np.uint64(1), vv.access(pos).astype(np.uint64))
                assert vv.serialize() == blob
    finally:
        gpu.set_option("rrr_raw_budget", 20)
    assert all(b == blobs[0] for b in blobs)
    assert blobs[0] == o.serialize()


@pytest.mark.parametrize("n", [0, 1, 14, 15, 16, 63, 64, 479, 480, 481, 959, 960, 961, 100_003, 3_000_001])
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 5, 6])
def test_sibling_streams_are_decoded_on_the_device(gpu, n, kind):
    """bit_vector_il<512 / 64>, the generic rrr_vector<15 / 31 / 62> and the rrr_vector<15> specialisation built by the REAL
    library, handed over as their serialised bytes: the device turns them back into the plain bits (exported words equal) and
    answers rank / select on them.  Densities include 0.97: full superblocks of the generic template are stored inverted."""
    if not ol.have_ref():
        pytest.skip("oracle/_ref not present")
    w = mk(n, (0.5, 0.03, 0.97, 0.3)[(n + kind) % 4], n + 7)
    blob = ol.ref_sibling_bytes(w, n, kind)
    K = gpu.capi
    ck = {0: K.SIBLING_IL, 2: K.SIBLING_IL, 1: K.SIBLING_RRR(15, 32), 3: K.SIBLING_RRR(15, 8), 4: K.SIBLING_RRR(31, 32),
          5: K.SIBLING_RRR(62, 16), 6: K.SIBLING_RRR15}[kind]
    bv = gpu.bit_vector(sdsl_bytes=blob, kind=ck)
    assert bv.size() == n
    plain = gpu.bit_vector(w, n)
    assert np.array_equal(bv.export_words(), plain.export_words())
    if n:
        idx = np.random.default_rng(n).integers(0, n + 1, size=20_000, dtype=np.uint64)
        assert np.array_equal(bv.rank(idx, 1), plain.rank(idx, 1))
        if plain.ones():
            i = np.random.default_rng(n + 1).integers(1, plain.ones() + 1, size=20_000, dtype=np.uint64)
            assert np.array_equal(bv.select(i, 1), plain.select(i, 1))
        with pytest.raises(gpu.capi.SdslHipError) as e:
            gpu.bit_vector(sdsl_bytes=blob[: len(blob) // 2], kind=ck)
        assert e.value.status == gpu.capi.ERR_FORMAT


@pytest.mark.parametrize("n", [0, 15, 961, 100_003, 3_000_001])
@pytest.mark.parametrize("kind", [0, 1, 4, 6])
def test_sibling_streams_stay_compressed_on_request(gpu, n, kind):
    """sdsl_hip_rrr_create_from_sibling: the same streams (bit_vector_il<512>, rrr_vector<15, ., 32> generic, rrr_vector<31>, the
    rrr_vector<15> specialisation of the REAL library) decoded on the device and kept as rrr records: rank / select / access / get_int of
    the source bits, and a sparse vector resident in fewer bytes than the source type's own stream"""
    if not ol.have_ref():
        pytest.skip("oracle/_ref not present")
    d = (0.05, 0.5, 0.97, 0.3)[(n + kind) % 4]
    w = mk(n, d, n + 7)
    blob = ol.ref_sibling_bytes(w, n, kind)
    K = gpu.capi
    ck = {0: K.SIBLING_IL, 1: K.SIBLING_RRR(15, 32), 4: K.SIBLING_RRR(31, 32), 6: K.SIBLING_RRR15}[kind]
    v = gpu.rrr_vector(sdsl_bytes=blob, sibling_kind=ck)
    plain = gpu.bit_vector(w, n)
    assert v.size() == n and v.ones() == plain.ones()
    if n:
        idx = np.random.default_rng(n).integers(0, n + 1, size=20_000, dtype=np.uint64)
        orc = ol.ORrr(w, n)                       # the oracle on the same bits (this test compared two HIP paths only)
        assert np.array_equal(v.rank(idx, 1), orc.rank(idx, 1)) and np.array_equal(v.rank(idx, 0), orc.rank(idx, 0))
        assert np.array_equal(v.rank(idx, 1), plain.rank(idx, 1)) and np.array_equal(v.rank(idx, 0), plain.rank(idx, 0))
        if plain.ones():
            i = np.random.default_rng(n + 1).integers(1, plain.ones() + 1, size=20_000, dtype=np.uint64)
            assert np.array_equal(v.select(i, 1), orc.select(i, 1))
            assert np.array_equal(v.select(i, 1), plain.select(i, 1))
        pos = idx[idx < n]
        bits = unpack_bits(w, n)
        assert np.array_equal(v.access(pos), bits[pos.astype(np.int64)])
        assert v.serialize() == gpu.rrr_vector(w, n).serialize(), "the stream written is rrr_vector<63>'s of the same bits"
    if n >= 3_000_000 and d <= 0.05 and kind in (1, 6):
        assert v.device_bytes() < len(blob), (v.device_bytes(), len(blob))
    with pytest.raises(gpu.capi.SdslHipError):
        gpu.rrr_vector(sdsl_bytes=blob[: max(1, len(blob) // 2)], sibling_kind=ck)


@pytest.mark.parametrize("d", [0.05, 0.5, 0.97])
def test_rrr_get_int_matches_reference(gpu, d):
    """rrr_vector::get_int(idx, len) (rrr_vector.hpp:308-356): windows of every length at every alignment to the 63-bit
    blocks, against the real library when it travelled with the repo and against the plain bits always"""
    n = 63 * 32 * 7 + 29
    w = mk(n, d, 11)
    v = gpu.rrr_vector(w, n)
    bits = unpack_bits(w, n).astype(np.uint64)
    rng = np.random.default_rng(3)
    r = ol.RRrr(w, n) if ol.have_ref() else None
    for length in (1, 2, 31, 62, 63, 64):
        idx = np.concatenate([np.arange(0, min(n - length, 300), dtype=np.uint64),
                              rng.integers(0, n - length + 1, size=4000, dtype=np.uint64),
                              np.array([n - length], dtype=np.uint64)])
        want = np.zeros(idx.size, dtype=np.uint64)
        for k in range(length):
            want |= bits[idx.astype(np.int64) + k] << np.uint64(k)
        got = v.get_int(idx, length)
        assert np.array_equal(got, want), length
        if r is not None:
            assert np.array_equal(got, r.get_int(idx, length))
    assert int(v.get_int(np.array([n - 10], dtype=np.uint64), 11)[0]) == int(NPOS)  # window beyond size()
    assert int(v.get_int(np.array([n], dtype=np.uint64), 0)[0]) == 0


def test_rrr_mixed_inline_and_stream_superblocks(gpu):
    """superblocks whose offsets fit the record's inline area keep no stream storage, dense ones do: a vector that
    alternates between the two, queried everywhere, serialised back to SDSL's bytes, loaded again from them"""
    nsb = 40
    n = 63 * 32 * nsb + 5
    rng = np.random.default_rng(8)
    bits = np.zeros(n, dtype=np.uint8)
    for s in range(nsb):
        lo, hi = s * 2016, min(n, (s + 1) * 2016)
        p = (0.03, 0.5, 0.08, 0.97)[s % 4]
        bits[lo:hi] = rng.random(hi - lo) < p
    w = np.packbits(np.concatenate([bits, np.zeros((-n) % 64, dtype=np.uint8)]), bitorder="little").view(np.uint64)
    o = ol.ORrr(w, n)
    v = gpu.rrr_vector(w, n)
    idx = np.arange(n + 1, dtype=np.uint64)
    assert np.array_equal(v.rank(idx, 1), o.rank(idx, 1))
    tot = v.ones()
    assert np.array_equal(v.select(np.arange(1, tot + 1, dtype=np.uint64), 1), o.select(np.arange(1, tot + 1, dtype=np.uint64), 1))
    assert np.array_equal(v.access(idx[:-1]), bits)
    blob = v.serialize()
    assert blob == o.serialize()
    v2 = gpu.rrr_vector(sdsl_bytes=blob)
    assert np.array_equal(v2.rank(idx, 0), o.rank(idx, 0))
    assert v2.serialize() == blob
    assert v.device_bytes() == v2.device_bytes()


# ---------------------------------------------------------------------------------------------------
# wt_huff
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gd.TEXTS)
def test_wt_golden(gpu, name):
    g = gd.text_golden()
    data = gd.text(name)
    wt = gpu.wt_huff(data)
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


def test_wt_reference_test_semantics(gpu):
    """test/wt_byte_test.cpp:134-168: rank(j+1, text[j]) = running count for every j; 1000 random
    positions for every absent symbol return 0; rank(size(), c) for all 256 c."""
    data = gd.text("faust.txt")
    arr = np.frombuffer(data, dtype=np.uint8)
    wt = gpu.wt_huff(data)
    cnt = np.zeros(256, dtype=np.int64)
    exp = np.empty(arr.size, dtype=np.uint64)
    for c in range(256):
        m = arr == c
        exp[m] = np.arange(1, int(m.sum()) + 1, dtype=np.uint64)
        cnt[c] = m.sum()
    assert np.array_equal(wt.rank(np.arange(1, arr.size + 1, dtype=np.uint64), arr), exp)
    absent = np.nonzero(cnt == 0)[0].astype(np.uint8)
    rng = np.random.default_rng(4)
    pos = rng.integers(0, arr.size + 1, size=1000 * absent.size, dtype=np.uint64)
    assert not wt.rank(pos, np.repeat(absent, 1000)).any()
    assert np.array_equal(wt.rank(np.full(256, arr.size, dtype=np.uint64), np.arange(256, dtype=np.uint8)),
                          cnt.astype(np.uint64))
    assert int(wt.rank(np.array([arr.size + 1], dtype=np.uint64), arr[:1])[0]) == int(NPOS)


@pytest.mark.parametrize("name", ["example01.txt", "abc_abc_abc.txt", "100a.txt", "one_byte.txt"])
@pytest.mark.parametrize("mcl", [True, False])
def test_wt_loads_sdsl_stream(gpu, name, mcl):
    g = gd.text_golden()
    blob = gd.sdsl_file(f"{name}.wt_huff_v5_{'mcl' if mcl else 'scan'}.sdsl")
    wt = gpu.wt_huff(sdsl_bytes=blob, select_is_mcl=mcl)
    assert wt.consumed == len(blob)
    assert np.array_equal(wt.rank(g[f"{name}/rank_i"], g[f"{name}/rank_c"]), g[f"{name}/rank"])
    assert np.array_equal(wt.select(g[f"{name}/sel_i"], g[f"{name}/sel_c"]), g[f"{name}/sel"])
    assert np.array_equal(wt.access(g[f"{name}/acc_i"]), g[f"{name}/acc"])


def test_wt_random_vs_oracle(gpu):
    rng = np.random.default_rng(9)
    for sigma, n in [(2, 5000), (3, 70000), (256, 300000), (40, 1_000_000)]:
        p = rng.random(sigma) ** 3
        p /= p.sum()
        syms = rng.choice(256, size=sigma, replace=False).astype(np.uint8)
        t = syms[rng.choice(sigma, size=n, p=p)]
        o = ol.OWt(t)
        wt = gpu.wt_huff(t)
        assert np.array_equal(wt.code_lengths(), o.code_lengths())
        i = rng.integers(0, n + 1, size=100_000, dtype=np.uint64)
        c = rng.integers(0, 256, size=100_000, dtype=np.uint8)
        c[:60_000] = t[rng.integers(0, n, size=60_000)]
        assert np.array_equal(wt.rank(i, c), o.rank(i, c))
        ia = rng.integers(0, n, size=3000, dtype=np.uint64)
        assert np.array_equal(wt.access(ia), t[ia.astype(np.int64)])
        cs = t[rng.integers(0, n, size=2000)]
        tot = o.rank(np.full(cs.size, n, dtype=np.uint64), cs)
        si = rng.integers(0, 2**62, size=cs.size, dtype=np.uint64) % tot + np.uint64(1)
        assert np.array_equal(wt.select(si, cs), o.select(si, cs))


# ---------------------------------------------------------------------------------------------------
# csa_wt: backward_search / count
# ---------------------------------------------------------------------------------------------------
FM_TEXTS = [t for t in gd.TEXTS if t not in ("empty.txt", "all_symbols.txt")]


def _fm_cases(name):
    g = gd.text_golden()
    if f"{name}/csa_meta" not in g.files:
        pytest.skip("text contains a 0 byte: not indexable (construct.hpp:41)")
    return g


@pytest.mark.parametrize("name", FM_TEXTS)
@pytest.mark.parametrize("how", ["bwt", "text"])
def test_fm_golden(gpu, name, how):
    g = _fm_cases(name)
    data = gd.text(name)
    if how == "bwt":
        csa = gpu.csa_wt(bwt=ol.OCsa(data).bwt())
    else:
        csa = gpu.csa_wt(text=data)  # suffix array built on the device
    assert [csa.size(), csa.sigma()] == [int(x) for x in g[f"{name}/csa_meta"]]
    c2c, Cc = csa.alphabet()
    assert np.array_equal(c2c, g[f"{name}/char2comp"]) and np.array_equal(Cc, g[f"{name}/C"])
    for m in (1, 2, 4, 20):
        if f"{name}/pat{m}" not in g.files:
            continue
        pats = g[f"{name}/pat{m}"]
        assert np.array_equal(csa.count(pats, m), g[f"{name}/count{m}"])
        l, r = csa.interval(pats, m)
        assert np.array_equal(l, g[f"{name}/ival_l{m}"]) and np.array_equal(r, g[f"{name}/ival_r{m}"])


@pytest.mark.parametrize("rrr", [False, True])
def test_fm_jump_table_depths_do_not_change_answers(gpu, rrr):
    """count / interval / locate with the k-mer jump table at depths 0..4 against the golden vectors (which hold
    patterns with absent characters, patterns shorter than the depth and empty intervals)"""
    name = "faust.txt"
    g = _fm_cases(name)
    csa = gpu.csa_wt(text=gd.text(name), rrr=rrr)
    assert csa.jump_depth() >= 1  # the default table exists
    for k in (0, 1, 2, 3, 4):
        csa.set_jump_depth(k)
        assert csa.jump_depth() == k
        for m in (1, 2, 4, 20):
            pats = g[f"{name}/pat{m}"]
            assert np.array_equal(csa.count(pats, m), g[f"{name}/count{m}"])
            l, r = csa.interval(pats, m)
            assert np.array_equal(l, g[f"{name}/ival_l{m}"]) and np.array_equal(r, g[f"{name}/ival_r{m}"])
        kk = int(g[f"{name}/loc_n4"][0])
        off, pos = csa.locate(g[f"{name}/pat4"][: kk * 4], 4)
        assert np.array_equal(off, g[f"{name}/loc_off4"]) and np.array_equal(pos, g[f"{name}/loc_pos4"])
        assert list(csa.count_ragged([b"", b"und", b"e", b"Faust", b"\xff\xfe", b"x" * 300000])) == \
            [csa.size(), 690, 22513, 53, 0, 0]
    with pytest.raises(gpu.capi.SdslHipError):
        csa.set_jump_depth(9)  # 92^9 entries: refused


@pytest.mark.parametrize("name,which", [("example01.txt", "csa_wt_huff_v5"), ("faust.txt", "csa_wt_huff_v5"),
                                        ("example01.txt", "csa_fm_huff"), ("faust.txt", "csa_fm_huff")])
def test_fm_loads_sdsl_stream(gpu, name, which):
    """serialised csa_wt files of the real library: the default type (mcl selects) and the FM_HUFF type of
    benchmark/indexing_count/index.config:8 (scan selects, 2^20 sampling)"""
    g = gd.text_golden()
    csa = gpu.csa_wt(sdsl_bytes=gd.sdsl_file(f"{name}.{which}.sdsl"), select_is_mcl=(which == "csa_wt_huff_v5"))
    assert [csa.size(), csa.sigma()] == [int(x) for x in g[f"{name}/csa_meta"]]
    for m in (1, 4, 20):
        if f"{name}/pat{m}" in g.files:
            assert np.array_equal(csa.count(g[f"{name}/pat{m}"], m), g[f"{name}/count{m}"])
    assert np.array_equal(csa.wavelet_tree.rank(g[f"{name}/rank_i"][:0], g[f"{name}/rank_c"][:0]), np.zeros(0, np.uint64))


def test_fm_known_answers_and_edges(gpu):
    f = gpu.csa_wt(text=gd.text("faust.txt"))
    assert (f.size(), f.sigma(), f.wavelet_tree.bv_size()) == (226836, 92, 1096825)
    assert list(f.count_ragged([b"und", b"", b"Faust", b"\xff\xfe", b"x" * 300000])) == [
        690, 226836, ol.OCsa(gd.text("faust.txt")).count(b"Faust"), 0, 0]
    a = gpu.csa_wt(text=b"abracadabra")
    assert list(a.count_ragged([b"abra", b"xyz", b"a" * 22, b""])) == [2, 0, 0, 12]
    assert list(a.alphabet()[1][:7]) == [0, 1, 6, 8, 9, 10, 12]
    # test/csa_byte_test.cpp:81-106: the whole text matches exactly once; empty pattern -> [0, size-1]
    t = gd.text("example01.txt")
    c = gpu.csa_wt(text=t)
    l, r = c.interval(np.frombuffer(t, dtype=np.uint8), len(t))
    assert int(r[0]) + 1 - int(l[0]) == 1
    with pytest.raises(gpu.capi.SdslHipError):
        gpu.csa_wt(text=b"ab\0cd")


def test_fm_backward_search_steps_vs_oracle(gpu):
    data = gd.text("faust.txt")
    o = ol.OCsa(data)
    csa = gpu.csa_wt(bwt=o.bwt())
    rng = np.random.default_rng(12)
    n = o.size()
    l = rng.integers(0, n, size=20000, dtype=np.uint64)
    r = np.minimum(l + rng.integers(0, 5000, size=l.size, dtype=np.uint64), np.uint64(n - 1))
    l[:100], r[:100] = 0, n - 1  # whole-interval shortcut
    c = np.frombuffer(data, dtype=np.uint8)[rng.integers(0, len(data), size=l.size)].copy()
    c[::9] = rng.integers(0, 256, size=c[::9].size)
    lo, ro = csa.backward_search(l, r, c)
    exp = np.array([o.backward_search(a, b, cc) for a, b, cc in zip(l, r, c)], dtype=np.uint64)
    assert np.array_equal(lo, exp[:, 0]) and np.array_equal(ro, exp[:, 1])


def test_fm_synthetic_text_device_sa_vs_oracle(gpu):
    rng = np.random.default_rng(8)
    words = [bytes(rng.integers(97, 123, size=rng.integers(2, 9), dtype=np.uint8)) for _ in range(300)]
    t = b" ".join(words[i] for i in rng.integers(0, len(words), size=60000))
    o = ol.OCsa(t)
    csa = gpu.csa_wt(text=t)
    arr = np.frombuffer(t, dtype=np.uint8)
    st = rng.integers(0, len(t) - 20, size=20000)
    pats = np.concatenate([arr[s:s + 20] for s in st])
    got = csa.count(pats, 20)
    assert np.array_equal(got, o.count_batch(pats, 20))
    assert (got >= 1).all()
    if ol.have_ref():
        assert np.array_equal(got, ol.RCsa(t).count_batch(pats, 20))


# ---------------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[1]: 2^34-bit vector) — size-independent invariants
# ---------------------------------------------------------------------------------------------------
def test_full_size_properties(gpu):
    import torch
    n = 1 << 34
    g = torch.Generator(device="cuda").manual_seed(42)
    words = torch.randint(-2**63, 2**63 - 1, (n // 64,), device="cuda", dtype=torch.int64, generator=g)
    bv = gpu.bit_vector(words, n)
    # exact ones count from an independent torch computation (byte LUT)
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.uint8, device="cuda")
    total = 0
    per_word = torch.empty(n // 64, dtype=torch.int32, device="cuda")
    step = 1 << 24
    for s in range(0, n // 64, step):
        pw = lut[words[s:s + step].view(torch.uint8).long()].view(-1, 8).sum(dim=1, dtype=torch.int32)
        per_word[s:s + step] = pw
        total += int(pw.sum())
    assert bv.ones() == total
    nq = 1 << 22
    idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    r1, r0 = bv.rank(idx, 1), bv.rank(idx, 0)
    assert bool((r1 + r0 == idx).all())
    # rank at word boundaries equals the torch prefix sum
    widx = torch.randint(0, n // 64 + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    csum = torch.cumsum(per_word.long(), 0)
    expect = torch.where(widx > 0, csum[(widx - 1).clamp(min=0)], torch.zeros_like(widx))
    assert bool((bv.rank(widx * 64, 1) == expect).all())
    del csum, per_word
    # select/rank round trips
    for b, tot in ((1, total), (0, n - total)):
        i = torch.randint(1, tot + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        pos = bv.select(i, b)
        assert bool((bv.rank(pos, b) == i - 1).all())
        assert bool((bv.access(pos) == b).all())
        assert bool((bv.rank(pos + 1, b) == i).all())
    i = torch.arange(1, nq + 1, device="cuda", dtype=torch.int64) * (total // nq)
    pos = bv.select(i, 1)
    assert bool((pos[1:] > pos[:-1]).all())  # sortedness


# ---------------------------------------------------------------------------------------------------
# small / degenerate inputs and larger-than-2^32 positions
# ---------------------------------------------------------------------------------------------------
def test_fm_tiny_texts(gpu):
    for t in (b"", b"a", b"ab", b"aaaa"):
        csa = gpu.csa_wt(text=t)
        o = ol.OCsa(t)
        assert (csa.size(), csa.sigma()) == (o.size(), o.sigma())
        pats = [b"", b"a", b"aa", b"b", b"ab", b"aaaaa"]
        assert list(csa.count_ragged(pats)) == [o.count(p) for p in pats]


def test_rrr_beyond_32_bit_positions(gpu):
    """2^32 + a few bits: select samples are quantised (sel_pshift = 1) and block indices exceed 2^26"""
    import torch
    n = (1 << 32) + 12345
    g = torch.Generator(device="cuda").manual_seed(3)
    nw = (n + 63) // 64
    w = torch.zeros(nw, dtype=torch.int64, device="cuda")
    bits = torch.randint(0, 64, (nw,), device="cuda", generator=g)
    w = (torch.ones_like(bits) << bits) | (torch.ones_like(bits) << ((bits * 7 + 3) % 64))  # ~2 ones per word
    rv = gpu.rrr_vector(w, n)
    bv = gpu.bit_vector(w, n)
    assert rv.ones() == bv.ones()
    nq = 1 << 20
    idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    assert bool((rv.rank(idx, 1) == bv.rank(idx, 1)).all())
    for b, tot in ((1, rv.ones()), (0, n - rv.ones())):
        i = torch.randint(1, tot + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        assert bool((rv.select(i, b) == bv.select(i, b)).all())
    ia = idx.clamp(max=n - 1)
    assert bool((rv.access(ia) == bv.access(ia)).all())


def test_host_and_device_arguments_agree(gpu):
    import torch
    n = 1 << 21
    w = mk(n, 0.5, 1)
    bv = gpu.bit_vector(w, n)
    idx = np.random.default_rng(2).integers(0, n + 1, size=4097, dtype=np.uint64)
    dev = bv.rank(torch.from_numpy(idx.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
    assert np.array_equal(dev, bv.rank(idx, 1))
    assert bv.rank(np.zeros(0, dtype=np.uint64)).size == 0


# ---------------------------------------------------------------------------------------------------
# loaders never trust a stream: truncated or corrupted SDSL bytes give ERR_FORMAT, not a crash or a hang
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fname,kind", [("rnd.200000.7.rrr63.sdsl", "rrr"), ("example01.txt.wt_huff_v5_mcl.sdsl", "wt"),
                                        ("abc_abc_abc.txt.wt_huff_v5_scan.sdsl", "wts"),
                                        ("example01.txt.csa_wt_huff_v5.sdsl", "csa"),
                                        ("example01.txt.csa_fm_huff.sdsl", "fm")])
def test_malformed_streams(gpu, fname, kind):
    blob = gd.sdsl_file(fname)

    def load(b):
        if kind == "rrr":
            return gpu.rrr_vector(sdsl_bytes=b)
        if kind in ("wt", "wts"):
            return gpu.wt_huff(sdsl_bytes=b, select_is_mcl=(kind == "wt"))
        return gpu.csa_wt(sdsl_bytes=b, select_is_mcl=(kind == "csa"))

    load(blob)  # intact
    cuts = sorted(set(list(range(0, min(len(blob), 200), 7)) + [len(blob) * k // 37 for k in range(37)] + [len(blob) - 1]))
    for cut in cuts:
        with pytest.raises(gpu.capi.SdslHipError) as e:
            load(blob[:cut])
        assert e.value.status == gpu.capi.ERR_FORMAT, cut
    rng = np.random.default_rng(len(blob))
    ok = bad = 0
    for _ in range(300):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            b[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            h = load(bytes(b))
            ok += 1
            # whatever was accepted must be safe to query
            if kind == "rrr":
                h.rank(np.array([0, h.size()], dtype=np.uint64))
                h.select(np.array([1, max(1, h.ones())], dtype=np.uint64))
            elif kind in ("wt", "wts"):
                q = np.array([0, h.size() // 2, h.size()], dtype=np.uint64)
                h.rank(q, np.array([97, 98, 0], dtype=np.uint8))
                if h.size():
                    h.access(np.array([0, h.size() - 1], dtype=np.uint64))
            else:
                h.count(np.frombuffer(b"abcaabca", dtype=np.uint8), 4)
        except gpu.capi.SdslHipError as e:
            assert e.status in (gpu.capi.ERR_FORMAT, gpu.capi.ERR_UNSUPPORTED)
            bad += 1
    assert ok + bad == 300


# ---------------------------------------------------------------------------------------------------
# structures BUILT on the GPU serialise to exactly SDSL's bytes (so unmodified SDSL code can load them)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gd.bv_case_names())
def test_rrr_serialize_equals_sdsl_bytes(gpu, name):
    import hashlib
    g = gd.bv_golden()
    w, n = gd.bv_case(name)
    blob = gpu.rrr_vector(w, n).serialize()
    assert hashlib.sha256(blob).hexdigest() == str(g[f"{name}/sha"][4])
    if ol.have_ref():
        assert blob == ol.RRrr(w, n).serialize()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 2015, 2016, 2017, 63 * 32 * 2, 63 * 32 * 2 + 1, 100_000])
@pytest.mark.parametrize("d", [0.5, 0.97, 0.03])
def test_rrr_serialize_vs_oracle_bytes(gpu, n, d):
    w = mk(n, d, n + 11)
    assert gpu.rrr_vector(w, n).serialize() == ol.ORrr(w, n).serialize()


@pytest.mark.parametrize("name", [t for t in gd.TEXTS if t != "empty.txt"])
def test_wt_serialize_equals_sdsl_bytes(gpu, name):
    import hashlib
    g = gd.text_golden()
    blob = gpu.wt_huff(gd.text(name)).serialize()
    assert hashlib.sha256(blob).hexdigest() == str(g[f"{name}/sha"][1])  # the select_support_scan flavour
    # and a tree loaded from SDSL's own bytes writes the same bytes back
    assert gpu.wt_huff(sdsl_bytes=blob, select_is_mcl=False).serialize() == blob


@pytest.mark.parametrize("name", ["example01.txt", "faust.txt"])
def test_fm_built_on_gpu_serializes_to_sdsl_bytes(gpu, name):
    """raw text -> suffix array, BWT, wavelet tree, samples on the GPU -> bytes identical to what the real SDSL's
    construct() + serialize() produced for the FM_HUFF type of the count benchmark (2^20 sampling)"""
    csa = gpu.csa_wt(text=gd.text(name))
    assert csa.serialize(1 << 20, 1 << 20) == gd.sdsl_file(f"{name}.csa_fm_huff.sdsl")
    if ol.have_ref():
        t = gd.text(name)[:5000]
        assert gpu.csa_wt(text=t).serialize(1 << 20, 1 << 20) == ol.RCsa(t, also_fm_huff=True).serialize(1)
    default_type = csa.serialize(32, 64)
    csa.drop_sa()                                      # the samples 32 / 64 stay: the stream of csa_wt<..., 32, 64> can still be written
    assert csa.serialize(32, 64) == default_type
    with pytest.raises(gpu.capi.SdslHipError):
        csa.serialize(1 << 20, 1 << 20)


# ---------------------------------------------------------------------------------------------------
# writers: device structures back as SDSL's own bytes (default types included)
# ---------------------------------------------------------------------------------------------------
def _sha(b):
    import hashlib
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("name", gd.bv_case_names())
def test_bv_supports_serialize_to_sdsl_bytes(gpu, name):
    """bit_vector, rank_support_v5<1>/<0>, select_support_mcl<1>/<0> (both construction paths of the reference, long and
    mini blocks), rank_support_v<1>/<0>: sha256 of the real library's streams"""
    g = gd.bv_golden()
    words, n = gd.bv_case(name)
    bv = gpu.bit_vector(words, n)
    sha = g[f"{name}/sha"]
    more = g[f"{name}/sha_more"]
    # select_support_mcl only looks at positions below size()
    assert _sha(bv.serialize(3)) == str(sha[2]) and _sha(bv.serialize(4)) == str(sha[3])
    # the vector itself and the rank directories see the whole last word; bits above size() are unspecified in SDSL
    # (util::set_random_bits leaves stray ones there), the device holds zeros there
    masked = words[: (n + 63) // 64].copy()
    if n % 64:
        masked[-1] &= np.uint64((1 << (n % 64)) - 1)
    assert bv.serialize(0) == ((1 << 56) | n).to_bytes(8, "little") + masked.tobytes()
    if n % 64 == 0 or name.startswith("CRAFTED"):
        assert _sha(bv.serialize(0)) == str(more[0])
        assert _sha(bv.serialize(1)) == str(sha[0]) and _sha(bv.serialize(2)) == str(sha[1])
        assert _sha(bv.serialize(5)) == str(more[1]) and _sha(bv.serialize(6)) == str(more[2])
    elif ol.have_ref():  # compare with the real library on the clean words
        rb = ol.RBitVector(masked, n)
        for what in (1, 2, 5, 6):
            assert bv.serialize(what) == rb.serialize(what)
    if name in ("CRAFTED-32", "rnd.200000.7"):
        assert bv.serialize(3) == gd.sdsl_file(f"{name}.select_mcl_1.sdsl")
        assert bv.serialize(4) == gd.sdsl_file(f"{name}.select_mcl_0.sdsl")


@pytest.mark.parametrize("name", [t for t in gd.TEXTS if t != "empty.txt"])
def test_wt_and_csa_serialize_default_types(gpu, name):
    g = gd.text_golden()
    data = gd.text(name)
    wt = gpu.wt_huff(data)
    assert _sha(wt.serialize(gpu.capi.LAYOUT_BV_MCL)) == str(g[f"{name}/sha"][0])       # wt_huff<bit_vector, rank_support_v5<>>
    assert _sha(wt.serialize(gpu.capi.LAYOUT_BV_SCAN)) == str(g[f"{name}/sha"][1])
    assert _sha(wt.serialize(gpu.capi.LAYOUT_BV_DEFAULT)) == str(g[f"{name}/sha_default"][0])  # wt_huff<>
    if f"{name}/sha_csa_default" not in g.files:
        return
    csa = gpu.csa_wt(text=data)
    shas = g[f"{name}/sha_csa_default"]
    assert _sha(csa.serialize(32, 64, gpu.capi.LAYOUT_BV_DEFAULT)) == str(shas[0])  # csa_wt<>
    assert _sha(csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL)) == str(shas[1])      # csa_wt<wt_huff<bit_vector, rank_support_v5<>>>
    # and the default stream loads back with its samples
    again = gpu.csa_wt(sdsl_bytes=csa.serialize(32, 64, gpu.capi.LAYOUT_BV_DEFAULT), select_is_mcl=True, sa_dens=32, isa_dens=64)
    idx = g[f"{name}/csa_idx"][:200]
    assert np.array_equal(again.sa(idx), g[f"{name}/csa_sa"][:200])


@pytest.mark.parametrize("name", ["CRAFTED-32", "CRAFTED-SPARSE-1", "rnd.8192.1043", "rnd.200000.7"])
def test_sd_vector_serializes_to_sdsl_bytes(gpu, name):
    words, n = gd.bv_case(name)
    assert gpu.sd_vector(words, n).serialize() == gd.sdsl_file(f"{name}.sd_vector.sdsl")
    g = gd.bv_golden()
    assert gpu.sd_vector(positions=g["sdpos/pos"], n_bits=int(g["sdpos/n"][0])).serialize() == gd.sdsl_file("sdpos.sd_vector.sdsl")


# ---------------------------------------------------------------------------------------------------
# sd_vector<>
# ---------------------------------------------------------------------------------------------------
SD_CASES = ["CRAFTED-32", "CRAFTED-SPARSE-0", "CRAFTED-SPARSE-1", "CRAFTED-BLOCK-1", "rnd.8.17", "rnd.64.222",
            "rnd.8192.1043", "rnd.200000.7"]


def _check_sd(sd, g, name, n):
    idx = g[f"{name}/idx"]
    assert sd.size() == n
    assert np.array_equal(sd.rank(idx, 1), g[f"{name}/sd_rank1"])
    assert np.array_equal(sd.rank(idx, 0), idx - g[f"{name}/sd_rank1"])
    for b in (0, 1):
        si = g[f"{name}/sel{b}_i"][:600]
        if si.size:
            assert np.array_equal(sd.select(si, b), g[f"{name}/sd_sel{b}"])
    ai = idx[idx < n]
    if ai.size:
        assert np.array_equal(sd.access(ai), g[f"{name}/sd_acc"])
    # outside the domain
    bad = np.array([n + 1, 2**63], dtype=np.uint64)
    assert np.all(sd.rank(bad, 1) == NPOS) and np.all(sd.access(np.array([n, n + 7], dtype=np.uint64)) == 0xFF)
    tot1 = int(g[f"{name}/total1"][0])
    assert np.all(sd.select(np.array([0, tot1 + 1], dtype=np.uint64), 1) == NPOS)
    assert np.all(sd.select(np.array([0, n - tot1 + 1], dtype=np.uint64), 0) == NPOS)


@pytest.mark.parametrize("name", SD_CASES)
@pytest.mark.parametrize("how", ["bits", "positions", "stream"])
def test_sd_vector_golden(gpu, name, how):
    g = gd.bv_golden()
    words, n = gd.bv_case(name)
    if how == "bits":
        sd = gpu.sd_vector(words, n)
    elif how == "positions":
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
        sd = gpu.sd_vector(positions=np.flatnonzero(bits).astype(np.uint64), n_bits=n)
    else:
        if not os.path.exists(os.path.join(gd.GOLDEN, "sdsl", f"{name}.sd_vector.sdsl")):
            pytest.skip("no stream fixture for this case")
        blob = gd.sdsl_file(f"{name}.sd_vector.sdsl")
        sd = gpu.sd_vector(sdsl_bytes=blob + b"tail")
        assert sd.consumed == len(blob)
        for cut in (5, len(blob) // 2, len(blob) - 3):
            with pytest.raises(gpu.capi.SdslHipError):
                gpu.sd_vector(sdsl_bytes=blob[:cut])
    assert sd.ones() == int(g[f"{name}/total1"][0])
    assert sd.low_width() == ol.OSd(words, n).wl()
    _check_sd(sd, g, name, n)


def test_sd_vector_large_universe(gpu):
    g = gd.bv_golden()
    pos, N = g["sdpos/pos"], int(g["sdpos/n"][0])
    for sd in (gpu.sd_vector(positions=pos, n_bits=N), gpu.sd_vector(sdsl_bytes=gd.sdsl_file("sdpos.sd_vector.sdsl"))):
        assert (sd.size(), sd.ones()) == (N, pos.size)
        assert np.array_equal(sd.rank(g["sdpos/idx"], 1), g["sdpos/rank1"])
        assert np.array_equal(sd.select(g["sdpos/sel1_i"], 1), g["sdpos/sel1"])
        assert np.array_equal(sd.select(g["sdpos/sel0_i"], 0), g["sdpos/sel0"])
        qi = g["sdpos/idx"]
        assert np.array_equal(sd.access(qi[qi < N]), g["sdpos/acc"])
    with pytest.raises(gpu.capi.SdslHipError):  # not strictly increasing
        gpu.sd_vector(positions=np.array([5, 5, 9], dtype=np.uint64), n_bits=100)
    with pytest.raises(gpu.capi.SdslHipError):  # beyond the size
        gpu.sd_vector(positions=np.array([5, 100], dtype=np.uint64), n_bits=100)


def test_sd_vector_million_ones_in_2_40(gpu):
    """2^22 ones over a universe of 2^40: select_1(rank_1(p) + 1) == p on members, rank is monotone and exact against
    numpy's searchsorted, select_0 inverts rank_0"""
    rng = np.random.default_rng(17)
    N = 1 << 40
    pos = np.unique(rng.integers(0, N, 1 << 22).astype(np.uint64))
    sd = gpu.sd_vector(positions=pos, n_bits=N)
    assert sd.ones() == pos.size and sd.size() == N
    q = np.concatenate([rng.integers(0, N + 1, 500000).astype(np.uint64), pos[:100000], pos[:100000] + np.uint64(1)])
    r = sd.rank(q, 1)
    assert np.array_equal(r, np.searchsorted(pos, q, side="left").astype(np.uint64))
    i = rng.integers(1, pos.size + 1, 300000).astype(np.uint64)
    assert np.array_equal(sd.select(i, 1), pos[(i - np.uint64(1)).astype(np.int64)])
    assert np.array_equal(sd.access(pos[:50000]), np.ones(50000, np.uint8))
    z = rng.integers(1, N - pos.size + 1, 20000).astype(np.uint64)
    pz = sd.select(z, 0)
    assert np.array_equal(sd.rank(pz, 0) + np.uint64(1), z) and np.all(sd.access(pz) == 0)


# ---------------------------------------------------------------------------------------------------
# two-bit patterns: rank_support_v5<pat,2> / select_support_mcl<pat,2>
# ---------------------------------------------------------------------------------------------------
PATTERNS = [(10, 2), (1, 2), (0, 2), (11, 2)]  # SDSL's template arguments <10,2> <01,2> <00,2> <11,2>


@pytest.mark.parametrize("name", ["CRAFTED-32", "CRAFTED-MAT-SELECT", "rnd.64.222", "rnd.8192.1043", "rnd.200000.7",
                                  "rnd.1000000.815"])
def test_two_bit_patterns_golden(gpu, name):
    g = gd.bv_golden()
    words, n = gd.bv_case(name)
    idx = g[f"{name}/idx"]
    for pat, targ in enumerate(PATTERNS):
        bv = gpu.bit_vector(words, n, select0=False, pattern=targ)
        assert bv.size() == n
        assert np.array_equal(bv.rank(idx, 1), g[f"{name}/pat{pat}_rank"])
        si = g[f"{name}/pat{pat}_sel_i"]
        if si.size:
            assert np.array_equal(bv.select(si, 1), g[f"{name}/pat{pat}_sel"])
            tot = int(si[-1])
            assert int(bv.select(np.array([tot + 1], dtype=np.uint64), 1)[0]) == int(NPOS)


def test_two_bit_patterns_large(gpu):
    """2^28 bits against the occurrence-vector model (pinned to the real library by the CPU suite)"""
    n = (1 << 28) + 37
    words = ol.set_random_bits(n, 99)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
    rng = np.random.default_rng(1)
    idx = np.concatenate([rng.integers(0, n + 1, 200000), [0, n, 64, 63, 65]]).astype(np.uint64)
    for pat, targ in enumerate(PATTERNS):
        d = ol.pattern_bits(bits, pat)
        cum = np.concatenate([[0], np.cumsum(d, dtype=np.int64)]).astype(np.uint64)
        bv = gpu.bit_vector(words, n, select0=False, pattern=targ)
        assert bv.ones() == int(cum[-1])
        assert np.array_equal(bv.rank(idx, 1), cum[idx.astype(np.int64)])
        i = rng.integers(1, int(cum[-1]) + 1, 200000).astype(np.uint64)
        pos = bv.select(i, 1)
        assert np.array_equal(bv.rank(pos, 1) + np.uint64(1), i) and bool(np.all(d[pos.astype(np.int64)] == 1))
    with pytest.raises(gpu.capi.SdslHipError):
        gpu.bit_vector(words, n, pattern=(12, 2))


# ---------------------------------------------------------------------------------------------------
# other wt_pc shapes: wt_blcd / wt_hutu streams load as they are, wt_blcd can be built on the device
# ---------------------------------------------------------------------------------------------------
def _shape_text(name):
    return gd.text("faust.txt")[:60000] if name == "faust60k" else gd.text(name)


def _check_wt_against_oracle(wt, data, seed):
    """rank / access / inverse_select / select do not depend on the tree shape: the Huffman-shaped oracle answers"""
    o = ol.OWt(data)
    n = len(data)
    arr = np.frombuffer(data, dtype=np.uint8)
    assert (wt.size(), wt.sigma()) == (n, o.sigma())
    i = np.concatenate([ol.mt19937_64(4000, seed) % np.uint64(n + 1), np.array([0, n], dtype=np.uint64)])
    c = (ol.mt19937_64(i.size, seed + 1) % np.uint64(256)).astype(np.uint8)
    c[::2] = arr[(ol.mt19937_64(i.size, seed + 2) % np.uint64(n)).astype(np.int64)][::2]
    assert np.array_equal(wt.rank(i, c), o.rank(i, c))
    j = ol.mt19937_64(3000, seed + 3) % np.uint64(n)
    assert np.array_equal(wt.access(j), arr[j.astype(np.int64)])
    r, cc = wt.inverse_select(j)
    assert np.array_equal(cc, arr[j.astype(np.int64)]) and np.array_equal(r, o.rank(j, cc))
    assert np.array_equal(wt.select(r + np.uint64(1), cc), j)


@pytest.mark.parametrize("name", ["example01.txt", "faust60k"])
@pytest.mark.parametrize("shape", ["wt_blcd", "wt_hutu"])
def test_wt_other_shapes_load(gpu, name, shape):
    blob = gd.sdsl_file(f"{name}.{shape}.sdsl")  # wt_blcd<> / wt_hutu<> of the real library (rank_support_v, mcl selects)
    wt = gpu.wt_huff(sdsl_bytes=blob, select_is_mcl=True)
    assert wt.consumed == len(blob)
    _check_wt_against_oracle(wt, _shape_text(name), 70)


@pytest.mark.parametrize("name", ["example01.txt", "faust60k"])
@pytest.mark.parametrize("rrr", [False, True])
def test_wt_blcd_built_on_gpu(gpu, name, rrr):
    data = _shape_text(name)
    wt = gpu.wt_huff(data, balanced=True, rrr=rrr)
    _check_wt_against_oracle(wt, data, 80)
    if not rrr:  # the same bytes as the real wt_blcd<bit_vector, rank_support_v5<>, scan, scan>
        assert wt.serialize() == gd.sdsl_file(f"{name}.wt_blcd_v5_scan.sdsl")
    # balanced codes: every symbol has depth ceil(log2 sigma) or one less
    lens = wt.code_lengths()
    present = lens[lens > 0]
    assert present.max() - present.min() <= 1 and present.max() == int(np.ceil(np.log2(wt.sigma())))


@pytest.mark.parametrize("name", ["example01.txt", "faust60k"])
def test_wt_hutu_and_blcd_default_types_built_on_gpu(gpu, name):
    """Hu-Tucker and balanced trees built on the device, written as wt_hutu<> / wt_blcd<> with SDSL's default arguments
    (rank_support_v, select_support_mcl): the real library's files byte for byte"""
    data = _shape_text(name)
    hutu = gpu.wt_huff(data, hutu=True)
    _check_wt_against_oracle(hutu, data, 85)
    assert hutu.serialize(gpu.capi.LAYOUT_BV_DEFAULT) == gd.sdsl_file(f"{name}.wt_hutu.sdsl")
    blcd = gpu.wt_huff(data, balanced=True)
    assert blcd.serialize(gpu.capi.LAYOUT_BV_DEFAULT) == gd.sdsl_file(f"{name}.wt_blcd.sdsl")
    _check_wt_against_oracle(gpu.wt_huff(data, hutu=True, rrr=True), data, 86)
    with pytest.raises(gpu.capi.SdslHipError):
        gpu.wt_huff(data, hutu=True, balanced=True)


def test_wt_hutu_tie_breaks_match_the_reference(gpu):
    """alphabets full of equal weights: the merge order of the reference's master queue decides the code lengths"""
    if not ol.have_ref():
        pytest.skip("needs oracle/_ref")
    rng = np.random.default_rng(3)
    for trial in range(60):
        sigma = int(rng.integers(1, 60))
        syms = np.sort(rng.choice(np.arange(1, 256), sigma, replace=False))
        w = rng.integers(1, 4, sigma) if trial % 2 else np.maximum(1, (300 / np.arange(1, sigma + 1)).astype(int))
        text = np.concatenate([np.full(int(k), c, dtype=np.uint8) for c, k in zip(syms, w)])
        rng.shuffle(text)
        wt = gpu.wt_huff(text, hutu=True)
        assert wt.serialize(gpu.capi.LAYOUT_BV_DEFAULT) == ol.ref_wt_shape_bytes(text.tobytes(), 2, 0)


@pytest.mark.parametrize("name", ["example01.txt", "faust60k"])
def test_fm_blcd_built_on_gpu(gpu, name):
    data = _shape_text(name)
    csa = gpu.csa_wt(text=data, balanced=True)
    assert csa.serialize(32, 64) == gd.sdsl_file(f"{name}.csa_wt_blcd_v5_scan.sdsl")
    ref = gpu.csa_wt(text=data)
    arr = np.frombuffer(data, dtype=np.uint8)
    m = 5
    st = ol.mt19937_64(2000, 90) % np.uint64(len(data) - m)
    pats = np.concatenate([arr[int(x):int(x) + m] for x in st])
    assert np.array_equal(csa.count(pats, m), ref.count(pats, m))
    o1, p1 = csa.locate(pats, m)
    o2, p2 = ref.locate(pats, m)
    assert np.array_equal(o1, o2) and np.array_equal(p1, p2)
    loaded = gpu.csa_wt(sdsl_bytes=gd.sdsl_file(f"{name}.csa_wt_blcd_v5_scan.sdsl"), select_is_mcl=False, sa_dens=32,
                        isa_dens=64)
    assert np.array_equal(loaded.count(pats, m), ref.count(pats, m))
    o3, p3 = loaded.locate(pats, m)
    assert np.array_equal(p3, p2)


# ---------------------------------------------------------------------------------------------------
# the rest of the csa_wt API: SA / ISA / LF / psi access, extract, locate
# ---------------------------------------------------------------------------------------------------
def _check_locate_api(csa, g, name, n_text):
    idx = g[f"{name}/csa_idx"]
    assert np.array_equal(csa.sa(idx), g[f"{name}/csa_sa"])
    assert np.array_equal(csa.isa(idx), g[f"{name}/csa_isa"])
    assert np.array_equal(csa.lf(idx), g[f"{name}/csa_lf"])
    assert np.array_equal(csa.psi(idx), g[f"{name}/csa_psi"])
    N = csa.size()
    bad = np.array([N, N + 5, 2**63], dtype=np.uint64)
    for fn in (csa.sa, csa.isa, csa.lf, csa.psi):
        assert np.all(fn(bad) == NPOS)
    eb, ee = g[f"{name}/ext_b"], g[f"{name}/ext_e"]
    off, text = csa.extract(eb, ee)
    assert text.tobytes() == g[f"{name}/ext_text"].tobytes()
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(ee - eb + np.uint64(1))]).astype(np.uint64))
    # one long range (cut at the ISA samples into parallel pieces): the whole indexed sequence, sentinel included
    offw, whole = csa.extract(np.array([0, 1], dtype=np.uint64), np.array([N - 1, N - 1], dtype=np.uint64))
    full = np.frombuffer(gd.text(name) if name in gd.TEXTS or name == "faust.txt" else b"", dtype=np.uint8)
    if full.size == n_text:
        assert whole[:N].tobytes() == full.tobytes() + b"\0" and whole[N:].tobytes() == full.tobytes()[1:] + b"\0"
    # queries outside the precondition (begin > end, end >= size) yield nothing
    off2, text2 = csa.extract(np.array([5, 0, 0], dtype=np.uint64), np.array([4, N, min(3, N - 1)], dtype=np.uint64))
    assert list(off2[:3]) == [0, 0, 0] and int(off2[3]) == text2.size == min(3, N - 1) + 1
    for m in (2, 4, 20):
        if f"{name}/loc_n{m}" not in g.files:
            continue
        k = int(g[f"{name}/loc_n{m}"][0])
        pats = g[f"{name}/pat{m}"][: k * m]
        off, pos = csa.locate(pats, m)
        assert np.array_equal(off, g[f"{name}/loc_off{m}"]) and np.array_equal(pos, g[f"{name}/loc_pos{m}"])
    # empty batch
    off, pos = csa.sa_range(np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    assert off.size == 1 and int(off[0]) == 0 and pos.size == 0


@pytest.mark.parametrize("name", FM_TEXTS)
@pytest.mark.parametrize("how", ["full_sa", "dropped", "rrr_dropped"])
def test_fm_locate_extract_golden(gpu, name, how):
    g = _fm_cases(name)
    data = gd.text(name)
    csa = gpu.csa_wt(text=data, rrr=how.startswith("rrr"))
    assert csa.sampling() == (0, 0, True)
    if how != "full_sa":
        csa.drop_sa()  # keeps SDSL's default samples: the answers come from LF walks now
        assert csa.sampling() == (32, 64, False)
    _check_locate_api(csa, g, name, len(data))


@pytest.mark.parametrize("name", ["example01.txt", "faust.txt"])
@pytest.mark.parametrize("which,rrr,dens", [("csa_wt_huff_v5", False, (32, 64)), ("csa_wt_huff_rrr63", True, (32, 64)),
                                            ("csa_fm_huff", False, (1 << 20, 1 << 20))])
def test_fm_locate_extract_on_loaded_stream(gpu, name, which, rrr, dens):
    """indexes written by the real library, loaded with their type's densities"""
    g = _fm_cases(name)
    blob = gd.sdsl_file(f"{name}.{which}.sdsl")
    csa = gpu.csa_wt(sdsl_bytes=blob, rrr=rrr, select_is_mcl=(which == "csa_wt_huff_v5"), sa_dens=dens[0],
                     isa_dens=dens[1])
    assert csa.sampling() == (dens[0], dens[1], False)
    if dens[0] > 32 and name == "faust.txt":
        # one SA sample for the whole text: every SA access walks up to n LF steps — keep the check small
        idx = g[f"{name}/csa_idx"][:64]
        assert np.array_equal(csa.sa(idx), g[f"{name}/csa_sa"][:64])
        assert np.array_equal(csa.isa(idx), g[f"{name}/csa_isa"][:64])
        return
    _check_locate_api(csa, g, name, len(gd.text(name)))
    # wrong densities are refused (sample vector sizes do not fit), no samples -> UNSUPPORTED
    if name == "faust.txt":  # (a text shorter than the density has one sample whatever the density)
        with pytest.raises(gpu.capi.SdslHipError):
            gpu.csa_wt(sdsl_bytes=blob, rrr=rrr, select_is_mcl=(which == "csa_wt_huff_v5"), sa_dens=dens[0] * 2 + 1,
                       isa_dens=dens[1])
    plain = gpu.csa_wt(sdsl_bytes=blob, rrr=rrr, select_is_mcl=(which == "csa_wt_huff_v5"))
    with pytest.raises(gpu.capi.SdslHipError) as e:
        plain.sa(np.array([0], dtype=np.uint64))
    assert e.value.status == gpu.capi.ERR_UNSUPPORTED


def test_fm_locate_c_entry_point(gpu):
    """sdsl_hip_fm_locate_batch (size query, then fill) equals interval + sa_range"""
    import ctypes as C
    g = gd.text_golden()
    name, m = "faust.txt", 4
    csa = gpu.csa_wt(text=gd.text(name))
    k = int(g[f"{name}/loc_n{m}"][0])
    pats = np.ascontiguousarray(g[f"{name}/pat{m}"][: k * m])
    L = gpu.capi.lib()
    total = C.c_uint64(0)
    off = np.empty(k + 1, dtype=np.uint64)
    gpu.capi.check(L.sdsl_hip_fm_locate_batch(csa._h, pats.ctypes.data, m, k, off.ctypes.data, None, 0, C.byref(total),
                                              None))
    assert total.value == g[f"{name}/loc_pos{m}"].size
    pos = np.empty(total.value, dtype=np.uint64)
    assert L.sdsl_hip_fm_locate_batch(csa._h, pats.ctypes.data, m, k, None, pos.ctypes.data, total.value - 1,
                                      C.byref(total), None) == gpu.capi.ERR_INVALID
    gpu.capi.check(L.sdsl_hip_fm_locate_batch(csa._h, pats.ctypes.data, m, k, None, pos.ctypes.data, total.value,
                                              C.byref(total), None))
    assert np.array_equal(off, g[f"{name}/loc_off{m}"]) and np.array_equal(pos, g[f"{name}/loc_pos{m}"])


def test_fm_locate_roundtrip_large(gpu):
    """32 MiB text: SA[ISA[i]] == i, psi(lf(i)) == i, extract == the text, every located position starts the pattern;
    whole-SA answers == sampled-walk answers"""
    rng = np.random.default_rng(11)
    n = 1 << 25
    data = rng.choice(np.arange(1, 100, dtype=np.uint8), size=n, p=_zipf(99)).astype(np.uint8)
    csa = gpu.csa_wt(text=data)
    N = csa.size()
    pos = rng.integers(0, N, 100000).astype(np.uint64)
    isa = csa.isa(pos)
    assert np.array_equal(csa.sa(isa), pos)
    j = rng.integers(0, N, 100000).astype(np.uint64)
    assert np.array_equal(csa.psi(csa.lf(j)), j) and np.array_equal(csa.lf(csa.psi(j)), j)
    b = rng.integers(0, n - 300, 5000).astype(np.uint64)
    e = b + rng.integers(0, 256, 5000).astype(np.uint64)
    off, text = csa.extract(b, e)
    for q in (0, 17, 4999):
        assert text[int(off[q]):int(off[q + 1])].tobytes() == data[int(b[q]):int(e[q]) + 1].tobytes()
    flat = np.concatenate([data[int(x):int(y) + 1] for x, y in zip(b[:500], e[:500])])
    assert np.array_equal(text[: flat.size], flat)
    off1, big = csa.extract(np.array([12345], dtype=np.uint64), np.array([12345 + (1 << 22)], dtype=np.uint64))
    assert np.array_equal(big, data[12345:12345 + (1 << 22) + 1])  # 4 MiB in one range: 65537 parallel pieces
    m = 6
    st = rng.integers(0, n - m, 20000)
    pats = np.concatenate([data[s:s + m] for s in st])
    off, occ = csa.locate(pats, m)
    assert np.array_equal(np.diff(off.astype(np.int64)).astype(np.uint64), csa.count(pats, m))
    which = np.repeat(np.arange(20000), np.diff(off.astype(np.int64)))
    sample = rng.integers(0, occ.size, 50000)
    for t in range(m):
        assert np.array_equal(data[(occ[sample] + np.uint64(t)).astype(np.int64)], pats[which[sample] * m + t])
    sa_full = csa.sa(j)
    csa.drop_sa()
    assert np.array_equal(csa.sa(j[:20000]), sa_full[:20000])
    off2, occ2 = csa.locate(pats[: 2000 * m], m)
    assert np.array_equal(occ2, occ[: int(off[2000])])


# ---------------------------------------------------------------------------------------------------
# wt_huff<rrr_vector<63>> and csa_wt over it: same answers as the plain tree, rrr-compressed on the device
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gd.TEXTS)
def test_wt_rrr_golden(gpu, name):
    g = gd.text_golden()
    data = gd.text(name)
    wt = gpu.wt_huff(data, rrr=True)
    n, sigma, bvs = (int(x) for x in g[f"{name}/meta"])
    assert (wt.size(), wt.sigma(), wt.bv_size()) == (n, sigma, bvs)
    assert np.array_equal(wt.rank(g[f"{name}/rank_i"], g[f"{name}/rank_c"]), g[f"{name}/rank"])
    assert np.array_equal(wt.rank(np.full(256, n, dtype=np.uint64), np.arange(256, dtype=np.uint8)),
                          g[f"{name}/rank_full"])
    assert int(wt.rank(np.array([n + 1], dtype=np.uint64), np.array([97], dtype=np.uint8))[0]) == int(NPOS)
    if n:
        ai = g[f"{name}/acc_i"]
        assert np.array_equal(wt.access(ai), g[f"{name}/acc"])
        r, c = wt.inverse_select(ai)
        assert np.array_equal(r, g[f"{name}/invsel_rank"]) and np.array_equal(c, g[f"{name}/acc"])
        assert np.array_equal(wt.select(g[f"{name}/sel_i"], g[f"{name}/sel_c"]), g[f"{name}/sel"])
        # absent symbol -> size(); i outside [1, occ] -> NPOS: the same contract as the plain tree
        absent = [c for c in range(256) if g[f"{name}/rank_full"][c] == 0]
        if absent:
            got = wt.select(np.array([1], dtype=np.uint64), np.array([absent[0]], dtype=np.uint8))
            assert int(got[0]) == n
        c0 = int(data[0])
        occ = int(g[f"{name}/rank_full"][c0])
        got = wt.select(np.array([0, occ + 1, occ], dtype=np.uint64), np.full(3, c0, dtype=np.uint8))
        assert int(got[0]) == int(NPOS) and int(got[1]) == int(NPOS) and int(got[2]) < n


@pytest.mark.parametrize("name", ["example01.txt", "faust.txt"])
def test_wt_rrr_streams(gpu, name):
    g = gd.text_golden()
    blob = gd.sdsl_file(f"{name}.wt_huff_rrr63.sdsl")
    built = gpu.wt_huff(gd.text(name), rrr=True)
    assert built.serialize() == blob  # built on the GPU == real SDSL's wt_huff<rrr_vector<63>> bytes
    loaded = gpu.wt_huff(sdsl_bytes=blob, rrr=True)
    assert loaded.consumed == len(blob)
    assert np.array_equal(loaded.rank(g[f"{name}/rank_i"], g[f"{name}/rank_c"]), g[f"{name}/rank"])
    assert np.array_equal(loaded.access(g[f"{name}/acc_i"]), g[f"{name}/acc"])
    assert np.array_equal(loaded.select(g[f"{name}/sel_i"], g[f"{name}/sel_c"]), g[f"{name}/sel"])
    assert loaded.serialize() == blob
    for cut in (17, len(blob) // 3, len(blob) - 1):
        with pytest.raises(gpu.capi.SdslHipError):
            gpu.wt_huff(sdsl_bytes=blob[:cut], rrr=True)


def _zipf(k):
    w = 1.0 / np.arange(1, k + 1)
    return w / w.sum()


def test_wt_rrr_select_roundtrip_large(gpu):
    """select(rank(i, wt[i]) + 1, wt[i]) == i on a 32 MiB text, both backends agreeing"""
    rng = np.random.default_rng(5)
    n = 1 << 25
    data = rng.choice(np.arange(1, 200, dtype=np.uint8), size=n, p=_zipf(199)).astype(np.uint8)
    wt = gpu.wt_huff(data, rrr=True)
    pos = rng.integers(0, n, 200000).astype(np.uint64)
    r, c = wt.inverse_select(pos)
    assert np.array_equal(c, data[pos.astype(np.int64)])
    assert np.array_equal(wt.select(r + np.uint64(1), c), pos)
    plain = gpu.wt_huff(data)
    i = rng.integers(1, 1000, 50000).astype(np.uint64)
    cc = data[rng.integers(0, n, 50000)]
    assert np.array_equal(plain.select(i, cc), wt.select(i, cc))


@pytest.mark.parametrize("name", FM_TEXTS)
def test_fm_rrr_golden(gpu, name):
    g = _fm_cases(name)
    data = gd.text(name)
    csa = gpu.csa_wt(text=data, rrr=True)
    assert [csa.size(), csa.sigma()] == [int(x) for x in g[f"{name}/csa_meta"]]
    for m in (1, 2, 4, 20):
        if f"{name}/pat{m}" not in g.files:
            continue
        pats = g[f"{name}/pat{m}"]
        assert np.array_equal(csa.count(pats, m), g[f"{name}/count{m}"])
        l, r = csa.interval(pats, m)
        assert np.array_equal(l, g[f"{name}/ival_l{m}"]) and np.array_equal(r, g[f"{name}/ival_r{m}"])


@pytest.mark.parametrize("name", ["example01.txt", "faust.txt"])
def test_fm_rrr_streams(gpu, name):
    g = gd.text_golden()
    blob = gd.sdsl_file(f"{name}.csa_wt_huff_rrr63.sdsl")
    built = gpu.csa_wt(text=gd.text(name), rrr=True)
    assert built.serialize(32, 64) == blob  # csa_wt<wt_huff<rrr_vector<63>>, 32, 64> of the real library
    loaded = gpu.csa_wt(sdsl_bytes=blob, rrr=True)
    for m in (1, 4, 20):
        if f"{name}/pat{m}" in g.files:
            assert np.array_equal(loaded.count(g[f"{name}/pat{m}"], m), g[f"{name}/count{m}"])


def test_fm_rrr_random_vs_oracle(gpu):
    rng = np.random.default_rng(21)
    words = [bytes(rng.integers(97, 123, size=rng.integers(2, 9), dtype=np.uint8)) for _ in range(200)]
    t = b" ".join(words[i] for i in rng.integers(0, len(words), size=40000))
    o = ol.OCsa(t)
    csa = gpu.csa_wt(text=t, rrr=True)
    arr = np.frombuffer(t, dtype=np.uint8)
    for m in (3, 20, 33):
        st = rng.integers(0, len(t) - m, size=100_000)
        pats = np.concatenate([arr[s:s + m] for s in st])
        pats[::13] = rng.integers(1, 256, size=pats[::13].size)
        assert np.array_equal(csa.count(pats, m), o.count_batch(pats, m))  # >= 2^16 patterns: suffix-ordered path
    n = o.size()
    l = rng.integers(0, n, size=5000, dtype=np.uint64)
    r = np.minimum(l + rng.integers(0, 3000, size=l.size, dtype=np.uint64), np.uint64(n - 1))
    c = arr[rng.integers(0, len(t), size=l.size)].copy()
    c[::7] = rng.integers(0, 256, size=c[::7].size)
    lo, ro = csa.backward_search(l, r, c)
    exp = np.array([o.backward_search(a, b, cc) for a, b, cc in zip(l, r, c)], dtype=np.uint64)
    assert np.array_equal(lo, exp[:, 0]) and np.array_equal(ro, exp[:, 1])
    qi = rng.integers(0, len(t) + 2, size=50_000, dtype=np.uint64)
    qc = rng.integers(0, 256, size=qi.size, dtype=np.uint8)
    assert np.array_equal(csa.wavelet_tree.rank(qi, qc), o.wt().rank(qi, qc))


@pytest.mark.parametrize("shape", ["dense", "sparse", "clustered", "one_long_gap"])
def test_select_mcl_writer_on_many_superblocks(gpu, shape):
    """the select_support_mcl writer cuts the argument sequence into stretches that worker threads handle on their own
    (bv_serialize.cpp): thousands of superblocks, long and mini blocks mixed, byte for byte the real library's stream"""
    if not ol.have_ref():
        pytest.skip("needs the compiled reference")
    rng = np.random.default_rng(17)
    n = (1 << 24) + 777
    bits = np.zeros(n, dtype=np.uint8)
    if shape == "dense":
        bits[rng.random(n) < 0.5] = 1
    elif shape == "sparse":
        bits[rng.random(n) < 0.003] = 1
    elif shape == "clustered":
        for s in rng.integers(0, n - 70000, 40):
            bits[s:s + int(rng.integers(1000, 70000))] = rng.random(1) < 0.5 or 1
        bits[rng.random(n) < 0.0005] = 1
    else:
        bits[: 1 << 20] = rng.random(1 << 20) < 0.7
        bits[-(1 << 20):] = rng.random(1 << 20) < 0.7
    words = np.packbits(np.concatenate([bits, np.zeros((-n) % 64, np.uint8)]), bitorder="little").view(np.uint64)
    bv = gpu.bit_vector(words, n)
    rb = ol.RBitVector(words, n)
    assert bv.serialize(3) == rb.serialize(3)
    assert bv.serialize(4) == rb.serialize(4)


def test_count_from_host_arrays_is_pipelined_and_equal(gpu):
    """>= 2^22 patterns with patterns and answers in host memory travel in pieces over several streams (fm.hip,
    common.hpp host_pipeline_bytes): same answers as one device-resident batch, and as the oracle on a sample"""
    import torch
    text = gd.text("faust.txt")
    csa = gpu.csa_wt(text=text)
    arr = np.frombuffer(text, dtype=np.uint8)
    rng = np.random.default_rng(8)
    n, m = (1 << 22) + 777, 7
    st = rng.integers(0, len(text) - m, n)
    pats = arr[(st[:, None] + np.arange(m)[None, :]).reshape(-1)].copy()
    pats[: 50 * m] = rng.integers(1, 256, 50 * m).astype(np.uint8)  # some that (mostly) do not occur
    host = csa.count(pats, m)
    dev = csa.count(torch.from_numpy(pats).cuda(), m).cpu().numpy().view(np.uint64)
    assert np.array_equal(host, dev)
    o = ol.OCsa(text)
    pick = np.concatenate([np.arange(200), rng.integers(0, n, 2000)])
    sample = np.concatenate([pats[q * m:(q + 1) * m] for q in pick])
    assert np.array_equal(host[pick], o.count_batch(sample, m))


def test_wt_rank_from_host_arrays_is_pipelined_and_equal(gpu):
    """>= 2^23 (i, c) pairs with all three arrays in host memory are answered in pieces over several streams"""
    import torch
    text = gd.text("faust.txt")
    wt = gpu.wt_huff(text=text)
    arr = np.frombuffer(text, dtype=np.uint8)
    rng = np.random.default_rng(21)
    n = (1 << 23) + 4321
    i = rng.integers(0, len(text) + 2, n).astype(np.uint64)
    c = arr[rng.integers(0, len(text), n)].copy()
    c[:100] = rng.integers(0, 256, 100).astype(np.uint8)
    host = wt.rank(i, c)
    dev = wt.rank(torch.from_numpy(i.view(np.int64)).cuda(), torch.from_numpy(c).cuda()).cpu().numpy().view(np.uint64)
    assert np.array_equal(host, dev)
    o = ol.OWt(text)
    pick = np.concatenate([np.arange(300), rng.integers(0, n, 3000)])
    assert np.array_equal(host[pick], o.rank(i[pick], c[pick]))
