"""CPU: randomized cross-check of the C restatement against the real sdsl-lite compiled from
/root/reference (oracle/_ref/libsdsl_ref.so).  Skipped where that build is absent; the committed
golden vectors (test_oracle_golden.py) carry the pin in that case."""
import numpy as np
import pytest

import golden_data as gd
import oracle_lib as ol

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref (real sdsl-lite build) not present")


def mk(n, d, seed):
    r = np.random.default_rng(seed)
    nw = (n + 63) // 64
    if d == 0.5:
        return r.integers(0, 2**64, size=nw, dtype=np.uint64)
    bits = (r.random(nw * 64) < d).astype(np.uint8)
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


@pytest.mark.parametrize("n", [0, 1, 64, 385, 2049, 6144, 99999, 100000, 200000, 63 * 32 * 3])
@pytest.mark.parametrize("d", [0.5, 0.05, 0.97, 0.0, 1.0])
def test_bitvector_structures(n, d):
    rng = np.random.default_rng(n)
    w = mk(n, d, n + int(d * 1000))
    o, r = ol.OBitVector(w, n), ol.RBitVector(w, n)
    idx = np.arange(n + 1, dtype=np.uint64) if n <= 10000 else rng.integers(0, n + 1, size=5000, dtype=np.uint64)
    for b in (0, 1):
        assert np.array_equal(o.rank(idx, b), r.rank(idx, b))
        assert o.serialize_rank(b) == r.serialize(1 if b else 2)
        ac = o.arg_cnt(b)
        if ac:
            ii = rng.integers(1, ac + 1, size=5000, dtype=np.uint64)
            assert np.array_equal(o.select(ii, b), r.select(ii, b))
        assert o.serialize_select(b) == r.serialize(3 if b else 4)
    orr, rr = ol.ORrr(w, n), ol.RRrr(w, n)
    assert orr.serialize() == rr.serialize()
    for b in (0, 1):
        assert np.array_equal(orr.rank(idx, b), rr.rank(idx, b))
        tot = int(rr.rank(np.array([n], dtype=np.uint64), b)[0])
        ii = rng.integers(1, tot + 2, size=3000, dtype=np.uint64)
        assert np.array_equal(orr.select(ii, b), rr.select(ii, b))
    if n:
        ia = idx[idx < n][:1000]
        assert np.array_equal(orr.access(ia), rr.access(ia))


def test_bits_primitives():
    rng = np.random.default_rng(1)
    L, R = ol.oracle().L, ol.ref().L
    for x in rng.integers(1, 2**64, size=300, dtype=np.uint64):
        x = int(x)
        assert L.orc_hi(x) == R.ref_bits_hi(x)
        for i in range(1, bin(x).count("1") + 1):
            assert L.orc_sel(x, i) == R.ref_bits_sel(x, i)


@pytest.mark.parametrize("name", ["example01.txt", "100a.txt", "faust.txt", "rnd"])
def test_wt_and_csa(name):
    rng = np.random.default_rng(3)
    t = bytes(rng.integers(1, 256, size=30000, dtype=np.uint8)) if name == "rnd" else gd.text(name)
    o, r = ol.OWt(t), ol.RWt(t)
    assert o.serialize(1) == r.serialize(1) and o.serialize(0) == r.serialize(0)
    n = len(t)
    i = rng.integers(0, n + 1, size=3000, dtype=np.uint64)
    c = rng.integers(0, 256, size=3000, dtype=np.uint8)
    assert np.array_equal(o.rank(i, c), r.rank(i, c))
    oc, rc = ol.OCsa(t), ol.RCsa(t)
    assert np.array_equal(oc.bwt(), rc.bwt())
    ser = rc.serialize(0)
    sw, sa = oc.wt().serialize(1), oc.serialize_alphabet()
    assert ser[: len(sw)] == sw and ser[-len(sa):] == sa
    arr = np.frombuffer(t, dtype=np.uint8)
    for m in (3, 20):
        if n < m:
            continue
        st = rng.integers(0, n - m + 1, size=400)
        pats = np.concatenate([arr[s:s + m] for s in st])
        assert np.array_equal(oc.count_batch(pats, m), rc.count_batch(pats, m))


def test_reference_loads_a_serialised_csa():
    """bench.py hands the real library the bytes of an index (there: the one the GPU built) and times its answers;
    here the bytes are the library's own, and the loaded object must answer like the one that wrote them"""
    text = gd.text("faust.txt")
    a = ol.RCsa(text)
    blob = ol._ref_bytes(ol.ref().L.ref_csa_serialize, a.h, 0)
    b = ol.RCsa(sdsl_bytes=blob)
    assert b.size() == a.size() == 226836 and b.count(b"und") == 690
    arr = np.frombuffer(text, dtype=np.uint8)
    rng = np.random.default_rng(2)
    st = rng.integers(0, len(text) - 9, 300)
    pats = np.concatenate([arr[s:s + 9] for s in st])
    assert np.array_equal(a.count_batch(pats, 9), b.count_batch(pats, 9))
    i = rng.integers(0, len(text) + 2, 2000).astype(np.uint64)
    c = arr[rng.integers(0, len(text), 2000)]
    assert np.array_equal(a.wt_rank(i, c), b.wt_rank(i, c))
    with pytest.raises(ValueError):
        ol.RCsa(sdsl_bytes=blob[: len(blob) // 2])
