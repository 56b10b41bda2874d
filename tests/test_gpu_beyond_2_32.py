"""More than 2^32 symbols.  The reference's structures are width-agnostic (`int_vector<0>`, 64-bit positions: wt_pc.hpp:366-474,
csa_wt.hpp:107-110, suffix_array_algorithm.hpp:167-248); 288 GB of HBM hold such inputs, so what can be built for them has to be
right on them:

* `wt_huff` over 2^32 + 10^6 symbols — rank, select and access against prefix counts computed on the device;
* `csa_wt` over a 6.3-Gsymbol text, created from its BWT — `count`, the SA intervals, single backward-search steps and LF
  against the CLOSED FORM of a periodic text (tests/periodic_text.py, checked against the oracle at small sizes on the CPU).

* `csa_wt` from a TEXT of 4.3 G symbols (64-bit suffix sorter, sa.hip): csa[i], isa[i], count, locate, extract against the same
  closed form; the sorter on small texts against the oracle.

Sequences of 2^32 .. 2^36 symbols walk the fused lines: 16-ary with 64-bit superblock counts and a select directory of 64-bit
entries (rank / access / LF / count / select; wt_device.hpp), or — a SDSL_HIP_FUSED_K=3 build — 8-ary with the list of places
where a count passes a multiple of 2^32 (select on the binary levels there); count takes the WIDE flat kernels and the k-mer
table with 40-bit intervals (fm_count2.hip)."""
import numpy as np
import pytest

import oracle_lib as ol
import periodic_text as pt

pytestmark = pytest.mark.gpu


def test_wt_huff_over_more_than_2_pow_32_symbols(gpu):
    import torch
    n, sigma = (1 << 32) + 1_000_003, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for a in range(0, n, step):                        # a skewed alphabet: symbol = 1 + floor(sigma * u^2)
        b = min(n, a + step)
        u = torch.rand(b - a, device="cuda", generator=g)
        text[a:b] = (1 + (u * u * sigma).to(torch.int64).clamp_(max=sigma - 1)).to(torch.uint8)
    wt = gpu.wt_huff(text=text)
    assert wt.size() == n and wt.sigma() == sigma
    nq = 1_000_000
    i = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    i[:1000] = n - torch.arange(1000, device="cuda")
    i[1000:2000] = (1 << 32) - 500 + torch.arange(1000, device="cuda")
    for c in (1, 2, sigma // 2, sigma, sigma + 3):
        cs = torch.cumsum((text == c).to(torch.int64), 0)
        want = torch.where(i > 0, cs[(i - 1).clamp_(min=0)], torch.zeros_like(i))
        cc = torch.full((nq,), c, dtype=torch.uint8, device="cuda")
        assert torch.equal(wt.rank(i, cc).to(torch.int64), want), f"rank(i, {c})"
        total = int(cs[-1])
        if total:
            k = torch.randint(1, total + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
            k[:1000] = total - torch.arange(1000, device="cuda").clamp_(max=total - 1)
            assert torch.equal(wt.select(k, cc).to(torch.int64), torch.searchsorted(cs, k)), f"select(k, {c})"
        del cs, want
    cq = torch.randint(1, sigma + 1, (nq,), device="cuda", dtype=torch.int64, generator=g).to(torch.uint8)
    j = torch.randint(0, n, (nq,), device="cuda", dtype=torch.int64, generator=g)
    j[:1000] = n - 1 - torch.arange(1000, device="cuda")
    assert torch.equal(wt.access(j).to(torch.uint8), text[j])
    # mixed symbols in one batch: rank(j, text[j]) + 1 == inverse_select's rank, and select undoes it
    r = wt.rank(j, text[j]).to(torch.int64)
    assert torch.equal(wt.select(r + 1, text[j]).to(torch.int64), j)
    del cq
    wt.close()


def test_csa_wt_over_more_than_2_pow_32_symbols_from_its_bwt(gpu):
    import torch
    p, k, sigma = 1 << 20, 6000, 20                                  # 6.3 Gsymbols: a third of the suffixes lie beyond 2^32
    n = p * k
    assert n > (1 << 32)
    u = pt.unit(p, sigma, 9)
    sau = pt.unit_suffix_array(u, ol.OCsa)
    inv = np.empty(p, dtype=np.int64)
    inv[sau] = np.arange(p)
    bwt = pt.bwt_device(u, sau, k)
    csa = gpu.csa_wt(bwt=bwt)
    del bwt
    assert csa.size() == n + 1 and csa.sigma() == sigma + 2          # V's bytes, '#', the sentinel
    rng = np.random.default_rng(4)
    for m in (1, 6, 20, 33):
        npat = 6000
        pos = rng.integers(0, n - m, npat)
        pos[:200] = n - m - np.arange(200)                            # the text's end: the last copy's suffixes
        pats = pt.text_at(u, pos, m).copy()
        mut = rng.random(npat) < 0.3
        pats[mut, rng.integers(0, m, int(mut.sum()))] = rng.integers(1, sigma + 3, int(mut.sum()), dtype=np.uint8)
        flat = np.ascontiguousarray(pats.reshape(-1))
        want = pt.count_in_text(u, k, pats)
        got = np.asarray(csa.count(flat, m)).astype(np.uint64)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"m {m}: pattern {bytes(pats[bad[0]])!r}: got {got[bad[0]]}, want {want[bad[0]]}"
        l, r = (np.asarray(a).astype(np.uint64) for a in csa.interval(flat, m))
        occ = want > 0
        assert np.array_equal((r + np.uint64(1) - l)[occ], want[occ])
        assert (r[occ] > np.uint64(1 << 32)).any() and (l[occ] < np.uint64(1 << 32)).any()
        # the interval itself: the suffixes that start with an occurring pattern are those of the offsets whose suffix of U U
        # starts with it — consecutive ranks [ra, rb] — all k copies of each; a pattern that does not fit into the last copy
        # contains '#', matches at ONE offset, and that offset loses its first (smallest) suffix
        if m == 20:
            uu = np.concatenate([u, u])
            for q in np.flatnonzero(occ)[:300]:
                i0 = int(pos[q] % p) if not mut[q] else None
                if i0 is None:
                    continue
                ra = rb = int(inv[i0])
                while ra > 0 and np.array_equal(uu[sau[ra - 1]:sau[ra - 1] + m], pats[q]):
                    ra -= 1
                while rb + 1 < p and np.array_equal(uu[sau[rb + 1]:sau[rb + 1] + m], pats[q]):
                    rb += 1
                first = 1 + ra * k + (1 if sau[ra] + m > p else 0)
                assert int(l[q]) == first, (q, int(l[q]), first)
    # LF at places on both sides of 2^32: the suffix in front of (offset i, copy j) is (i - 1, j); in front of offset 0 stands
    # the '#' of the copy before
    x = np.concatenate([rng.integers(1, n + 1, 200_000), np.array([1, n, (1 << 32) - 1, 1 << 32, (1 << 32) + 1])]).astype(np.int64)
    rk, t = (x - 1) // k, (x - 1) % k
    i = sau[rk]
    want_lf = np.where(i > 0, 1 + inv[np.maximum(i - 1, 0)] * k + t, 1 + inv[p - 1] * k + t + 1)
    whole = (i == 0) & (t == k - 1)                                   # the suffix that is all of T: LF leads to the sentinel's row
    want_lf[whole] = 0
    got_lf = np.asarray(csa.lf(x.astype(np.uint64))).astype(np.int64)
    assert np.array_equal(got_lf, want_lf)
    # single backward-search steps on intervals beyond 2^32 (suffix_array_algorithm.hpp:167-200): extend every pattern's interval
    # by the character in front of one of its occurrences and compare with the search for the longer pattern
    m = 12
    pos = rng.integers(1, n - m, 4000)
    pats = pt.text_at(u, pos, m)
    longer = pt.text_at(u, pos - 1, m + 1)
    l, r = csa.interval(np.ascontiguousarray(pats.reshape(-1)), m)
    l2, r2 = csa.backward_search(l, r, np.ascontiguousarray(longer[:, 0]))
    lw, rw = csa.interval(np.ascontiguousarray(longer.reshape(-1)), m + 1)
    assert np.array_equal(np.asarray(l2), np.asarray(lw)) and np.array_equal(np.asarray(r2), np.asarray(rw))
    assert (np.asarray(lw).astype(np.uint64) > np.uint64(1 << 32)).any()
    csa.close()


def test_the_64_bit_suffix_sorter_on_small_texts(gpu, monkeypatch):
    """SDSL_HIP_SA64=1 sends a text of any size through the sorter for 2^32 symbols and more (sa.hip: two stable sorts per
    doubling round, 64-bit suffixes) and the index keeps SA / ISA samples instead of the whole array: suffix array, inverse,
    count, locate and extract against the oracle, on texts with long repeats (many rounds) and on a periodic one."""
    rng = np.random.default_rng(8)
    texts = [rng.integers(1, 5, 70_001, dtype=np.uint8), np.tile(pt.unit(257, 3, 1), 41), np.full(5000, 7, dtype=np.uint8),
             np.frombuffer(b"abracadabra", dtype=np.uint8), np.array([9], dtype=np.uint8)]
    texts[0][30_000:34_000] = texts[0][1000:5000]
    for text in texts:
        n = text.size
        ocsa = ol.OCsa(bytes(text))
        monkeypatch.delenv("SDSL_HIP_SA64", raising=False)
        ref = gpu.csa_wt(text=text)
        monkeypatch.setenv("SDSL_HIP_SA64", "1")
        csa = gpu.csa_wt(text=text)
        assert csa.sampling() == (32, 64, True) and ref.sampling() == (0, 0, True)  # samples AND the 64-bit suffix array
        assert csa.serialize(32, 64) == ref.serialize(32, 64), "the stream of csa_wt<wt_huff<>, 32, 64> from the samples the index keeps"
        with pytest.raises(Exception):
            csa.serialize(16, 64)
        ref.close()
        idx = np.arange(n + 1, dtype=np.uint64) if n < 20_000 else rng.integers(0, n + 1, 20_000).astype(np.uint64)
        assert np.array_equal(np.asarray(csa.sa(idx)), np.asarray(ocsa.sa(idx)))
        assert np.array_equal(np.asarray(csa.isa(idx)), np.asarray(ocsa.isa(idx)))
        m = min(6, n)
        st = rng.integers(0, n - m + 1, 3000)
        pats = text[st[:, None] + np.arange(m)[None, :]].copy()
        pats[::4, 0] = 3
        want = np.array([ocsa.count(bytes(r)) for r in pats], dtype=np.uint64)
        assert np.array_equal(np.asarray(csa.count(np.ascontiguousarray(pats.reshape(-1)), m)).astype(np.uint64), want)
        for r in pats[:20]:
            got = np.sort(np.asarray(csa.locate(np.ascontiguousarray(r), m)[1]))
            assert np.array_equal(got, np.sort(ocsa.locate(bytes(r)))), bytes(r)
        assert bytes(np.asarray(csa.extract(np.array([0], dtype=np.uint64), np.array([n - 1], dtype=np.uint64))[1])) == bytes(text)
        csa.drop_sa()                                   # from the samples alone: the walks, and count() without the text
        assert csa.sampling() == (32, 64, False)
        assert np.array_equal(np.asarray(csa.sa(idx)), np.asarray(ocsa.sa(idx)))
        assert np.array_equal(np.asarray(csa.count(np.ascontiguousarray(pats.reshape(-1)), m)).astype(np.uint64), want)
        csa.close()


def test_csa_wt_from_a_text_of_more_than_2_pow_32_symbols(gpu):
    """construct(csa, text) on 4.3 G symbols: the suffix sorter's array against the closed form (through the index's SA / ISA
    samples: csa[i], isa[i]), the BWT through count, locate and extract."""
    import torch
    p, k, sigma = 1 << 20, 4100, 20
    n = p * k
    assert n > (1 << 32)
    u = pt.unit(p, sigma, 9)
    sau = pt.unit_suffix_array(u, ol.OCsa)
    inv = np.empty(p, dtype=np.int64)
    inv[sau] = np.arange(p)
    text = torch.from_numpy(u).cuda().repeat(k)
    csa = gpu.csa_wt(text=text)
    assert csa.size() == n + 1 and csa.sigma() == sigma + 2 and csa.sampling() == (32, 64, True)
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.integers(1, n + 1, 100_000), np.array([0, 1, n, (1 << 32) - 1, 1 << 32, (1 << 32) + 1])]).astype(np.int64)
    rk, t = (x - 1) // k, (x - 1) % k
    want_sa = np.where(x > 0, sau[np.maximum(rk, 0)] + (k - 1 - t) * p, n)
    assert np.array_equal(np.asarray(csa.sa(x.astype(np.uint64))).astype(np.int64), want_sa)
    pos = np.concatenate([rng.integers(0, n, 100_000), np.array([0, n - 1, n, (1 << 32) - 1, 1 << 32])]).astype(np.int64)
    want_isa = np.where(pos < n, 1 + inv[pos % p] * k + (k - 1 - pos // p), 0)
    assert np.array_equal(np.asarray(csa.isa(pos.astype(np.uint64))).astype(np.int64), want_isa)
    m = 24
    st = rng.integers(0, n - m, 3000)
    pats = pt.text_at(u, st, m).copy()
    pats[::5, 3] = 1 + rng.integers(0, sigma + 1)
    want = pt.count_in_text(u, k, pats)
    assert np.array_equal(np.asarray(csa.count(np.ascontiguousarray(pats.reshape(-1)), m)).astype(np.uint64), want)
    q = int(np.flatnonzero(want > 0)[0])
    off, where = csa.locate(np.ascontiguousarray(pats[q]), m)
    where = np.sort(np.asarray(where).astype(np.int64))
    assert where.size == int(want[q]) and np.unique(where).size == where.size
    assert all(np.array_equal(pt.text_at(u, where[:50], m)[i], pats[q]) for i in range(min(50, where.size)))
    assert where[-1] > (1 << 32)
    b = np.array([0, (1 << 32) - 10, n - 40], dtype=np.uint64)
    e = b + np.uint64(39)
    offs, got = csa.extract(b, e)
    got = np.asarray(got)
    for i in range(3):
        assert np.array_equal(got[i * 40:(i + 1) * 40], pt.text_at(u, np.array([int(b[i])]), 40)[0])
    # the same from the samples alone (no 64-bit suffix array, no text: every character an LF step, csa[i] a walk)
    csa.drop_sa()
    assert csa.sampling() == (32, 64, False)
    assert np.array_equal(np.asarray(csa.count(np.ascontiguousarray(pats.reshape(-1)), m)).astype(np.uint64), want)
    assert np.array_equal(np.asarray(csa.sa(x[:20_000].astype(np.uint64))).astype(np.int64), want_sa[:20_000])
    assert np.array_equal(np.asarray(csa.isa(pos[:20_000].astype(np.uint64))).astype(np.int64), want_isa[:20_000])
    csa.close()


def test_an_index_of_more_than_2_pow_32_symbols_against_the_real_library(gpu):
    """The regime pinned to sdsl-lite itself (only a closed form stood behind it).  A csa_wt over a random skewed text
    of 2^32 + 777 symbols is built on the GPU (64-bit suffix sorter), serialised as csa_wt<wt_huff<bit_vector, rank_support_v5<>>, 32, 64>
    and LOADED BY THE REAL LIBRARY on the host (oracle/_ref; divsufsort of 4 GiB does not fit a test, loading 4 GB does); both then
    answer the same count / csa[i] / isa[i] / wavelet_tree.rank / extract queries, with arguments on both sides of 2^32."""
    import torch
    if not ol.have_ref():
        pytest.skip("oracle/_ref (the real sdsl-lite, built in the container that holds /root/reference) did not travel with the repo")
    n, sigma = (1 << 32) + 777, 40
    g = torch.Generator(device="cuda").manual_seed(1)
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        u = torch.rand(b - a, device="cuda", generator=g)
        text[a:b] = (1 + (u * u * sigma).to(torch.int64).clamp_(max=sigma - 1)).to(torch.uint8)
    csa = gpu.csa_wt(text=text)
    assert csa.size() == n + 1 and csa.sampling()[:2] == (32, 64)
    blob = csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL)
    ref = ol.RCsa(sdsl_bytes=blob)
    del blob
    assert ref.size() == n + 1 and ref.sigma() == csa.sigma()
    rng = np.random.default_rng(12)
    nq, m = 100_000, 20
    st = np.concatenate([rng.integers(0, n - m, nq - 4), np.array([0, (1 << 32) - 10, (1 << 32) - 1, n - m])]).astype(np.int64)
    th = torch.from_numpy(st).cuda()
    pats = text[(th.view(-1, 1) + torch.arange(m, device="cuda").view(1, m)).reshape(-1)].contiguous()
    mut = torch.from_numpy(rng.random(nq) < 0.25).cuda()
    pv = pats.view(-1, m)
    pv[mut, 7] = (pv[mut, 7] % sigma) + 1                          # a quarter of the patterns changed in one place
    flat = pats.cpu().numpy()
    want = ref.count_batch(flat, m)
    full_bytes = csa.device_bytes()
    for stage in ("suffix array and text resident", "samples only", "footprint: fused lines + samples + a k-mer table within 8 GB",
                  "suffix array and text restored"):
        if stage.startswith("footprint"):
            # set_footprint on an index of 2^32 symbols and more: the 64-bit suffix array and the text come back for the time of the call
            # (the k-mer table is built from them), the binary levels go, the samples stay 64 bits wide
            csa.set_footprint(8 << 30)
            parts = csa.footprint_parts()
            assert csa.device_bytes() <= (8 << 30) and parts["wt_binary_levels"] == 0 and parts["suffix_array"] == 0 and parts["text"] == 0
            assert parts["sa_isa_samples"] == 8 * ((n + 32) // 32 + (n + 64) // 64) and csa.kmer_table_depth() >= 1
        elif stage.endswith("restored"):
            csa.restore_suffix_array()
            assert csa.footprint_parts()["suffix_array"] == 8 * (n + 1) and csa.sampling() == (32, 64, True)
        got = np.asarray(csa.count(flat, m)).astype(np.uint64)
        assert np.array_equal(got, want), f"count ({stage}): first difference at {np.flatnonzero(got != want)[:3]}"
        short = np.ascontiguousarray(flat.reshape(-1, m)[:20_000, m - 5:]).reshape(-1)   # 5-byte patterns: wide intervals across 2^32
        l, r = csa.interval(short, 5)
        lw, rw = ref.interval_batch(short, 5)
        assert np.array_equal(np.asarray(l), lw) and np.array_equal(np.asarray(r), rw), stage
        assert (rw > np.uint64(1 << 31)).any() and (rw + np.uint64(1) - lw > np.uint64(1000)).any()  # wide intervals, both halves of the array
        idx = np.concatenate([rng.integers(0, n + 1, 20_000), np.array([0, 1, n, (1 << 32) - 1, 1 << 32, (1 << 32) + 1])]).astype(np.uint64)
        assert np.array_equal(np.asarray(csa.sa(idx)), ref.sa(idx)), f"csa[i] ({stage})"
        assert np.array_equal(np.asarray(csa.isa(idx)), ref.isa(idx)), f"isa[i] ({stage})"
        qi = np.concatenate([rng.integers(0, n + 2, 100_000), np.array([0, n + 1, 1 << 32, (1 << 32) + 1])]).astype(np.uint64)
        qc = rng.integers(1, sigma + 3, qi.size).astype(np.uint8)
        assert np.array_equal(np.asarray(csa.wavelet_tree.rank(qi, qc)), ref.wt_rank(qi, qc)), f"wavelet_tree.rank ({stage})"
        b = np.array([0, (1 << 32) - 30, n - 64], dtype=np.uint64)
        off, got_t = csa.extract(b, b + np.uint64(63))
        got_t = np.asarray(got_t)
        for i in range(3):
            assert bytes(got_t[i * 64:(i + 1) * 64]) == ref.extract(int(b[i]), int(b[i]) + 63), f"extract ({stage})"
        if stage.startswith("suffix array and text resident"):
            csa.drop_sa()
    assert csa.device_bytes() < full_bytes  # (the binary levels stay released)
    csa.close()


def test_wt_huff_over_more_than_2_pow_35_symbols(gpu):
    """Positions from 2^35 on inside ONE node of the fused 16-ary layout: (i >> 3) no longer fits 32 bits there, the form of the
    line arithmetic that round 5 shipped (wt_device.hpp: fused_line; tests/test_fused_addressing.py is its host-side check).  The
    sequence is periodic — a unit of P symbols, P prime — so rank / select / access have closed forms:
        rank(i, c) = (i div P) * occ_c + prefix_c[i mod P],    select(k, c) = ((k - 1) div occ_c) * P + where_c[(k - 1) mod occ_c].
    The reference is 64-bit throughout (wt_pc.hpp:371-399, 401-474)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()                               # (what earlier tests of the run left in torch's caching allocator)
    free, total = torch.cuda.mem_get_info()
    if free < (195 << 30):                                 # text 34.4 GB + level keys 2 x 68.7 GB + the tree's bits 17.8 GB = 189.6 GB at the peak
        pytest.skip(f"needs 195 GB of free HBM (text 34 GB + the level sorter's 137 GB + 18 GB of tree bits), {free >> 30} GB free")
    P, sigma = 1_000_003, 24
    n = (1 << 35) + 5 * P + 17
    rng = np.random.default_rng(35)
    uh = (1 + np.minimum((rng.random(P) ** 2 * sigma).astype(np.int64), sigma - 1)).astype(np.uint8)   # skewed: codes of several lengths
    unit = torch.from_numpy(uh).cuda()
    reps = (n + P - 1) // P
    text = unit.repeat(reps)[:n].contiguous()
    assert text.numel() == n
    wt = gpu.wt_huff(text=text)
    del text
    torch.cuda.empty_cache()
    assert wt.size() == n and wt.sigma() == sigma
    assert wt.fused_steps().any(), "the fused layout was not built: this test is about its lines"
    nq = 2_000_000
    g = torch.Generator(device="cuda").manual_seed(7)
    i = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    i[:1000] = n - torch.arange(1000, device="cuda")
    i[1000:3000] = (1 << 35) - 1000 + torch.arange(2000, device="cuda")                    # both sides of 2^35
    i[3000:5000] = (1 << 32) - 1000 + torch.arange(2000, device="cuda")
    i[5000:nq // 2] = torch.randint(1 << 35, n + 1, (nq // 2 - 5000,), device="cuda", dtype=torch.int64, generator=g)
    full, part = i // P, i % P
    occ_all = np.bincount(uh, minlength=256)
    for c in (1, 2, sigma // 2, sigma, sigma + 3):
        pre = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum((unit == c).to(torch.int64), 0)])
        want = full * int(occ_all[c]) + pre[part]
        cc = torch.full((nq,), c, dtype=torch.uint8, device="cuda")
        got = wt.rank(i, cc).to(torch.int64)
        assert torch.equal(got, want), f"rank(i, {c}): first difference at i = {int(i[(got != want).nonzero()[0, 0]])}"
        tot = int(want.new_tensor(n // P * int(occ_all[c])) + pre[n % P])
        if tot:
            where = torch.nonzero(unit == c).view(-1)
            k = torch.randint(1, tot + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
            k[:1000] = tot - torch.arange(1000, device="cuda").clamp_(max=tot - 1)
            want_s = ((k - 1) // int(occ_all[c])) * P + where[(k - 1) % int(occ_all[c])]
            got_s = wt.select(k, cc).to(torch.int64)
            assert torch.equal(got_s, want_s), f"select(k, {c})"
            assert int(want_s.max()) >= (1 << 35)
    j = i.clamp(max=n - 1)
    sym = unit[j % P]
    assert torch.equal(wt.access(j).to(torch.uint8), sym)
    r = wt.rank(j, sym).to(torch.int64)                      # mixed symbols in one batch; select undoes inverse_select
    assert torch.equal(wt.select(r + 1, sym).to(torch.int64), j)
    wt.close()
