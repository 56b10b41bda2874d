"""GPU, BASELINE size: the HIP path against answers of the REAL sdsl-lite on the SURVEY.md 8(d) inputs.

tests/golden/golden_large.json was produced in the build container by tests/golden/make_golden_large.py through
oracle/_ref (the reference's own headers): for configs[1] (2^34-bit vector of mt19937_64(42) words), configs[2]
(2^34 bits, 5 % dense, mt19937_64(9)) and configs[3]/[4] (wavelet tree / FM-index of the 2^30-byte English-class text) it
holds, per query stream, sum / xor / sha256 of the reference's answers and the first 10^4 answers.  Here the same inputs are
regenerated from their seeds, the structures are built on the GPU, and the answers must have the same digests — bit-exact
at full size, no property test standing in for the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden_large.json")))


def check(ans, want, what):
    a = np.ascontiguousarray(ans).view(np.uint64)
    assert a.size == want["n"], what
    first = np.array(want["first"], dtype=np.uint64)
    assert np.array_equal(a[: first.size], first), f"{what}: first answers differ from the reference's"
    assert int(np.add.reduce(a, dtype=np.uint64)) == want["sum"], f"{what}: sum of answers"
    assert int(np.bitwise_xor.reduce(a)) == want["xor"], f"{what}: xor of answers"
    assert hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"], f"{what}: sha256 of answers"


@pytest.fixture(scope="module")
def c2_vector(gpu):
    c = G["c2"]
    n = 1 << c["log_n"]
    bv = gpu.bit_vector(gpu.set_random_bits(n, c["words_seed"]), n, device=0)
    assert bv.ones() == c["ones"]
    yield bv, n
    bv.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["direct", "bucketed"])
def test_c2_rank_1_matches_reference_digests(gpu, c2_vector, mode):
    import torch
    bv, n = c2_vector
    c = G["c2"]
    idx = torch.from_numpy(gpu.rnd_positions(c["rank_seed"], c["rank_1"]["n"], n + 1, 0).view(np.int64)).cuda()
    gpu.set_option("rank_sorted", mode)
    try:
        out = bv.rank(idx, 1)
        zeros = bv.rank(idx, 0)
    finally:
        gpu.set_option("rank_sorted", -1)
    check(out.cpu().numpy(), c["rank_1"], f"configs[1] rank_1 ({'bucketed' if mode else 'direct'} path)")
    assert bool((out + zeros == idx).all())


@pytest.mark.parametrize("mode", [0, 1], ids=["direct", "bucketed"])
def test_c2_bit_value_0_matches_reference_digests(gpu, c2_vector, mode):
    """rank_support_v5<0> and select_support_mcl<0> of the real library at 2^34 bits (b = 0 was only compared with the
    direct kernel and through rank_1 + rank_0 == idx)."""
    import torch
    bv, n = c2_vector
    c = G["c2"]
    if "rank_0" not in c:
        pytest.skip("golden_large.json predates the b = 0 digests")
    idx = torch.from_numpy(gpu.rnd_positions(c["rank_seed"], c["rank_0"]["n"], n + 1, 0).view(np.int64)).cuda()
    i0 = torch.from_numpy(gpu.rnd_positions(c["select0_seed"], c["select_0"]["n"], n - c["ones"], 1).view(np.int64)).cuda()
    gpu.set_option("rank_sorted", mode)
    gpu.set_option("select_sorted", mode)
    try:
        r0 = bv.rank(idx, 0)
        s0 = bv.select(i0, 0)
    finally:
        gpu.set_option("rank_sorted", -1)
        gpu.set_option("select_sorted", -1)
    what = "bucketed" if mode else "direct"
    check(r0.cpu().numpy(), c["rank_0"], f"configs[1] rank_0 ({what} path)")
    check(s0.cpu().numpy(), c["select_0"], f"configs[1] select_0 ({what} path)")


@pytest.mark.parametrize("mode", [0, 1], ids=["direct", "bucketed"])
def test_c2_select_1_matches_reference_digests(gpu, c2_vector, mode):
    import torch
    bv, n = c2_vector
    c = G["c2"]
    i = torch.from_numpy(gpu.rnd_positions(c["select_seed"], c["select_1"]["n"], c["ones"], 1).view(np.int64)).cuda()
    gpu.set_option("select_sorted", mode)
    try:
        got = bv.select(i, 1)
    finally:
        gpu.set_option("select_sorted", -1)
    check(got.cpu().numpy(), c["select_1"], f"configs[1] select_1 ({'bucketed' if mode else 'direct'} path)")


def test_c2_default_dispatch_takes_the_bucketed_path_at_bench_size(gpu, c2_vector):
    """With no option set, a batch of >= 4 queries per rank line (select: 2) goes through bv_sorted.hip — for rank AND for select on
    this very vector (its 2^33 + 116138 ones once pushed select over the 2^16-bucket limit and silently back to the
    direct kernel).  The first 10^7 answers are the reference's."""
    import torch
    bv, n = c2_vector
    c = G["c2"]
    nq = 160_000_000
    gpu.set_option("trace_phases", 1)
    try:
        idx = torch.from_numpy(gpu.rnd_positions(c["rank_seed"], nq, n + 1, 0).view(np.int64)).cuda()
        out = bv.rank(idx, 1)
        torch.cuda.synchronize()
        ph = gpu.last_phases()
        assert ph.get("select") == 0 and ph.get("part1", 0) > 0, f"rank took the direct kernel: {ph}"
        check(out[: c["rank_1"]["n"]].cpu().numpy(), c["rank_1"], "configs[1] rank_1 (default dispatch)")
        del idx, out
        i = torch.from_numpy(gpu.rnd_positions(c["select_seed"], nq, c["ones"], 1).view(np.int64)).cuda()
        got = bv.select(i, 1)
        torch.cuda.synchronize()
        ph = gpu.last_phases()
        assert ph.get("select") == 1 and ph.get("part2", 0) > 0, f"select took the direct kernel: {ph}"
        check(got[: c["select_1"]["n"]].cpu().numpy(), c["select_1"], "configs[1] select_1 (default dispatch)")
        back = bv.rank(got[: 1 << 20].clone(), 1)
        assert bool((back == i[: 1 << 20] - 1).all())
    finally:
        gpu.set_option("trace_phases", 0)
        bv.release_scratch()


@pytest.mark.parametrize("shape", ["window", "sorted"])
def test_c2_default_dispatch_keeps_local_batches_on_the_direct_kernel(gpu, c2_vector, shape):
    """A batch confined to a 2^20-bit window, or a sorted one, is served from cache by the direct kernel (9.7 ms against
    21.7 ms per 10^9 queries for the window): the spread sample of the default dispatch sends it there, for rank and for
    select, and the answers are those of the bucketed path."""
    import torch
    bv, n = c2_vector
    nq = 160_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    if shape == "window":
        idx = (n // 3) + torch.randint(0, 1 << 20, (nq,), device="cuda", dtype=torch.int64, generator=g)
    else:
        idx = torch.sort(torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)).values
    ones = G["c2"]["ones"]
    sel = torch.clamp(idx // 2, 1, ones) if shape == "sorted" else (ones // 3) + torch.randint(1, 1 << 19, (nq,), device="cuda", dtype=torch.int64, generator=g)
    try:
        gpu.set_option("trace_phases", 1)
        r_auto = bv.rank(idx, 1)
        s_auto = bv.select(sel, 1)
        torch.cuda.synchronize()
        assert gpu.last_phases() == {}, f"a local batch went through the bucketed passes: {gpu.last_phases()}"
        gpu.set_option("rank_sorted", 1)
        gpu.set_option("select_sorted", 1)
        assert torch.equal(bv.rank(idx, 1), r_auto)
        assert torch.equal(bv.select(sel, 1), s_auto)
        assert gpu.last_phases().get("select") == 1
    finally:
        gpu.set_option("trace_phases", 0)
        gpu.set_option("rank_sorted", -1)
        gpu.set_option("select_sorted", -1)
        bv.release_scratch()


def test_c2_batches_beyond_one_pass(gpu, c2_vector):
    """A batch of more than 2^30 - 2^20 queries goes through the passes in two rounds (32-bit cursors, 30-bit look-back
    counts): rank and select agree with the direct kernels on all 1.1 * 10^9 answers."""
    import torch
    bv, n = c2_vector
    nq = 1_100_000_000
    g = torch.Generator(device="cuda").manual_seed(21)
    idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    try:
        gpu.set_option("rank_sorted", 0)
        want = bv.rank(idx, 1)
        gpu.set_option("rank_sorted", 1)
        got = bv.rank(idx, 1)
        assert torch.equal(got, want)
        del got
        idx.clamp_(1, G["c2"]["ones"])
        gpu.set_option("select_sorted", 0)
        bv.select(idx, 1, out=want)
        gpu.set_option("select_sorted", 1)
        got = bv.select(idx, 1)
        assert torch.equal(got, want)
    finally:
        gpu.set_option("rank_sorted", -1)
        gpu.set_option("select_sorted", -1)
        bv.release_scratch()


def test_c3_rrr63_rank_select_match_reference_digests(gpu):
    import torch
    c = G["c3"]
    n = 1 << c["log_n"]
    ck = np.fromfile(os.path.join(HERE, "golden", "mt9_checkpoints.bin"), dtype=np.uint64).reshape(-1, 313)
    assert ck.shape[0] == c["checkpoints"]
    words = gpu.density_bits(n, c["bits_seed"], c["percent"], ck, c["checkpoint_stride"])
    assert hashlib.sha256(words.tobytes()).hexdigest() == c["words_sha256"], "configs[2] input vector"
    rv = gpu.rrr_vector(words, n, device=0)
    del words
    assert rv.ones() == c["ones"]
    idx = torch.from_numpy(gpu.rnd_positions(c["rank_seed"], c["rank_1"]["n"], n + 1, 0).view(np.int64)).cuda()
    check(rv.rank(idx, 1).cpu().numpy(), c["rank_1"], "configs[2] rrr rank_1")
    i = torch.from_numpy(gpu.rnd_positions(c["select_seed"], c["select_1"]["n"], c["ones"], 1).view(np.int64)).cuda()
    check(rv.select(i, 1).cpu().numpy(), c["select_1"], "configs[2] rrr select_1")
    rv.close()


def test_c4_wt_rank_and_count_match_reference_digests(gpu):
    import torch
    c = G["c4"]
    nt = 1 << c["text_log"]
    text = gpu.english_text(nt, c["text_seed"])
    assert hashlib.sha256(text.tobytes()).hexdigest() == c["text_sha256"], "configs[3] text"
    csa = gpu.csa_wt(text=torch.from_numpy(text).cuda(), device=0)
    assert csa.size() == c["csa_size"] and csa.sigma() == c["sigma"]
    gi = torch.from_numpy(gpu.rnd_positions(c["wt_i_seed"], c["wt_rank"]["n"], nt + 2, 0).view(np.int64)).cuda()
    gc = torch.from_numpy(text[gpu.rnd_positions(c["wt_c_seed"], c["wt_rank"]["n"], nt, 0).astype(np.int64)]).cuda()
    check(csa.wavelet_tree.rank(gi, gc).cpu().numpy(), c["wt_rank"], "configs[3] wt_huff rank(i, c)")
    m = c["m"]
    st = gpu.rnd_positions(c["pattern_seed"], c["count"]["n"], nt - m, 0).astype(np.int64)
    pats = torch.from_numpy(np.ascontiguousarray(text[st[:, None] + np.arange(m)[None, :]].reshape(-1))).cuda()
    check(csa.count(pats, m).cpu().numpy(), c["count"], "configs[4] count of 20-byte patterns")
    if "wt_select" in c:  # wt_huff<>::select of the real library on the same text (make_golden_large.py c4sel)
        ns = c["wt_select"]["n"]
        gcs = torch.from_numpy(text[gpu.rnd_positions(c["wt_c_seed"], ns, nt, 0).astype(np.int64)]).cuda()
        occ = torch.bincount(torch.from_numpy(text).cuda(), minlength=256)
        ks = 1 + torch.from_numpy(gpu.rnd_positions(c["wt_k_seed"], ns, 1 << 62, 0).view(np.int64)).cuda() % occ[gcs.long()]
        csa.close()
        wt_t = gpu.wt_huff(text=torch.from_numpy(text).cuda(), device=0)  # (the digest was made on the text's own tree)
        check(wt_t.select(ks, gcs).cpu().numpy(), c["wt_select"], "configs[3] wt_huff select(k, c)")
        return
    csa.close()


def check_strided(ans_dev, want, what):
    """every `stride`-th answer of the whole BASELINE batch against the real library's (make_golden_large.py c2s / c4s)"""
    assert ans_dev.numel() == want["count"], what
    check(ans_dev[:: want["stride"]].contiguous().cpu().numpy(), want, what)


def test_c2_whole_batch_strided_digests(gpu, c2_vector):
    """configs[1] in full: all 10^9 rank_1 and all 10^9 select_1 queries through the default dispatch (the bucketed passes), every 100th
    answer compared with the real rank_support_v5 / select_support_mcl — the first-10^7 digests above see one slab of the batch only"""
    import torch
    bv, n = c2_vector
    c = G["c2"]
    if "rank_1_strided" not in c:
        pytest.skip("golden_large.json holds no strided digests (make_golden_large.py c2s)")
    w = c["rank_1_strided"]
    idx = gpu.rnd_positions_device(c["rank_seed"], w["count"], n + 1, 0, 0)
    out = torch.empty_like(idx)
    bv.rank(idx, 1, out)
    check_strided(out, w, "configs[1] rank_1, whole batch")
    w = c["select_1_strided"]
    idx = gpu.rnd_positions_device(c["select_seed"], w["count"], c["ones"], 1, 0)
    bv.select(idx, 1, out)
    check_strided(out, w, "configs[1] select_1, whole batch")
    del idx, out
    bv.release_scratch()


def test_c3_whole_batch_strided_digests(gpu):
    """configs[2] in full: 10^9 rank_1 and 10^9 select_1 on the 5 %-dense 2^34-bit rrr_vector<63> through the default dispatch (the
    bucketed decoder of rrr_sorted.hip), every 100th answer against the real library's"""
    import torch
    c = G["c3"]
    if "rank_1_strided" not in c:
        pytest.skip("golden_large.json holds no strided digests for configs[2] (make_golden_large.py c3s)")
    n = 1 << c["log_n"]
    ck = np.fromfile(os.path.join(HERE, "golden", "mt9_checkpoints.bin"), dtype=np.uint64).reshape(-1, 313)
    words = gpu.density_bits(n, c["bits_seed"], c["percent"], ck, c["checkpoint_stride"])
    rv = gpu.rrr_vector(words, n, device=0)
    del words
    w = c["rank_1_strided"]
    idx = gpu.rnd_positions_device(c["rank_seed"], w["count"], n + 1, 0, 0)
    out = torch.empty_like(idx)
    rv.rank(idx, 1, out)
    check_strided(out, w, "configs[2] rrr rank_1, whole batch")
    w = c["select_1_strided"]
    idx = gpu.rnd_positions_device(c["select_seed"], w["count"], c["ones"], 1, 0)
    rv.select(idx, 1, out)
    check_strided(out, w, "configs[2] rrr select_1, whole batch")
    rv.close()


def test_c4_whole_batch_strided_digests_at_both_footprints(gpu):
    """configs[3] / [4] in full: 10^8 rank(i, c) and 10^8 count() of 20-byte patterns, every 100th answer against the real library's
    csa_wt<wt_huff<>> — on the index as created from text AND on the same index reduced to 1.5 x the reference's own bytes
    (sdsl_hip_fm_set_footprint: fused tree lines, 32-bit samples, the k-mer table the budget holds)"""
    import torch
    c = G["c4"]
    if "count_strided" not in c:
        pytest.skip("golden_large.json holds no strided digests (make_golden_large.py c4s)")
    nt = 1 << c["text_log"]
    text = torch.from_numpy(gpu.english_text(nt, c["text_seed"])).cuda()
    csa = gpu.csa_wt(text=text, device=0)
    w = c["wt_rank_strided"]
    gi = gpu.rnd_positions_device(c["wt_i_seed"], w["count"], nt + 2, 0, 0)
    gc = text[gpu.rnd_positions_device(c["wt_c_seed"], w["count"], nt, 0, 0)]
    out = torch.empty(w["count"], dtype=torch.int64, device="cuda")
    csa.wavelet_tree.rank(gi, gc, out)
    check_strided(out, w, "configs[3] wt_huff rank(i, c), whole batch")
    del gi, gc
    w = c["count_strided"]
    m = c["m"]
    st = gpu.rnd_positions_device(c["pattern_seed"], w["count"], nt - m, 0, 0)
    pats = text[(st.view(-1, 1) + torch.arange(m, device="cuda").view(1, m)).reshape(-1)].contiguous()
    del st
    csa.count(pats, m, out)
    check_strided(out, w, "configs[4] count, whole batch")
    blob_bytes = len(csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL))
    csa.set_footprint(int(1.5 * blob_bytes))
    assert csa.device_bytes() <= 1.5 * blob_bytes and csa.footprint_parts()["suffix_array"] == 0
    out.zero_()
    csa.count(pats, m, out)
    check_strided(out, w, "configs[4] count at 1.5 x the reference's footprint, whole batch")
    csa.close()


def test_c4r_the_repetitive_stand_in_matches_reference_digests(gpu):
    """configs[3] / [4] on the REPETITIVE stand-in (english_text_repetitive(2^30, 1234, 30): 30 % of the 64 KiB blocks are rotated copies of
    earlier ones — the duplicated passages of a real collection; a 20-byte pattern cut from the text occurs 2.4 times on average and 46 % of
    the patterns more than once, against 1.017 times on the first stand-in): the first 10^6 and every 100th of 10^8 wt.rank / count answers
    of the real csa_wt<wt_huff<>> (make_golden_large.py c4r) — with suffix array and text resident (the single-suffix shortcut fires late
    or never for half of the patterns), with both dropped, and at 1.5 x the reference's bytes."""
    import torch
    c = G.get("c4r")
    if not c:
        pytest.skip("golden_large.json holds no c4r block (make_golden_large.py c4r)")
    nt = 1 << c["text_log"]
    host = gpu.english_text_repetitive(nt, c["text_seed"], c["copy_percent"])
    assert hashlib.sha256(host.tobytes()).hexdigest() == c["text_sha256"]
    text = torch.from_numpy(host).cuda()
    del host
    csa = gpu.csa_wt(text=text, device=0)
    assert csa.size() == c["csa_size"] and csa.sigma() == c["sigma"]
    w = c["wt_rank_strided"]
    gi = gpu.rnd_positions_device(c["wt_i_seed"], w["count"], nt + 2, 0, 0)
    gc = text[gpu.rnd_positions_device(c["wt_c_seed"], w["count"], nt, 0, 0)]
    out = torch.empty(w["count"], dtype=torch.int64, device="cuda")
    csa.wavelet_tree.rank(gi, gc, out)
    check(out[: c["wt_rank"]["n"]].cpu().numpy(), c["wt_rank"], "wt_huff rank(i, c) on the repetitive text, first answers")
    check_strided(out, w, "wt_huff rank(i, c) on the repetitive text, whole batch")
    del gi, gc
    w = c["count_strided"]
    m = c["m"]
    st = gpu.rnd_positions_device(c["pattern_seed"], w["count"], nt - m, 0, 0)
    pats = text[(st.view(-1, 1) + torch.arange(m, device="cuda").view(1, m)).reshape(-1)].contiguous()
    del st
    for stage in ("suffix array and text resident", "samples only", "1.5 x the reference's bytes"):
        if stage == "samples only":
            blob_bytes = len(csa.serialize(32, 64, gpu.capi.LAYOUT_BV_MCL))
            csa.drop_sa()
        elif stage.startswith("1.5"):
            csa.set_footprint(int(1.5 * blob_bytes))
            assert csa.device_bytes() <= 1.5 * blob_bytes
        out.zero_()
        csa.count(pats, m, out)
        check(out[: c["count"]["n"]].cpu().numpy(), c["count"], f"count on the repetitive text ({stage}), first answers")
        check_strided(out, w, f"count on the repetitive text ({stage}), whole batch")
    mean = float(out[: c["count"]["n"]].double().mean())
    assert abs(mean - c["mean_count"]) < 1e-9 and mean > 2.0
    csa.close()
