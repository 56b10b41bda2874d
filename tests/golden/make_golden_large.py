"""BASELINE-size answers of the REAL sdsl-lite (run in the build container only; ~62 GB of RAM, 30-60 minutes).

    python tests/golden/make_golden_large.py [c2] [c2w] [c3] [c4] [c4sel] [c2s] [c4s] [c4r] [--text-file FILE]

Builds, through oracle/_ref/libsdsl_ref.so (the reference's headers compiled where they lie), the structures of
BASELINE.json configs[1..4] on the SURVEY.md 8(d) inputs and stores what the GPU tests and bench.py compare against:
per query stream the sum, the xor and the sha256 of the answers plus the first 10^4 answers themselves
(the reference's own harness prints such a checksum: benchmark/rrr_vector/src/rrr_time_and_space.cpp:86-108).

  c2  n = 2^34, words = std::mt19937_64(42) (util::set_random_bits, util.hpp:467-485); rank_support_v5<1>,
      select_support_mcl<1>; 10^7 rank_1 at mt19937_64(7) % (n+1), 10^7 select_1 at 1 + mt19937_64(11) % ones
  c3  n = 2^34, bit i = (mt19937_64(9)_i % 100 < 5) drawn sequentially; rrr_vector<63> rank_1 / select_1, same streams
      (+ the generator checkpoints tests/golden/mt9_checkpoints.bin, validated here by comparing the whole vector
      produced from them with std::mt19937_64's sequential one)
  c4  text = the library's English-class stand-in (sdsl_hip_util_english_text, seed 1234), 2^30 bytes;
      csa_wt<wt_huff<bit_vector, rank_support_v5<>>> built by sdsl::construct_im; 10^6 wavelet_tree.rank(i, c) with
      i = mt19937_64(13) % (n+1), c = text[mt19937_64(14) % n]; 10^6 count() of the 20 bytes at mt19937_64(15) % (n-20)

  c2s / c4s  STRIDED digests over the WHOLE BASELINE batch (every 100th answer of the 10^9 rank_1 / select_1 queries of c2, of the
      10^8 rank(i, c) / count() queries of c4) instead of its first 10^7: a wrong answer anywhere in the batch has a 1 % chance per
      answer of being seen, a systematic one (a slab, a bucket, a chunk boundary) is seen for sure
  --text-file FILE   c4 / c4sel / c4s on a text of your own (e.g. Pizza&Chili english.1GB: first 2^TEXT_LOG bytes, zero bytes dropped)
      — the digests go to golden_large_<basename>.json beside this script; bench.py --text-file FILE picks that file up

Everything written is DATA (seeded inputs are regenerated, never stored): tests/golden/golden_large.json,
tests/golden/mt9_checkpoints.bin.
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

OUT = os.path.join(HERE, "golden_large.json")
CKPT = os.path.join(HERE, "mt9_checkpoints.bin")
LOG_N = int(os.environ.get("GOLDEN_LOG_N", "34"))
NQ = int(os.environ.get("GOLDEN_NQ", str(10**7)))
TEXT_LOG = int(os.environ.get("GOLDEN_TEXT_LOG", "30"))
NQ_TEXT = int(os.environ.get("GOLDEN_NQ_TEXT", str(10**6)))
CKPT_STRIDE = 1 << 28


def digest(ans: np.ndarray) -> dict:
    a = np.ascontiguousarray(ans, dtype=np.uint64)
    return {"n": int(a.size), "sum": int(np.add.reduce(a, dtype=np.uint64)), "xor": int(np.bitwise_xor.reduce(a)),
            "sha256": hashlib.sha256(a.tobytes()).hexdigest(), "first": [int(x) for x in a[:10_000]]}


STRIDE = int(os.environ.get("GOLDEN_STRIDE", "100"))
NQ_FULL = int(os.environ.get("GOLDEN_NQ_FULL", str(10**9)))            # BASELINE configs[1]: 10^9 queries
NQ_TEXT_FULL = int(os.environ.get("GOLDEN_NQ_TEXT_FULL", str(10**8)))  # configs[3], [4]: 10^8 queries / patterns


def strided(ans: np.ndarray, count: int) -> dict:
    d = digest(ans)
    d.update(stride=STRIDE, count=count)  # answers 0, stride, 2 * stride, ... of the first `count`
    return d


def load_text(pkg, text_file):
    nt = 1 << TEXT_LOG
    if text_file:
        raw = np.fromfile(text_file, dtype=np.uint8, count=nt)
        return np.ascontiguousarray(raw[raw != 0])
    return pkg.english_text(nt, 1234)


def main():
    global OUT
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    pkg = importlib.import_module("sdsl-lite_amd")
    argv = sys.argv[1:]
    text_file = None
    if "--text-file" in argv:
        i = argv.index("--text-file")
        text_file = argv[i + 1]
        del argv[i:i + 2]
        OUT = os.path.join(HERE, "golden_large_%s.json" % os.path.basename(text_file))
    want = set(argv) or {"c2", "c3", "c4"}
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    R = ol.ref().L
    n = 1 << LOG_N

    if "c2w" in want:
        # rank_support_v5<1> / <0> of the real library on a 2^36-bit vector (words = mt19937_64(42) as for c2): beyond the 2^26
        # rank lines the bucketed path handled until round 4 (a slice is 2^12 lines here, answered in four rounds)
        t0 = time.time()
        nw = 1 << 36
        words = np.zeros(nw // 64 + 2, dtype=np.uint64)
        R.ref_set_random_bits(words.ctypes.data, nw, 42)
        h = R.ref_bv_create_rank(words.ctypes.data, nw)
        del words
        idx = pkg.rnd_positions(7, NQ, nw + 1, 0)
        out = np.empty(NQ, dtype=np.uint64)
        R.ref_bv_rank(h, 1, idx.ctypes.data, NQ, out.ctypes.data)
        c2w = {"log_n": 36, "words_seed": 42, "rank_seed": 7, "rank_1": digest(out)}
        R.ref_bv_rank(h, 0, idx.ctypes.data, NQ, out.ctypes.data)
        c2w.update(rank_0=digest(out))
        tot = np.empty(1, dtype=np.uint64)
        R.ref_bv_rank(h, 1, np.array([nw], dtype=np.uint64).ctypes.data, 1, tot.ctypes.data)
        c2w["ones"] = int(tot[0])
        R.ref_bv_destroy(h)
        res["c2w"] = c2w
        json.dump(res, open(OUT, "w"))
        print(f"c2w done in {time.time() - t0:.0f}s: ones={c2w['ones']}", flush=True)

    if "c2" in want:
        t0 = time.time()
        words = np.zeros(n // 64 + 2, dtype=np.uint64)
        R.ref_set_random_bits(words.ctypes.data, n, 42)  # the real util::set_random_bits
        assert np.array_equal(words[:1024], pkg.set_random_bits(1 << 16, 42)), "product generator != util::set_random_bits"
        h = R.ref_bv_create(words.ctypes.data, n)
        idx = pkg.rnd_positions(7, NQ, n + 1, 0)
        assert np.array_equal(idx[:100000], ol.mt19937_64(100000, 7) % np.uint64(n + 1))
        out = np.empty(NQ, dtype=np.uint64)
        R.ref_bv_rank(h, 1, idx.ctypes.data, NQ, out.ctypes.data)
        one = np.uint64(n)
        tot = np.empty(1, dtype=np.uint64)
        R.ref_bv_rank(h, 1, np.array([n], dtype=np.uint64).ctypes.data, 1, tot.ctypes.data)
        ones = int(tot[0])
        c2 = {"log_n": LOG_N, "words_seed": 42, "ones": ones, "rank_seed": 7, "rank_1": digest(out)}
        si = pkg.rnd_positions(11, NQ, ones, 1)
        R.ref_bv_select(h, 1, si.ctypes.data, NQ, out.ctypes.data)
        c2.update(select_seed=11, select_1=digest(out))
        # the other bit value (rank_support_v5<0>, select_support_mcl<0>): the same positions; arguments 1 + mt19937_64(12) % zeros
        R.ref_bv_rank(h, 0, idx.ctypes.data, NQ, out.ctypes.data)
        c2.update(rank_0=digest(out))
        s0 = pkg.rnd_positions(12, NQ, n - ones, 1)
        R.ref_bv_select(h, 0, s0.ctypes.data, NQ, out.ctypes.data)
        c2.update(select0_seed=12, select_0=digest(out))
        del s0
        R.ref_bv_destroy(h)
        res["c2"] = c2
        del words, idx, out, si
        print(f"c2 done in {time.time() - t0:.0f}s: ones={ones}", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c2s" in want:
        t0 = time.time()
        words = np.zeros(n // 64 + 2, dtype=np.uint64)
        R.ref_set_random_bits(words.ctypes.data, n, 42)
        h = R.ref_bv_create(words.ctypes.data, n)
        ns = (NQ_FULL + STRIDE - 1) // STRIDE
        idx = np.ascontiguousarray(pkg.rnd_positions(7, NQ_FULL, n + 1, 0)[::STRIDE])
        out = np.empty(ns, dtype=np.uint64)
        R.ref_bv_rank(h, 1, idx.ctypes.data, ns, out.ctypes.data)
        res["c2"]["rank_1_strided"] = strided(out, NQ_FULL)
        del idx
        si = np.ascontiguousarray(pkg.rnd_positions(11, NQ_FULL, res["c2"]["ones"], 1)[::STRIDE])
        R.ref_bv_select(h, 1, si.ctypes.data, ns, out.ctypes.data)
        res["c2"]["select_1_strided"] = strided(out, NQ_FULL)
        R.ref_bv_destroy(h)
        del words, si, out
        print(f"c2s done in {time.time() - t0:.0f}s", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c4s" in want:
        t0 = time.time()
        text = load_text(pkg, text_file)
        nt = int(text.size)
        if not text_file:
            assert hashlib.sha256(text.tobytes()).hexdigest() == res["c4"]["text_sha256"]
        csa = ol.RCsa(text=text.tobytes())
        print(f"c4s csa built {time.time() - t0:.0f}s", flush=True)
        ns = (NQ_TEXT_FULL + STRIDE - 1) // STRIDE
        gi = np.ascontiguousarray(pkg.rnd_positions(13, NQ_TEXT_FULL, nt + 2, 0)[::STRIDE])
        gc = np.ascontiguousarray(text[pkg.rnd_positions(14, NQ_TEXT_FULL, nt, 0)[::STRIDE].astype(np.int64)])
        res.setdefault("c4", {})["wt_rank_strided"] = strided(csa.wt_rank(gi, gc), NQ_TEXT_FULL)
        m = 20
        st = pkg.rnd_positions(15, NQ_TEXT_FULL, nt - m, 0)[::STRIDE].astype(np.int64)
        pats = np.ascontiguousarray(text[st[:, None] + np.arange(m)[None, :]].reshape(-1))
        res["c4"]["count_strided"] = strided(csa.count_batch(pats, m), NQ_TEXT_FULL)
        print(f"c4s done in {time.time() - t0:.0f}s", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c3s" in want:
        # configs[2] in full: every 100th of the 10^9 rank_1 / select_1 answers of the real rrr_vector<63> on the 5 % vector
        t0 = time.time()
        ck = np.fromfile(CKPT, dtype=np.uint64).reshape(-1, 313)
        w5 = pkg.density_bits(n, 9, 5, ck, CKPT_STRIDE)
        assert hashlib.sha256(w5.tobytes()).hexdigest() == res["c3"]["words_sha256"]
        wp = np.zeros(n // 64 + 2, dtype=np.uint64)
        wp[: n // 64] = w5
        del w5
        h = R.ref_rrr_create(wp.ctypes.data, n)
        print(f"c3s rrr built {time.time() - t0:.0f}s", flush=True)
        ns = (NQ_FULL + STRIDE - 1) // STRIDE
        idx = np.ascontiguousarray(pkg.rnd_positions(7, NQ_FULL, n + 1, 0)[::STRIDE])
        out = np.empty(ns, dtype=np.uint64)
        R.ref_rrr_rank(h, 1, idx.ctypes.data, ns, out.ctypes.data)
        res["c3"]["rank_1_strided"] = strided(out, NQ_FULL)
        del idx
        si = np.ascontiguousarray(pkg.rnd_positions(11, NQ_FULL, res["c3"]["ones"], 1)[::STRIDE])
        R.ref_rrr_select(h, 1, si.ctypes.data, ns, out.ctypes.data)
        res["c3"]["select_1_strided"] = strided(out, NQ_FULL)
        R.ref_rrr_destroy(h)
        del wp, si, out
        print(f"c3s done in {time.time() - t0:.0f}s", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c3" in want:
        t0 = time.time()
        n_ck = (n + CKPT_STRIDE - 1) // CKPT_STRIDE
        ck = pkg.mt_checkpoints(9, CKPT_STRIDE, n_ck)
        w_par = pkg.density_bits(n, 9, 5, ck, CKPT_STRIDE)
        w_ref = np.zeros(n // 64 + 2, dtype=np.uint64)
        R.ref_density_bits(w_ref.ctypes.data, n, 9, 5)  # std::mt19937_64, sequential
        assert np.array_equal(w_par, w_ref[: n // 64]), "checkpointed generator != std::mt19937_64"
        ck.tofile(CKPT)
        del w_par
        print(f"c3 vector {time.time() - t0:.0f}s", flush=True)
        h = R.ref_rrr_create(w_ref.ctypes.data, n)
        print(f"c3 rrr built {time.time() - t0:.0f}s", flush=True)
        idx = pkg.rnd_positions(7, NQ, n + 1, 0)
        out = np.empty(NQ, dtype=np.uint64)
        R.ref_rrr_rank(h, 1, idx.ctypes.data, NQ, out.ctypes.data)
        tot = np.empty(1, dtype=np.uint64)
        R.ref_rrr_rank(h, 1, np.array([n], dtype=np.uint64).ctypes.data, 1, tot.ctypes.data)
        ones = int(tot[0])
        c3 = {"log_n": LOG_N, "bits_seed": 9, "percent": 5, "ones": ones, "checkpoint_stride": CKPT_STRIDE,
              "checkpoints": int(n_ck), "words_sha256": hashlib.sha256(w_ref[: n // 64].tobytes()).hexdigest(),
              "rank_seed": 7, "rank_1": digest(out)}
        si = pkg.rnd_positions(11, NQ, ones, 1)
        R.ref_rrr_select(h, 1, si.ctypes.data, NQ, out.ctypes.data)
        c3.update(select_seed=11, select_1=digest(out))
        R.ref_rrr_destroy(h)
        res["c3"] = c3
        del w_ref, idx, out, si
        print(f"c3 done in {time.time() - t0:.0f}s: ones={ones}", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c4sel" in want:
        # wt_huff<>::select on the same text (wt_pc.hpp:443-474): c = text[mt19937_64(14) % n] as for rank, k = 1 + mt19937_64(16) % occ(c)
        t0 = time.time()
        text = load_text(pkg, text_file)
        nt = int(text.size)
        assert text_file or hashlib.sha256(text.tobytes()).hexdigest() == res["c4"]["text_sha256"]
        wt = ol.RWt(text)
        print(f"c4sel wt built {time.time() - t0:.0f}s", flush=True)
        nqs = int(os.environ.get("GOLDEN_NQ_WTSEL", str(10**6)))
        gc = np.ascontiguousarray(text[pkg.rnd_positions(14, nqs, nt, 0).astype(np.int64)])
        occ = np.bincount(text, minlength=256).astype(np.uint64)
        ks = np.uint64(1) + pkg.rnd_positions(16, nqs, 1 << 62, 0) % occ[gc]
        res["c4"].update(wt_k_seed=16, wt_select=digest(wt.select(ks, gc)))
        print(f"c4sel done in {time.time() - t0:.0f}s", flush=True)
        json.dump(res, open(OUT, "w"))

    if "c4" in want:
        t0 = time.time()
        text = load_text(pkg, text_file)
        nt = int(text.size)
        cnt = np.bincount(text, minlength=256)
        p = cnt[cnt > 0] / nt
        c4 = {"text_log": TEXT_LOG, "text_bytes": nt, "text_file": os.path.basename(text_file) if text_file else None, "text_seed": 1234, "text_sha256": hashlib.sha256(text.tobytes()).hexdigest(),
              "sigma_without_sentinel": int((cnt > 0).sum()), "H0": float(-(p * np.log2(p)).sum())}
        csa = ol.RCsa(text=text.tobytes())
        print(f"c4 csa built {time.time() - t0:.0f}s", flush=True)
        c4["csa_size"] = int(csa.size())
        c4["sigma"] = int(csa.sigma())
        gi = pkg.rnd_positions(13, NQ_TEXT, nt + 2, 0)  # i in [0, size()] with size() = nt + 1 (sentinel)
        gc = text[pkg.rnd_positions(14, NQ_TEXT, nt, 0).astype(np.int64)]
        c4.update(wt_i_seed=13, wt_c_seed=14, wt_rank=digest(csa.wt_rank(gi, np.ascontiguousarray(gc))))
        m = 20
        st = pkg.rnd_positions(15, NQ_TEXT, nt - m, 0).astype(np.int64)
        pats = np.ascontiguousarray(text[st[:, None] + np.arange(m)[None, :]].reshape(-1))
        c4.update(pattern_seed=15, m=m, count=digest(csa.count_batch(pats, m)))
        c4.update({k: v for k, v in res.get("c4", {}).items() if k.endswith("_strided") or k.startswith("wt_select") or k == "wt_k_seed"})
        res["c4"] = c4
        print(f"c4 done in {time.time() - t0:.0f}s", flush=True)
        json.dump(res, open(OUT, "w"))


    if "c4r" in want:
        # the REPETITIVE stand-in (round 6): english_text_repetitive(2^30, 1234, 30) — 30 % of the 64 KiB blocks are rotated copies of earlier
        # ones, so patterns drawn from the text (genpatterns.c:183-203) keep wide SA intervals; first NQ_TEXT answers + every STRIDE-th of 10^8
        t0 = time.time()
        nt = 1 << TEXT_LOG
        pct = int(os.environ.get("GOLDEN_REP_PERCENT", "30"))
        text = pkg.english_text_repetitive(nt, 1234, pct)
        cnt = np.bincount(text, minlength=256)
        p = cnt[cnt > 0] / nt
        c4r = {"text_log": TEXT_LOG, "text_bytes": nt, "text_seed": 1234, "copy_percent": pct, "text_sha256": hashlib.sha256(text.tobytes()).hexdigest(),
               "sigma_without_sentinel": int((cnt > 0).sum()), "H0": float(-(p * np.log2(p)).sum())}
        csa = ol.RCsa(text=text.tobytes())
        print(f"c4r csa built {time.time() - t0:.0f}s", flush=True)
        c4r["csa_size"] = int(csa.size())
        c4r["sigma"] = int(csa.sigma())
        m = 20
        gi = pkg.rnd_positions(13, NQ_TEXT_FULL, nt + 2, 0)
        gci = pkg.rnd_positions(14, NQ_TEXT_FULL, nt, 0).astype(np.int64)
        st = pkg.rnd_positions(15, NQ_TEXT_FULL, nt - m, 0).astype(np.int64)
        c4r.update(wt_i_seed=13, wt_c_seed=14, wt_rank=digest(csa.wt_rank(np.ascontiguousarray(gi[:NQ_TEXT]), np.ascontiguousarray(text[gci[:NQ_TEXT]]))))
        pats = np.ascontiguousarray(text[st[:NQ_TEXT, None] + np.arange(m)[None, :]].reshape(-1))
        cnts = csa.count_batch(pats, m)
        c4r.update(pattern_seed=15, m=m, count=digest(cnts), mean_count=float(np.asarray(cnts, dtype=np.float64).mean()),
                   share_count_1=float((np.asarray(cnts) == 1).mean()))
        c4r["wt_rank_strided"] = strided(csa.wt_rank(np.ascontiguousarray(gi[::STRIDE]), np.ascontiguousarray(text[gci[::STRIDE]])), NQ_TEXT_FULL)
        sts = st[::STRIDE]
        pats = np.ascontiguousarray(text[sts[:, None] + np.arange(m)[None, :]].reshape(-1))
        c4r["count_strided"] = strided(csa.count_batch(pats, m), NQ_TEXT_FULL)
        res["c4r"] = c4r
        print(f"c4r done in {time.time() - t0:.0f}s: mean count {c4r['mean_count']:.3f}, share with count 1: {c4r['share_count_1']:.3f}", flush=True)
        json.dump(res, open(OUT, "w"))


if __name__ == "__main__":
    main()
