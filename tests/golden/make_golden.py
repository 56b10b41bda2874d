"""Generates the committed golden fixtures from the REAL sdsl-lite (run in the build container only).

    python tests/golden/make_golden.py

Needs /root/reference (the read-only reference tree) and oracle/_ref/libsdsl_ref.so (built by
`make -C oracle`).  Everything written here is DATA: inputs the reference's own tests use
(test/test_cases/*, the bit vectors produced by test/bit_vector_generator.cpp compiled where it
lies) and outputs of the real library on seeded inputs (answers + serialised streams).  No
reference source text is stored.  Query streams come from std::mt19937_64 (restated in
oracle.c and cross-checked against std:: here), so they are reproducible on any machine.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import tarfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

REF = os.environ.get("SDSL_REF", "/root/reference")
BV_IDS = ["CRAFTED-32", "CRAFTED-SPARSE-0", "CRAFTED-SPARSE-1", "CRAFTED-BLOCK-0", "CRAFTED-BLOCK-1",
          "CRAFTED-MAT-SELECT"]
TEXTS = ["100a.txt", "abc_abc_abc.txt", "abc_abc_abc2.txt", "all_symbols.txt", "empty.txt", "example01.txt",
         "one_byte.txt"]


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def read_sdsl_bitvector(path):
    raw = open(path, "rb").read()
    hdr = int.from_bytes(raw[:8], "little")
    n = hdr & ((1 << 56) - 1)
    words = np.frombuffer(raw[8:], dtype=np.uint64).copy()
    assert words.size == (n + 63) // 64
    return words, n


def queries(n_q, mod, seed):
    return (ol.mt19937_64(n_q, seed) % np.uint64(mod)).astype(np.uint64) if mod else np.zeros(0, np.uint64)


def main():
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    # std::mt19937_64 restatement check against the real library's util::set_random_bits
    w_ref = np.zeros(16384, dtype=np.uint64)
    ol.ref().L.ref_set_random_bits(w_ref.ctypes.data, 1 << 20, 815)
    assert np.array_equal(w_ref, ol.set_random_bits(1 << 20, 815))

    # 1. the reference's generated bit-vector fixtures ------------------------------------------------
    gen = os.path.join(ROOT, "oracle", "_ref", "bit_vector_generator")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", f"-I{REF}/include",
                           os.path.join(REF, "test", "bit_vector_generator.cpp"), "-o", gen])
    os.makedirs(os.path.join(HERE, "bitvec"), exist_ok=True)
    for bid in BV_IDS:
        subprocess.check_call([gen, os.path.join(HERE, "bitvec", f"bit-vec.{bid}"), bid])

    # 2. test texts (data files of the reference's tests) ---------------------------------------------
    os.makedirs(os.path.join(HERE, "texts"), exist_ok=True)
    for t in TEXTS:
        data = open(os.path.join(REF, "test", "test_cases", t), "rb").read()
        open(os.path.join(HERE, "texts", t), "wb").write(data)
    with tarfile.open(os.path.join(REF, "test", "test_cases", "faust.txt.tar.gz")) as tf:
        data = tf.extractfile("faust.txt").read()
        open(os.path.join(HERE, "texts", "faust.txt"), "wb").write(data)

    # 3. plain bit vector + rrr answers -----------------------------------------------------------------
    cases = {}
    for bid in BV_IDS:
        cases[bid] = read_sdsl_bitvector(os.path.join(HERE, "bitvec", f"bit-vec.{bid}"))
    # intended semantics of the int-vec.N.1.r.SEED entries of test/rank_support_test.config
    for n, seed in [(8, 17), (16, 42), (32, 111), (64, 222), (128, 73), (256, 4887), (512, 432), (1024, 898),
                    (2048, 5432), (4096, 793), (8192, 1043), (1000000, 815), (1 << 20, 815), (200000, 7),
                    (63 * 32 * 5, 9)]:
        cases[f"rnd.{n}.{seed}"] = (ol.set_random_bits(n, seed), n)
    out = {}
    for name, (w, n) in cases.items():
        rb = ol.RBitVector(w, n)
        rr = ol.RRrr(w, n)
        idx = np.concatenate([queries(2048, n + 1, 17), np.array([0, n, n // 2], dtype=np.uint64)])
        out[f"{name}/n"] = np.array([n], dtype=np.uint64)
        out[f"{name}/idx"] = idx
        for b in (0, 1):
            out[f"{name}/rank{b}"] = rb.rank(idx, b)
            tot = int(rb.rank(np.array([n], dtype=np.uint64), b)[0])
            out[f"{name}/total{b}"] = np.array([tot], dtype=np.uint64)
            si = (queries(2048, tot, 11 + b) + np.uint64(1)) if tot else np.zeros(0, np.uint64)
            if tot:
                si = np.concatenate([si, np.array([1, tot], dtype=np.uint64)])
            out[f"{name}/sel{b}_i"] = si
            out[f"{name}/sel{b}"] = rb.select(si, b) if tot else np.zeros(0, np.uint64)
            assert np.array_equal(rr.rank(idx, b), out[f"{name}/rank{b}"])
            sio = np.concatenate([si, np.array([tot + 1], dtype=np.uint64)])  # overflow -> size()
            out[f"{name}/rrr_sel{b}_i"] = sio
            out[f"{name}/rrr_sel{b}"] = rr.select(sio, b)
        if name in ("CRAFTED-32", "CRAFTED-MAT-SELECT", "rnd.8192.1043", "rnd.200000.7", "rnd.1000000.815", "rnd.64.222"):
            # two-bit pattern supports: rank_support_v5<pat,2> and select_support_mcl<pat,2> of the real library
            for pat in range(4):
                out[f"{name}/pat{pat}_rank"] = ol.ref_bv_pattern(w, n, pat, 0, idx)
                tot = int(ol.ref_bv_pattern(w, n, pat, 1, np.array([n], dtype=np.uint64))[0])
                si = (queries(1024, tot, 50 + pat) + np.uint64(1)) if tot else np.zeros(0, np.uint64)
                if tot:
                    si = np.concatenate([si, np.array([1, tot], dtype=np.uint64)])
                out[f"{name}/pat{pat}_sel_i"] = si
                out[f"{name}/pat{pat}_sel"] = ol.ref_bv_pattern(w, n, pat, 2, si) if tot else np.zeros(0, np.uint64)
        if name in ("CRAFTED-32", "CRAFTED-SPARSE-0", "CRAFTED-SPARSE-1", "CRAFTED-BLOCK-1", "rnd.8192.1043", "rnd.200000.7",
                    "rnd.64.222", "rnd.8.17"):
            # sd_vector<> of the real library on the same bits: rank_1, select_1, select_0 (i <= #zeros), operator[]
            sd = ol.RSd(w, n)
            out[f"{name}/sd_rank1"] = sd.rank(idx, 1)
            for b in (0, 1):
                si = out[f"{name}/sel{b}_i"][:600]
                out[f"{name}/sd_sel{b}"] = sd.select(si, b) if si.size else np.zeros(0, np.uint64)
            ai = idx[idx < n]
            out[f"{name}/sd_acc"] = sd.access(ai) if ai.size else np.zeros(0, np.uint8)
            if name in ("CRAFTED-32", "rnd.8192.1043", "rnd.200000.7", "CRAFTED-SPARSE-1"):
                open(os.path.join(HERE, "sdsl", f"{name}.sd_vector.sdsl"), "wb").write(sd.serialize())
        out[f"{name}/sha"] = np.array([sha(rb.serialize(1)), sha(rb.serialize(2)), sha(rb.serialize(3)),
                                       sha(rb.serialize(4)), sha(rr.serialize())])
        out[f"{name}/sha_more"] = np.array([sha(rb.serialize(0)), sha(rb.serialize(5)), sha(rb.serialize(6))])
    # a sparse set over a large universe (what sd_vector is for): 20000 positions below 2^40
    pos = np.unique(ol.mt19937_64(20000, 4242) % np.uint64(1 << 40)).astype(np.uint64)
    sd = ol.RSd(positions=pos)
    N = sd.size()
    qi = np.concatenate([ol.mt19937_64(3000, 4243) % np.uint64(N + 1), pos[:500], pos[:500] + np.uint64(1),
                         np.array([0, N], dtype=np.uint64)]).astype(np.uint64)
    out["sdpos/pos"], out["sdpos/n"], out["sdpos/idx"] = pos, np.array([N], dtype=np.uint64), qi
    out["sdpos/rank1"] = sd.rank(qi, 1)
    s1 = (ol.mt19937_64(2000, 4244) % np.uint64(pos.size)) + np.uint64(1)
    out["sdpos/sel1_i"], out["sdpos/sel1"] = s1, sd.select(s1, 1)
    s0 = (ol.mt19937_64(500, 4245) % np.uint64(N - pos.size)) + np.uint64(1)
    out["sdpos/sel0_i"], out["sdpos/sel0"] = s0, sd.select(s0, 0)
    out["sdpos/acc"] = sd.access(qi[qi < N])
    open(os.path.join(HERE, "sdsl", "sdpos.sd_vector.sdsl"), "wb").write(sd.serialize())
    np.savez_compressed(os.path.join(HERE, "golden_bitvectors.npz"), **out)

    # serialised streams for the loader tests (small ones only)
    os.makedirs(os.path.join(HERE, "sdsl"), exist_ok=True)
    for name in ["CRAFTED-32", "rnd.8192.1043", "rnd.200000.7", f"rnd.{63*32*5}.9"]:
        w, n = cases[name]
        open(os.path.join(HERE, "sdsl", f"{name}.rrr63.sdsl"), "wb").write(ol.RRrr(w, n).serialize())
        rb = ol.RBitVector(w, n)
        for which, tag in [(0, "bv"), (1, "rank_v5_1"), (3, "select_mcl_1"), (4, "select_mcl_0")]:
            if name in ("CRAFTED-32", "rnd.200000.7"):
                open(os.path.join(HERE, "sdsl", f"{name}.{tag}.sdsl"), "wb").write(rb.serialize(which))

    # 4. wavelet tree + FM-index answers ----------------------------------------------------------------
    out = {}
    for t in TEXTS + ["faust.txt"]:
        data = open(os.path.join(HERE, "texts", t), "rb").read()
        n = len(data)
        wt = ol.RWt(data)
        arr = np.frombuffer(data, dtype=np.uint8)
        qi = np.concatenate([queries(3000, n + 1, 5), np.array([0, n], dtype=np.uint64)])
        qc = (ol.mt19937_64(qi.size, 6) % np.uint64(256)).astype(np.uint8)
        if n:
            take = queries(qi.size, n, 8)
            qc[: qi.size // 2] = arr[take[: qi.size // 2]]  # half of the symbols drawn from the text
        out[f"{t}/meta"] = np.array([n, wt.sigma(), wt.bv_size()], dtype=np.uint64)
        out[f"{t}/rank_i"], out[f"{t}/rank_c"] = qi, qc
        out[f"{t}/rank"] = wt.rank(qi, qc)
        out[f"{t}/rank_full"] = wt.rank(np.full(256, n, dtype=np.uint64), np.arange(256, dtype=np.uint8))
        if n:
            ai = queries(1000, n, 9)
            out[f"{t}/acc_i"] = ai
            out[f"{t}/acc"] = wt.access(ai)
            r, c = wt.inverse_select(ai)
            out[f"{t}/invsel_rank"] = r
            sc = arr[queries(500, n, 10)]
            tot = wt.rank(np.full(sc.size, n, dtype=np.uint64), sc)
            si = (ol.mt19937_64(sc.size, 12) % tot) + np.uint64(1)
            out[f"{t}/sel_i"], out[f"{t}/sel_c"] = si, sc
            out[f"{t}/sel"] = wt.select(si, sc)
        out[f"{t}/sha"] = np.array([sha(wt.serialize(1)), sha(wt.serialize(0))])
        out[f"{t}/sha_default"] = np.array([sha(ol.ref_wt_default_bytes(data))])
        if t in ("example01.txt", "abc_abc_abc.txt", "100a.txt", "one_byte.txt"):
            open(os.path.join(HERE, "sdsl", f"{t}.wt_huff_v5_mcl.sdsl"), "wb").write(wt.serialize(1))
            open(os.path.join(HERE, "sdsl", f"{t}.wt_huff_v5_scan.sdsl"), "wb").write(wt.serialize(0))
        if b"\0" in data or n == 0:
            continue
        csa = ol.RCsa(data, also_fm_huff=True)
        out[f"{t}/csa_meta"] = np.array([csa.size(), csa.sigma()], dtype=np.uint64)
        out[f"{t}/bwt_sha"] = np.array([sha(csa.bwt().tobytes())])
        c2c, Cc = csa.alphabet()
        out[f"{t}/char2comp"], out[f"{t}/C"] = c2c, Cc
        for m in (1, 2, 4, 20):
            if n < m:
                continue
            st = queries(1500, n - m + 1, 20 + m)
            pats = np.concatenate([arr[int(s):int(s) + m] for s in st])
            mut = pats.copy()  # every 5th byte replaced: mixes absent characters and misses in
            mut[::5] = (ol.mt19937_64(mut[::5].size, 30 + m) % np.uint64(255) + np.uint64(1)).astype(np.uint8)
            allp = np.concatenate([pats, mut])
            out[f"{t}/pat{m}"] = allp
            out[f"{t}/count{m}"] = csa.count_batch(allp, m)
            l, r = csa.interval_batch(allp, m)
            out[f"{t}/ival_l{m}"], out[f"{t}/ival_r{m}"] = l, r
        # the rest of the csa_wt API (default densities 32 / 64): SA, ISA, LF, psi, extract, locate
        N = csa.size()
        ai = np.concatenate([queries(1500, N, 40), np.array([0, N - 1, N // 2], dtype=np.uint64)])
        out[f"{t}/csa_idx"] = ai
        out[f"{t}/csa_sa"], out[f"{t}/csa_isa"] = csa.sa(ai), csa.isa(ai)
        out[f"{t}/csa_lf"], out[f"{t}/csa_psi"] = csa.lf(ai), csa.psi(ai)
        eb = queries(300, N, 41)
        ee = np.minimum(eb + queries(300, 120, 42), np.uint64(N - 1))
        eb = np.concatenate([eb, np.array([0, N - 1, 0], dtype=np.uint64)])
        ee = np.concatenate([ee, np.array([0, N - 1, min(N - 1, 700)], dtype=np.uint64)])
        out[f"{t}/ext_b"], out[f"{t}/ext_e"] = eb, ee
        out[f"{t}/ext_text"] = np.frombuffer(b"".join(csa.extract(int(b), int(e)) for b, e in zip(eb, ee)),
                                             dtype=np.uint8)
        for m, cnt in ((2, 200), (4, 600), (20, 3000)):
            if n < m:
                continue
            allp = out[f"{t}/pat{m}"]
            k = min(cnt, allp.size // m)
            locs = [csa.locate(allp[i * m:(i + 1) * m].tobytes()) for i in range(k)]
            out[f"{t}/loc_n{m}"] = np.array([k], dtype=np.uint64)
            out[f"{t}/loc_off{m}"] = np.concatenate([[0], np.cumsum([x.size for x in locs])]).astype(np.uint64)
            out[f"{t}/loc_pos{m}"] = np.concatenate(locs).astype(np.uint64) if locs else np.zeros(0, np.uint64)
        out[f"{t}/sha_csa_default"] = np.array([sha(ol.ref_csa_default_bytes(data)), sha(csa.serialize(0))])
        if t in ("example01.txt", "faust.txt"):
            open(os.path.join(HERE, "sdsl", f"{t}.csa_wt_huff_v5.sdsl"), "wb").write(csa.serialize(0))
            open(os.path.join(HERE, "sdsl", f"{t}.csa_fm_huff.sdsl"), "wb").write(csa.serialize(1))
            # compressed flavour: wt_huff<rrr_vector<63>> and csa_wt<wt_huff<rrr_vector<63>>, 32, 64>
            open(os.path.join(HERE, "sdsl", f"{t}.wt_huff_rrr63.sdsl"), "wb").write(ol.ref_wt_rrr_bytes(data))
            open(os.path.join(HERE, "sdsl", f"{t}.csa_wt_huff_rrr63.sdsl"), "wb").write(ol.ref_csa_rrr_bytes(data))
    np.savez_compressed(os.path.join(HERE, "golden_text.npz"), **out)

    # 5. other wt_pc shapes: wt_blcd / wt_hutu streams of the real library (answers do not depend on the shape)
    for name, data in (("example01.txt", open(os.path.join(HERE, "texts", "example01.txt"), "rb").read()),
                       ("faust60k", open(os.path.join(HERE, "texts", "faust.txt"), "rb").read()[:60000])):
        open(os.path.join(HERE, "sdsl", f"{name}.wt_blcd.sdsl"), "wb").write(ol.ref_wt_shape_bytes(data, 1, 0))
        open(os.path.join(HERE, "sdsl", f"{name}.wt_hutu.sdsl"), "wb").write(ol.ref_wt_shape_bytes(data, 2, 0))
        open(os.path.join(HERE, "sdsl", f"{name}.wt_blcd_v5_scan.sdsl"), "wb").write(ol.ref_wt_shape_bytes(data, 1, 1))
        open(os.path.join(HERE, "sdsl", f"{name}.csa_wt_blcd_v5_scan.sdsl"), "wb").write(ol.ref_csa_blcd_bytes(data))
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
