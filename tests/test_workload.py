"""CPU: the SURVEY.md 8(d) workload generators of the library (csrc/workload.cpp) — integer-only and seeded, so that the
build container and the GPU box hold identical inputs — against the C restatement of std::mt19937_64 in oracle/ and the
committed fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rnd_positions_is_std_mt19937_64(pkg):
    import oracle_lib as ol
    raw = pkg.rnd_positions(815, 100_000)
    assert np.array_equal(raw, ol.mt19937_64(100_000, 815))
    n = (1 << 34) + 1
    assert np.array_equal(pkg.rnd_positions(7, 300_001, n, 0), ol.mt19937_64(300_001, 7) % np.uint64(n))
    assert np.array_equal(pkg.rnd_positions(11, 1000, 12345, 1), ol.mt19937_64(1000, 11) % np.uint64(12345) + np.uint64(1))
    assert pkg.rnd_positions(1, 0).size == 0


def test_density_bits_sequential_and_checkpointed_agree(pkg):
    import oracle_lib as ol
    n = (1 << 22) + 77
    w = pkg.density_bits(n, 9, 5)
    ref = ol.mt19937_64(n, 9) % np.uint64(100) < 5
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(bits, ref)
    assert not np.unpackbits(w.view(np.uint8), bitorder="little")[n:].any(), "bits past the end must be clear"
    ck = pkg.mt_checkpoints(9, 1 << 18, 17)
    assert np.array_equal(pkg.density_bits(n, 9, 5, ck, 1 << 18), w)
    assert np.array_equal(pkg.density_bits(n, 9, 5, ck[:3], 1 << 18), w), "stretches past the last checkpoint continue from it"
    assert hashlib.sha256(pkg.density_bits(1 << 20, 9, 5).tobytes()).hexdigest() == \
        "cf7070183abdb86ae7c9b4a6949cc8bd7a8874633d77e97dbdf779860df343d0"


def test_committed_checkpoints_are_states_of_mt19937_64_9(pkg):
    g = json.load(open(os.path.join(HERE, "golden", "golden_large.json")))["c3"]
    ck = np.fromfile(os.path.join(HERE, "golden", "mt9_checkpoints.bin"), dtype=np.uint64).reshape(-1, 313)
    assert ck.shape[0] == g["checkpoints"] == (1 << g["log_n"]) // g["checkpoint_stride"]
    # the first two are regenerated here (2^28 draws); all of them were validated against std::mt19937_64 over the whole
    # 2^34-bit vector when the fixture was made (make_golden_large.py), and the GPU test checks the vector's sha256
    assert np.array_equal(pkg.mt_checkpoints(g["bits_seed"], g["checkpoint_stride"], 2), ck[:2])


def test_english_text_is_reproducible_and_english_class(pkg):
    t = pkg.english_text(1 << 20, 1234)
    assert hashlib.sha256(t.tobytes()).hexdigest() == "e5b8e6d42ee7c9c234c07cd1159eec298e353bd8628773b64f1f402c19758b57"
    assert np.array_equal(pkg.english_text(1 << 18, 1234), t[: 1 << 18]), "a prefix of the text is the text of that length"
    big = pkg.english_text(1 << 24, 1234)
    cnt = np.bincount(big, minlength=256)
    assert cnt[0] == 0, "no zero byte (construct.hpp rejects it)"
    p = cnt[cnt > 0] / big.size
    h0 = float(-(p * np.log2(p)).sum())
    assert (cnt > 0).sum() >= 200 and 4.5 <= h0 <= 4.8, ((cnt > 0).sum(), h0)
    assert not np.array_equal(pkg.english_text(1 << 12, 1), pkg.english_text(1 << 12, 2))


def test_golden_large_has_every_config(pkg):
    g = json.load(open(os.path.join(HERE, "golden", "golden_large.json")))
    assert g["c2"]["log_n"] == 34 and g["c3"]["log_n"] == 34 and g["c4"]["text_log"] == 30
    for c, keys in (("c2", ("rank_1", "select_1")), ("c3", ("rank_1", "select_1")), ("c4", ("wt_rank", "count"))):
        for k in keys:
            assert len(g[c][k]["first"]) == 10_000 and len(g[c][k]["sha256"]) == 64
    assert g["c4"]["sigma"] >= 200 and 4.5 <= g["c4"]["H0"] <= 4.8


def test_the_repetitive_text_copies_whole_blocks_and_stays_prefix_stable(pkg):
    """english_text_repetitive: percent 0 is english_text; with 30 % copies a prefix of the text is the text of that length, no zero byte,
    the alphabet statistics stay English-class, and a copied block is a ROTATION of the (chain of) earlier block(s) it copies"""
    import hashlib
    n = 1 << 23
    plain = pkg.english_text(n, 1234)
    rep = pkg.english_text_repetitive(n, 1234, 30)
    assert np.array_equal(pkg.english_text_repetitive(1 << 20, 1234, 0), plain[: 1 << 20])
    assert np.array_equal(pkg.english_text_repetitive(1 << 20, 1234, 30), rep[: 1 << 20]) and (rep != 0).all()
    blk = 1 << 16
    nb = n // blk
    same = np.array([np.array_equal(rep[b * blk:(b + 1) * blk], plain[b * blk:(b + 1) * blk]) for b in range(nb)])
    assert same[0] and 0.55 < same.mean() < 0.85, same.mean()
    # every copied block is a rotation of an ORIGINAL block in front of it
    digests = {hashlib.sha256(np.sort(plain[b * blk:(b + 1) * blk]).tobytes()).hexdigest(): b for b in range(nb) if same[b]}
    for b in np.flatnonzero(~same)[:20]:
        src = digests.get(hashlib.sha256(np.sort(rep[b * blk:(b + 1) * blk]).tobytes()).hexdigest())
        assert src is not None and src < b, b
        s = plain[src * blk:(src + 1) * blk]
        dbl = np.concatenate([s, s]).tobytes()
        assert rep[b * blk:(b + 1) * blk].tobytes() in dbl, "a rotation of its source"
    cnt = np.bincount(rep, minlength=256)
    p = cnt[cnt > 0] / n
    assert (cnt > 0).sum() > 200 and 4.3 < float(-(p * np.log2(p)).sum()) < 4.8
    with pytest.raises(Exception):
        pkg.english_text_repetitive(1 << 12, 1, 96)
