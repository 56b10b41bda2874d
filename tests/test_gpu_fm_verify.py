"""count() stops walking a pattern once ONE suffix is left and compares the remaining characters with the text at SA[l]
(fm.hip: k_fm_count<verify> + k_fm_verify).  Every way that can go: the pattern occurs once, it differs from the text in front
of the unique suffix, the unique suffix stands too close to the text's start, variable lengths, patterns that never get unique."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rrr", [False, True])
@pytest.mark.parametrize("sigma,n", [(4, 50_000), (26, 200_000), (200, 300_000)])
def test_count_with_text_verification_equals_oracle(gpu, sigma, n, rrr):
    rng = np.random.default_rng(sigma * 1000 + 7)
    text = rng.integers(1, sigma + 1, n, dtype=np.uint8)
    text[1000:1200] = text[5000:5200]  # a long repeat: intervals of size two deep into the pattern
    csa = gpu.csa_wt(text=text, rrr=rrr)  # (rrr: csa_wt<wt_huff<rrr_vector<63>>>, k_fm_count_rrr<verify>)
    ocsa = ol.OCsa(bytes(text))
    for m in (2, 3, 8, 20, 41):
        npat = 30_000
        st = rng.integers(0, n - m, npat)
        st[:200] = np.arange(200) % 7            # cut at the very start of the text
        pats = text[st[:, None] + np.arange(m)[None, :]].copy()
        mut = rng.random(npat) < 0.5             # half of them differ from the text in one place
        where = rng.integers(0, m, npat)
        pats[mut, where[mut]] = rng.integers(1, sigma + 1, int(mut.sum()), dtype=np.uint8)
        # a unique suffix too close to the start: "?? + text[0 : m - 2]"
        pats[200:260, 2:] = text[: m - 2] if m > 2 else pats[200:260, 2:]
        flat = np.ascontiguousarray(pats.reshape(-1))
        got = csa.count(flat, m)
        want = ocsa.count_batch(flat, m)
        assert np.array_equal(np.asarray(got).astype(np.uint64), np.asarray(want).astype(np.uint64)), f"sigma {sigma}, m {m}"
    csa.close()
