"""tests/periodic_text.py against the oracle at sizes a suffix sorter finishes in no time: the closed-form suffix array and BWT of
T = (V#)^k are the oracle's (csa_wt over T, pinned to the real library by test_oracle_vs_ref.py), and so are the counts."""
import numpy as np
import pytest

import oracle_lib as ol
import periodic_text as pt


@pytest.mark.parametrize("p,k,sigma", [(2, 3, 1), (5, 1, 2), (7, 4, 2), (64, 9, 3), (300, 17, 5), (5000, 6, 40)])
def test_closed_form_suffix_array_and_bwt(p, k, sigma):
    u = pt.unit(p, sigma, p + k)
    sau = pt.unit_suffix_array(u, ol.OCsa)
    t = np.tile(u, k)
    c = ol.OCsa(bytes(t))
    n = t.size
    sa = np.asarray(c.sa(np.arange(n + 1, dtype=np.uint64))).astype(np.int64)
    want = np.empty(n + 1, dtype=np.int64)
    want[0] = n
    want[1:] = (sau[:, None] + (k - 1 - np.arange(k))[None, :] * p).reshape(-1)
    assert np.array_equal(sa, want)
    assert np.array_equal(np.asarray(c.bwt()), pt.bwt_host(u, sau, k))


def test_counts_in_the_periodic_text():
    p, k, sigma = 3000, 7, 4
    u = pt.unit(p, sigma, 1)
    c = ol.OCsa(bytes(np.tile(u, k)))
    rng = np.random.default_rng(2)
    for m in (1, 3, 8, 40):
        pos = rng.integers(0, p * k - m, 300)
        pats = pt.text_at(u, pos, m).copy()
        pats[::3, rng.integers(0, m)] = rng.integers(1, sigma + 2)   # (1 = '#': patterns across a copy's end)
        want = np.array([c.count(bytes(r)) for r in pats], dtype=np.uint64)
        assert np.array_equal(pt.count_in_text(u, k, pats), want), m
