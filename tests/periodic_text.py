"""A text whose suffix array is known in closed form, so that an FM-index of MORE than 2^32 symbols can be checked without
sorting 2^32 suffixes on the host: T = U^k, U = V + '#', where '#' (byte 1) occurs nowhere in V and is smaller than V's bytes.

Two suffixes from different offsets inside U differ before either of them has passed its first '#', so their order is that of
the suffixes of U alone; of two suffixes from the same offset the one that starts in a later copy is a proper prefix of the
other followed by the sentinel, hence smaller.  With SAu = suffix array of U (p entries) the suffix array of T$ is therefore

    SA[0] = n (the sentinel),   SA[1 + r * k + t] = SAu[r] + (k - 1 - t) * p      (r < p, t < k)

and BWT[x] = T[SA[x] - 1] is k times the byte in front of offset SAu[r] — '#' for offset 0, whose first copy is preceded by the
sentinel.  `tests/test_periodic_text.py` checks all of it against the oracle (and through it the real library) at small sizes."""
import numpy as np

HASH = 1


def unit(p: int, sigma: int, seed: int) -> np.ndarray:
    """U: p - 1 bytes in 2 .. sigma + 1 followed by '#'; a few repeats inside so that patterns occur more than once per copy"""
    rng = np.random.default_rng(seed)
    v = rng.integers(2, sigma + 2, p - 1, dtype=np.uint8)
    if p > 4096:
        v[1000:1400] = v[3000:3400]
        v[p // 2:p // 2 + 64] = v[1100:1164]
    return np.concatenate([v, np.array([HASH], dtype=np.uint8)])


def unit_suffix_array(u: np.ndarray, ocsa_cls) -> np.ndarray:
    """suffix array of U (offsets in increasing order of U[i:]), through the oracle's csa of U (drop the sentinel's entry)"""
    c = ocsa_cls(bytes(u))
    sa = np.asarray(c.sa(np.arange(u.size + 1, dtype=np.uint64))).astype(np.int64)
    assert sa[0] == u.size
    return sa[1:]


def bwt_rows(u: np.ndarray, sau: np.ndarray) -> np.ndarray:
    """by rank r of an offset: the byte in front of it ('#' in front of offset 0)"""
    return np.where(sau > 0, u[sau - 1], HASH).astype(np.uint8)


def bwt_host(u: np.ndarray, sau: np.ndarray, k: int) -> np.ndarray:
    out = np.empty(1 + u.size * k, dtype=np.uint8)
    out[0] = HASH
    out[1:] = np.repeat(bwt_rows(u, sau), k)
    r0 = int(np.flatnonzero(sau == 0)[0])
    out[1 + r0 * k + (k - 1)] = 0  # the suffix that is all of T: preceded by the sentinel
    return out


def bwt_device(u: np.ndarray, sau: np.ndarray, k: int):
    import torch
    rows = torch.from_numpy(bwt_rows(u, sau)).cuda()
    out = torch.empty(1 + u.size * k, dtype=torch.uint8, device="cuda")
    out[0] = HASH
    out[1:] = torch.repeat_interleave(rows, k)
    r0 = int(np.flatnonzero(sau == 0)[0])
    out[1 + r0 * k + (k - 1)] = 0
    return out


def count_in_text(u: np.ndarray, k: int, pats: np.ndarray) -> np.ndarray:
    """occurrences of every row of pats (m <= p bytes, no 0) in T = U^k: starts in the copies 0 .. k-2 may run into the next
    copy, starts in the last one may not"""
    p, m = u.size, pats.shape[1]
    uu = np.concatenate([u, u])
    w = min(m, 8)                                      # candidates by the first w bytes as one number, then compared in full
    weights = (np.uint64(256) ** np.arange(w, dtype=np.uint64)).astype(np.uint64)
    keys = (np.lib.stride_tricks.sliding_window_view(uu[:p + w - 1], w).astype(np.uint64) * weights).sum(axis=1, dtype=np.uint64)
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    pk = (pats[:, :w].astype(np.uint64) * weights).sum(axis=1, dtype=np.uint64)
    lo, hi = np.searchsorted(sk, pk, "left"), np.searchsorted(sk, pk, "right")
    out = np.zeros(pats.shape[0], dtype=np.uint64)
    for q in range(pats.shape[0]):
        cand = order[lo[q]:hi[q]]
        if m > w and cand.size:
            cand = cand[(uu[cand[:, None] + np.arange(w, m)[None, :]] == pats[q, w:][None, :]).all(axis=1)]
        out[q] = (k - 1) * cand.size + int((cand + m <= p).sum())
    return out


def text_at(u: np.ndarray, pos: np.ndarray, m: int) -> np.ndarray:
    """T[pos : pos + m] for every pos (the caller keeps pos + m <= n)"""
    return u[(pos[:, None] + np.arange(m)[None, :]) % u.size]
