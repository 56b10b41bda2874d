"""Stress of the bucketed path (hand tool for gpurun): random vectors, batch sizes and distributions for a fixed time; every
bucketed answer (plain vector and rrr_vector<63> in both record formats) is compared with the direct kernel's.  usage: stress_bucketed.py [seconds=90] [seed=1]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
g = torch.Generator(device="cuda").manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
rounds = 0
seen = {}
queries = 0
while time.time() < t_end:
    logn = int(torch.randint(20, 33, (1,), generator=g, device="cuda"))
    n = (1 << logn) + int(torch.randint(0, 1 << 12, (1,), generator=g, device="cuda"))
    seen[logn] = seen.get(logn, 0) + 1
    dens = [0.5, 0.03, 0.97, 0.2][rounds % 4]
    nw = (n + 63) // 64
    if dens == 0.5:
        w = torch.randint(-2**63, 2**63 - 1, (nw,), device="cuda", dtype=torch.int64, generator=g)
    else:
        w = torch.zeros(nw, dtype=torch.int64, device="cuda")
        for s in range(0, nw, 1 << 21):
            e = min(nw, s + (1 << 21))
            b = (torch.rand((e - s, 64), device="cuda", generator=g) < dens).to(torch.int64)
            w[s:e] = (b << torch.arange(64, device="cuda")).sum(dim=1)
    bv = pkg.bit_vector(w, n, device=0)
    ones = bv.ones()
    for _ in range(3):
        nq = int(torch.randint(1, 1 << int(torch.randint(10, 26, (1,), generator=g, device="cuda")), (1,), generator=g, device="cuda"))
        kind = int(torch.randint(0, 3, (1,), generator=g, device="cuda"))
        idx = torch.randint(0, n + 2, (nq,), device="cuda", dtype=torch.int64, generator=g)
        if kind == 1:
            idx = torch.sort(idx).values
        elif kind == 2:
            idx = (idx % 4096) + n // 2
        bit = rounds & 1
        pkg.set_option("rank_sorted", 0); want = bv.rank(idx, bit)
        pkg.set_option("rank_sorted", 1); got = bv.rank(idx, bit)
        assert torch.equal(got, want), ("rank", logn, n, nq, kind, bit)
        queries += 2 * nq
        tot = ones if bit else n - ones
        if tot >= 2:
            i = torch.randint(0, tot + 2, (nq,), device="cuda", dtype=torch.int64, generator=g)
            pkg.set_option("select_sorted", 0); want = bv.select(i, bit)
            pkg.set_option("select_sorted", 1); got = bv.select(i, bit)
            assert torch.equal(got, want), ("select", logn, n, nq, kind, bit)
    bv.close()
    # the same vector as rrr_vector<63>, in the record format of this round (slim / wide / chosen), bucketed against direct
    if logn <= 30:
        pkg.set_option("rrr_format", [1, 0, -1][rounds % 3])
        rv = pkg.rrr_vector(w, n, device=0)
        pkg.set_option("rrr_format", -1)
        r_ones = rv.ones()
        assert r_ones == ones
        for _ in range(2):
            nq = int(torch.randint(1, 1 << int(torch.randint(10, 24, (1,), generator=g, device="cuda")), (1,), generator=g, device="cuda"))
            idx = torch.randint(0, n + 2, (nq,), device="cuda", dtype=torch.int64, generator=g)
            bit = (rounds >> 1) & 1
            pkg.set_option("rrr_sorted", 0); want = rv.rank(idx, bit)
            pkg.set_option("rrr_sorted", 1); got = rv.rank(idx, bit)
            assert torch.equal(got, want), ("rrr rank", logn, n, nq, bit, dens)
            tot = ones if bit else n - ones
            if tot >= 2:
                i = torch.randint(0, tot + 2, (nq,), device="cuda", dtype=torch.int64, generator=g)
                pkg.set_option("rrr_sorted", 0); want = rv.select(i, bit)
                pkg.set_option("rrr_sorted", 1); got = rv.select(i, bit)
                assert torch.equal(got, want), ("rrr select", logn, n, nq, bit, dens)
        rv.close()
    rounds += 1
pkg.set_option("rank_sorted", -1); pkg.set_option("select_sorted", -1); pkg.set_option("rrr_sorted", -1)
print(f"stress ok: {rounds} vectors in {budget:.0f} s; log2 sizes {dict(sorted(seen.items()))}; {queries:.3g} rank queries on the plain vectors alone")
