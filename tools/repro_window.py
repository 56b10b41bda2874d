"""Forced bucketed rank on a batch confined to a window, against the direct kernel (hand tool)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
logn, nq, wlog = int(sys.argv[1]), int(float(sys.argv[2])), int(sys.argv[3])
n = (1 << logn) - 37
g = torch.Generator(device="cuda").manual_seed(42)
words = torch.randint(-2**63, 2**63 - 1, ((n + 63) // 64,), device="cuda", dtype=torch.int64, generator=g)
bv = pkg.bit_vector(words, n)
del words
idx = (n // 3) + torch.randint(0, 1 << wlog, (nq,), device="cuda", dtype=torch.int64, generator=g)
pkg.set_option("rank_sorted", 0)
want = bv.rank(idx, 1).clone()
pkg.set_option("rank_sorted", 1)
got = bv.rank(idx, 1)
torch.cuda.synchronize()
bad = (got != want).nonzero().flatten()
print(f"logn {logn} nq {nq} window 2^{wlog}: mismatches {bad.numel()}")
if bad.numel():
    b = bad[:8].cpu()
    print("first bad idx", b.tolist(), "last", int(bad[-1]))
    print("got ", got[b.cuda()].tolist())
    print("want", want[b.cuda()].tolist())
    print("pos ", idx[b.cuda()].tolist())
    # is the answer some other query's answer?
    d = (got[bad] - want[bad])
    print("diff min/max", int(d.min()), int(d.max()))
    tiles = torch.unique(bad // 8192)
    print("bad tiles", tiles.numel(), tiles[:12].tolist(), "items", torch.unique(tiles // 8).tolist()[:12])
