"""Bucketed rank on rrr_vector<63> against the direct kernel: equality + timing (hand tool for gpurun).
usage: rrr_sorted_probe.py <log2 bits> <queries> [density]"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
logn = int(sys.argv[1]); nq = int(float(sys.argv[2])); dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
n = (1 << logn) - 41
dev = "cuda"
gw = torch.Generator(device=dev).manual_seed(9)
nw = (n + 63) // 64
w = torch.empty(nw, dtype=torch.int64, device=dev)
weights = (torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(64, device=dev)).view(1, 64)
for s in range(0, nw, 1 << 22):
    e = min(nw, s + (1 << 22))
    w[s:e] = ((torch.rand((e - s, 64), device=dev, generator=gw) < dens).to(torch.int64) * weights).sum(dim=1)
rv = pkg.rrr_vector(w, n)
del w
print(f"n=2^{logn}-41 dens={dens} ones={rv.ones()} bits/bit={rv.device_bytes()*8/n:.3f}", flush=True)
idx = torch.randint(0, n + 1, (nq,), device=dev, dtype=torch.int64, generator=gw)
idx[:6] = torch.tensor([0, n, n + 1, 2**62, 1, n - 1], device=dev)
out = torch.empty_like(idx)
pkg.set_timing(True)
res = {}
for mode in (0, 1):
    pkg.set_option("rrr_sorted", mode)
    for bit in (1, 0):
        rv.rank(idx, bit, out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            rv.rank(idx, bit, out); ts.append(pkg.last_kernel_ms())
        ms = min(ts)
        res[(mode, bit)] = out.clone()
        print(f"mode {mode} rank{bit}: {ms:.3f} ms {nq/ms/1e6:.2f} G/s frac {144*nq/ms/1e6/8000:.3f}", flush=True)
for bit in (1, 0):
    same = torch.equal(res[(0, bit)], res[(1, bit)])
    print(f"bit {bit}: bucketed == direct: {same}")
    if not same:
        bad = (res[(0, bit)] != res[(1, bit)]).nonzero().flatten()
        print("  mismatches", bad.numel(), bad[:5].tolist(), idx[bad[:5]].tolist(), res[(0, bit)][bad[:5]].tolist(), res[(1, bit)][bad[:5]].tolist())
for bit in (1, 0):
    tot = rv.ones() if bit else n - rv.ones()
    si = torch.randint(1, tot + 1, (nq,), device=dev, dtype=torch.int64, generator=gw)
    si[:5] = torch.tensor([1, tot, tot + 1, 0, 2], device=dev)
    r = {}
    for mode in (0, 1):
        pkg.set_option("rrr_sorted", mode)
        rv.select(si, bit, out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            rv.select(si, bit, out); ts.append(pkg.last_kernel_ms())
        ms = min(ts)
        r[mode] = out.clone()
        print(f"mode {mode} select{bit}: {ms:.3f} ms {nq/ms/1e6:.2f} G/s frac {144*nq/ms/1e6/8000:.3f}", flush=True)
    same = torch.equal(r[0], r[1])
    print(f"select{bit}: bucketed == direct: {same}")
    if not same:
        bad = (r[0] != r[1]).nonzero().flatten()
        print("  mismatches", bad.numel(), bad[:5].tolist(), si[bad[:5]].tolist(), r[0][bad[:5]].tolist(), r[1][bad[:5]].tolist())
pkg.set_option("rrr_sorted", -1)
rv.rank(idx, 1, out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    rv.rank(idx, 1, out); ts.append(pkg.last_kernel_ms())
print(f"automatic rank1: {min(ts):.3f} ms; equals direct: {torch.equal(out, res[(0, 1)])}")
