"""rank_1 on a vector of 2^36 bits (wide slices: bv_sorted.hip, k_sr_rank_lds<MULTI>), bucketed and direct (hand tool for gpurun).
usage: wide_probe.py [log2 bits = 36] [queries = 1e9]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 36
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
n = 1 << ln
g = torch.Generator(device="cuda").manual_seed(3)
words = torch.randint(-2**63, 2**63 - 1, (n // 64,), device="cuda", dtype=torch.int64, generator=g)
bv = pkg.bit_vector(words, n, select1=False, select0=False)
del words
idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
out = torch.empty_like(idx)
pkg.set_timing(True)
res = {}
for mode, name in ((1, "bucketed"), (0, "direct")):
    pkg.set_option("rank_sorted", mode)
    bv.rank(idx, 1, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        bv.rank(idx, 1, out)
        ts.append(pkg.last_kernel_ms())
    res[name] = out[:2_000_000].clone()
    print(f"2^{ln} bits, {nq:.0e} queries, {name}: {min(ts):.2f} ms = {nq / min(ts) / 1e6:.1f} G/s", flush=True)
    if mode == 1:
        pkg.set_option("trace_phases", 1)
        bv.rank(idx, 1, out)
        torch.cuda.synchronize()
        print("   phases:", pkg.last_phases())
        pkg.set_option("trace_phases", 0)
pkg.set_option("rank_sorted", -1)
assert torch.equal(res["bucketed"], res["direct"])
print(f"PROBE_UNITS {8 * nq}")
