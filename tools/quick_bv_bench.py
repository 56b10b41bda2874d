"""Quick rank/select throughput probe (hand tool for gpurun; bench.py is the contract bench)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 34
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 28
n = 1 << logn
g = torch.Generator(device="cuda").manual_seed(42)
words = torch.randint(-2**63, 2**63 - 1, (n // 64,), device="cuda", dtype=torch.int64, generator=g)
t0 = time.time(); bv = pkg.bit_vector(words, n); torch.cuda.synchronize(); t1 = time.time()
print(f"n=2^{logn} build {t1-t0:.3f}s ones={bv.ones()} device_bytes={bv.device_bytes()/2**30:.3f} GiB")
del words
idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
out = torch.empty_like(idx)
pkg.set_timing(True)
for name, fn, arg in (("rank1", lambda a: bv.rank(a, 1, out), idx), ("rank0", lambda a: bv.rank(a, 0, out), idx)):
    fn(arg); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        fn(arg); ts.append(pkg.last_kernel_ms())
    ms = min(ts)
    print(f"{name}: {ms:.3f} ms  {nq/ms/1e6:.2f} Gq/s  alg {96*nq/ms/1e6:.0f} GB/s = {96*nq/ms/1e6/8000:.3f} of 8 TB/s")
ones = bv.ones()
i1 = torch.randint(1, ones + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
i0 = torch.randint(1, n - ones + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
for name, b, arg in (("select1", 1, i1), ("select0", 0, i0)):
    bv.select(arg, b, out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        bv.select(arg, b, out); ts.append(pkg.last_kernel_ms())
    ms = min(ts)
    print(f"{name}: {ms:.3f} ms  {nq/ms/1e6:.2f} Gq/s  alg {112*nq/ms/1e6:.0f} GB/s = {112*nq/ms/1e6/8000:.3f} of 8 TB/s")
# spot check vs torch on a sample
samp = idx[:1000].cpu()
