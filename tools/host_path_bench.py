"""Throughput of the HOST-pointer path of the C ABI (what an unmodified SDSL caller with std::vectors sees): pageable
numpy arrays in and out.  Hand tool for gpurun."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("sdsl-lite_amd")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**8
n = 1 << logn
w = pkg.set_random_bits(n, 42)
bv = pkg.bit_vector(w, n)
rng = np.random.default_rng(1)
idx = rng.integers(0, n + 1, nq).astype(np.uint64)
out = np.empty(nq, dtype=np.uint64)
for name, fn in (("rank (host in/out)", lambda: bv.rank(idx, 1, out)),):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{name}: {t*1e3:.1f} ms  {nq/t/1e9:.3f} Gq/s  ({16*nq/t/1e9:.1f} GB/s over PCIe, both directions)")
i1 = rng.integers(1, bv.ones() + 1, nq).astype(np.uint64)
for name, fn in (("select (host in/out)", lambda: bv.select(i1, 1, out)),):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{name}: {t*1e3:.1f} ms  {nq/t/1e9:.3f} Gq/s  ({16*nq/t/1e9:.1f} GB/s over PCIe, both directions)")
# correctness of the pipelined path against the device-resident path
import torch
d = bv.rank(torch.from_numpy(idx.view(np.int64)).cuda(), 1).cpu().numpy().view(np.uint64)
bv.rank(idx, 1, out)
print("pipelined == resident:", bool(np.array_equal(out, d)))
