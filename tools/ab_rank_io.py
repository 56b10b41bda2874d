"""A/B of the nontemporal query-I/O variant of k_rank on ONE allocation (hand tool for gpurun)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
n, nq = 1 << 34, 10**9
g = torch.Generator(device="cuda").manual_seed(42)
words = torch.randint(-2**63, 2**63 - 1, (n // 64,), device="cuda", dtype=torch.int64, generator=g)
bv = pkg.bit_vector(words, n, select1=True, select0=False)
del words
idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
out = torch.empty_like(idx)
pkg.set_timing(True)
for rnd in range(1):
    for v in ("0", "1"):
        os.environ["SDSL_HIP_RANK_IO_NT"] = v
        ts = []
        for _ in range(4):
            bv.rank(idx, 1, out); ts.append(pkg.last_kernel_ms())
        print(f"round {rnd} io_nt={v}: min {min(ts):.3f} ms  mean {sum(ts)/len(ts):.3f} ms  {nq/min(ts)/1e6:.2f} Gq/s")

i1 = torch.randint(1, bv.ones() + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
for rnd in range(3):
    for v in ("rq", "wq"):
        os.environ["SDSL_HIP_SELECT_VARIANT"] = v
        ts = []
        for _ in range(4):
            bv.select(i1, 1, out); ts.append(pkg.last_kernel_ms())
        print(f"select round {rnd} variant={v}: min {min(ts):.3f} ms  {nq/min(ts)/1e6:.2f} Gq/s")
