"""sd_vector rank/select throughput probe (hand tool for gpurun): m ones over a universe of n."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 40
logm = int(sys.argv[2]) if len(sys.argv) > 2 else 28
nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
n, m = 1 << logn, 1 << logm
pos = torch.unique(torch.randint(0, n, (m,), device=dev, dtype=torch.int64, generator=g))
t0 = time.time(); sd = pkg.sd_vector(positions=pos, n_bits=n); torch.cuda.synchronize()
print(f"n=2^{logn} m={pos.numel()} build {time.time()-t0:.2f}s wl={sd.low_width()} bits/one={sd.device_bytes()*8/pos.numel():.2f}")
idx = torch.randint(0, n + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
out = torch.empty_like(idx)
pkg.set_timing(True)
def run(name, fn, k):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        fn(); ts.append(pkg.last_kernel_ms())
    ms = min(ts); print(f"{name}: {ms:.3f} ms  {k/ms/1e6:.2f} Gq/s")
run("rank1", lambda: sd.rank(idx, 1, out), nq)
i1 = torch.randint(1, pos.numel() + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
run("select1", lambda: sd.select(i1, 1, out), nq)
assert torch.equal(out, pos[i1 - 1])
o8 = torch.empty(nq, dtype=torch.uint8, device=dev)
run("access", lambda: sd.access(idx[: nq // 2], o8[: nq // 2]), nq // 2)
i0 = torch.randint(1, n - pos.numel() + 1, (nq // 100,), device=dev, dtype=torch.int64, generator=g)
run("select0", lambda: sd.select(i0, 0, out[: nq // 100]), nq // 100)
