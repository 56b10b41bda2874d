"""csa[i] on SDSL's default samples and extract: steady-state rates (hand tool for gpurun)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
pkg = importlib.import_module("sdsl-lite_amd")
dev = torch.device("cuda", 0)
nt = 1 << 30
text = bench.synthetic_text(nt, 1234, dev)
csa = pkg.csa_wt(text=text)
csa.drop_sa()
g = torch.Generator(device=dev).manual_seed(3)
pkg.set_timing(True)
for nq in (2_000_000, 20_000_000, 100_000_000):
    idx = torch.randint(0, nt + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    csa.sa(idx); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        csa.sa(idx); ts.append(pkg.last_kernel_ms())
    print(f"csa[i] dens 32, {nq} queries: {min(ts):.2f} ms  {nq / min(ts) / 1e3:.1f} M/s")
eb = torch.randint(0, nt - 64, (10_000_000,), device=dev, dtype=torch.int64, generator=g)
off, t = csa.extract(eb, eb + 63); torch.cuda.synchronize()
import time
t0 = time.time(); off, t = csa.extract(eb, eb + 63); torch.cuda.synchronize(); dt = time.time() - t0
print(f"extract 64 B x {eb.numel()}: {dt*1e3:.1f} ms  {t.numel()/dt/1e9:.2f} GB/s")
