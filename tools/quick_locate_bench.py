"""SA access / locate / extract throughput probe on a synthetic text (hand tool for gpurun)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
rrr = len(sys.argv) > 3 and sys.argv[3] == "rrr"
dev = torch.device("cuda", 0)
nt = mib << 20
text = bench.synthetic_text(nt, 1234, dev)
t0 = time.time(); csa = pkg.csa_wt(text=text, rrr=rrr); print(f"text {mib} MiB rrr={rrr} build {time.time()-t0:.2f}s")
N = csa.size()
g = torch.Generator(device=dev).manual_seed(5)
idx = torch.randint(0, N, (nq,), device=dev, dtype=torch.int64, generator=g)
pkg.set_timing(True)
def run(name, fn, n, unit="q"):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        t = time.time(); fn(); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    ms = min(ts); print(f"{name}: {ms:.3f} ms  {n/ms/1e6:.4f} G{unit}/s (kernel {pkg.last_kernel_ms():.3f} ms)")
run("sa (whole SA)", lambda: csa.sa(idx), nq)
m = 20
npat = nq // 10
st = torch.randint(0, nt - m, (npat,), device=dev, generator=g)
pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
off, pos = csa.locate(pats, m)
print("locate occurrences:", pos.numel(), "for", npat, "patterns")
run("locate m=20 (whole SA)", lambda: csa.locate(pats, m), npat, "pat")
csa.drop_sa()
print("sampling:", csa.sampling())
small = idx[: nq // 10]
run("sa (dens 32 walk)", lambda: csa.sa(small), small.numel())
run("isa (dens 64 walk)", lambda: csa.isa(small), small.numel())
run("lf", lambda: csa.lf(idx), nq)
run("psi", lambda: csa.psi(idx), nq)
run("locate m=20 (walk)", lambda: csa.locate(pats[: (npat // 10) * m], m), npat // 10, "pat")
b = torch.randint(0, nt - 64, (nq // 10,), device=dev, dtype=torch.int64, generator=g)
e = b + 63
off, txt = csa.extract(b, e)
ok = bool((txt.view(-1, 64)[:1000] == text[(b[:1000].view(-1, 1) + torch.arange(64, device=dev).view(1, 64))]).all())
print("extract correct:", ok)
run("extract 64 B snippets", lambda: csa.extract(b, e), (nq // 10) * 64, "B")
