"""count() with patterns and answers in HOST memory (numpy arrays): PCIe-inclusive rate (hand tool for gpurun)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
pkg = importlib.import_module("sdsl-lite_amd")
dev = torch.device("cuda", 0)
nt = 1 << 30
text = bench.synthetic_text(nt, 1234, dev)
csa = pkg.csa_wt(text=text)
m, n = 20, int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
g = torch.Generator(device=dev).manual_seed(4)
st = torch.randint(0, nt - m, (n,), device=dev, generator=g)
pats_d = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
want = csa.count(pats_d, m)
pats_h = pats_d.cpu().numpy()
out_h = np.empty(n, dtype=np.uint64)
for rep in range(3):
    t0 = time.time(); csa.count(pats_h, m, out_h); dt = time.time() - t0
    print(f"host arrays, {n} patterns: {dt*1e3:.1f} ms  {n/dt/1e6:.1f} Mcount/s  equal to the device-resident answers: {np.array_equal(out_h, want.cpu().numpy().view(np.uint64))}")
