"""wt rank / count throughput probe on a synthetic text (hand tool for gpurun)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
rrr = len(sys.argv) > 3 and sys.argv[3] == "rrr"
dev = torch.device("cuda", 0)
nt = mib << 20
text = bench.synthetic_text(nt, 1234, dev)
t0 = time.time(); csa = pkg.csa_wt(text=text, rrr=rrr); print(f"text {mib} MiB rrr={rrr} index build {time.time()-t0:.2f}s sigma={csa.sigma()} wt_bits={csa.wavelet_tree.bv_size()} index_bytes={csa.device_bytes()/2**20:.1f} MiB")
wt = csa.wavelet_tree
g = torch.Generator(device=dev).manual_seed(5)
gi = torch.randint(0, nt + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
gc = text[torch.randint(0, nt, (nq,), device=dev, generator=g)]
out = torch.empty(nq, dtype=torch.int64, device=dev)
pkg.set_timing(True)
def run(name, fn, n):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        fn(); ts.append(pkg.last_kernel_ms())
    ms = min(ts); print(f"{name}: {ms:.3f} ms  {n/ms/1e6:.3f} G/s")
run("wt_rank", lambda: wt.rank(gi, gc, out), nq)
m = 20
st = torch.randint(0, nt - m, (nq,), device=dev, generator=g)
pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
run("fm_count", lambda: csa.count(pats, m, out), nq)
if len(sys.argv) > 4 and sys.argv[4] == "depths":
    for k in (0, 3, 4, 5, 6):
        csa.set_jump_depth(k)
        run(f"fm_count jump depth {k}", lambda: csa.count(pats, m, out), nq)
if len(sys.argv) > 4 and sys.argv[4] == "select":
    occ = torch.bincount(text.long(), minlength=256)
    cs = text[torch.randint(0, nt, (nq,), device=dev, generator=g)]
    ks = (torch.rand(nq, device=dev, generator=g, dtype=torch.float64) * occ[cs.long()].double()).long().clamp_(min=0) + 1
    ks = torch.minimum(ks, occ[cs.long()])
    run("wt_select", lambda: wt.select(ks, cs, out), nq)
    run("wt_inverse_select", lambda: wt.inverse_select(gi.clamp(max=nt)), nq)
    run("wt_access", lambda: wt.access(gi.clamp(max=nt)), nq)
