"""csa[i] on SA-order samples at several densities (csa_wt<..., t_dens, t_inv_dens>, csa_wt.hpp:51-57): rate and resident bytes — the
reference's own lever for the walk length.  usage: sa_density_probe.py [text MiB = 1024] [queries = 2e7]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
nt = mib << 20
text = torch.from_numpy(pkg.english_text(nt, 1234)).cuda()
g = torch.Generator(device="cuda").manual_seed(1007)
idx = torch.randint(0, nt + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
pkg.set_timing(True)
ref = None
for sa_d, isa_d in ((32, 64), (16, 64), (16, 32), (8, 16)):
    csa = pkg.csa_wt(text=text)
    csa.drop_sa(sa_d, isa_d)
    out = csa.sa(idx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        out = csa.sa(idx)
        ts.append(pkg.last_kernel_ms())
    if ref is None:
        ref = out.clone()
    p = csa.footprint_parts()
    print(f"SA / ISA samples at {sa_d} / {isa_d}: csa[i] {nq / min(ts) / 1e3:.0f} Msa/s ({min(ts):.2f} ms per {nq:.0e}); samples {p['sa_isa_samples'] / 1e6:.0f} MB (64-bit entries; packed to 32 bits by "
          f"set_footprint: {p['sa_isa_samples'] / 2e6:.0f} MB); same answers: {bool(torch.equal(out, ref))}", flush=True)
    csa.close()
