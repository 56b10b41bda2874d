"""csa_wt<wt_huff<rrr_vector<63>>> on the bench text (hand tool for gpurun): what the device holds against the real library's stream of the
same type (our serialiser writes its bytes: tests/test_gpu_parity.py), and count() with the suffix array kept / dropped / at a footprint.
usage: fm_rrr_probe.py [text MiB = 1024] [patterns = 2e7] [variants = default,dropped,lean]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
variants = (sys.argv[3] if len(sys.argv) > 3 else "default,dropped,lean").split(",")
dev = torch.device("cuda", 0)
nt = mib << 20
text = torch.from_numpy(pkg.english_text(nt, 1234)).to(dev)
t0 = time.time()
csa = pkg.csa_wt(text=text, rrr=True)
torch.cuda.synchronize()
print(f"text {mib} MiB: index build {time.time() - t0:.2f} s, sigma {csa.sigma()}, index {csa.device_bytes()} B: {csa.footprint_parts()}", flush=True)
blob = len(csa.serialize(32, 64, pkg.capi.LAYOUT_RRR63))
print(f"csa_wt<wt_huff<rrr_vector<63>>, 32, 64> stream: {blob} B = {blob / nt:.4f} B/symbol", flush=True)
m = 20
st = bench.to_dev(pkg.rnd_positions(15, nq, nt - m, 0), dev)
pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
out = torch.empty(nq, dtype=torch.int64, device=dev)
ref = None
units = 0
pkg.set_timing(True)


def run(name):
    global ref, units
    units += 4 * nq
    csa.count(pats, m, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        csa.count(pats, m, out)
        ts.append(pkg.last_kernel_ms())
    if ref is None:
        ref = out.clone()
        assert bool((ref >= 1).all())
    ok = bool(torch.equal(out, ref))
    print(f"{name}: {min(ts):.3f} / {sorted(ts)[len(ts) // 2]:.3f} / {max(ts):.3f} ms  {nq / min(ts) / 1e3:.0f} Mcount/s  same answers: {ok}  "
          f"index {csa.device_bytes()} B = {csa.device_bytes() / blob:.3f} x the stream (k-mer table k = {csa.kmer_table_depth()}, {csa.kmer_table_bytes() / 2**20:.0f} MiB)",
          flush=True)


for v in variants:
    if v == "default":
        run("default")
    elif v == "dropped":
        csa.drop_sa()
        print(f"  {csa.footprint_parts()}")
        run("suffix array and text dropped")
    elif v.startswith("lean"):
        x = float(v[4:]) if len(v) > 4 else 1.5
        csa.set_footprint(int(x * blob))
        print(f"  footprint {csa.device_bytes()} B = {csa.device_bytes() / blob:.3f} x the stream: {csa.footprint_parts()}")
        run(f"lean ({x} x the stream)")
    elif v.startswith("dk"):
        csa.set_kmer_table(int(v[2:]), 64 << 30)
        run(f"k-mer table k = {csa.kmer_table_depth()}")
print(f"PROBE_UNITS {units}")
