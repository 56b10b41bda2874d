"""count() throughput probe on the bench text (hand tool for gpurun): the bench's patterns, every k-mer table depth,
suffix array kept / dropped.  SDSL_HIP_FM_FAST=0 in the environment measures the lock-step kernel of fm.hip instead.
usage: fm_probe.py [text MiB = 1024] [patterns = 1e8] [variants = default,none,k7,k8,dropped]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
variants = (sys.argv[3] if len(sys.argv) > 3 else "default,none,k7,k8,dropped").split(",")
dev = torch.device("cuda", 0)
nt = mib << 20
text = torch.from_numpy(pkg.english_text(nt, 1234)).to(dev)
t0 = time.time()
csa = pkg.csa_wt(text=text)
torch.cuda.synchronize()
print(f"text {mib} MiB: index build {time.time() - t0:.2f} s, sigma {csa.sigma()}, index {csa.device_bytes() / 2**20:.0f} MiB, "
      f"jump depth {csa.jump_depth()}, k-mer table depth {csa.kmer_table_depth()} ({csa.kmer_table_bytes() / 2**20:.0f} MiB)", flush=True)
m = 20
st = bench.to_dev(pkg.rnd_positions(15, nq, nt - m, 0), dev)
pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
out = torch.empty(nq, dtype=torch.int64, device=dev)
ref = None
units = 0
pkg.set_timing(True)


def run(name):
    global ref, units
    units += 5 * nq
    csa.count(pats, m, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        csa.count(pats, m, out)
        ts.append(pkg.last_kernel_ms())
    if ref is None:
        ref = out.clone()
        assert bool((ref >= 1).all())
    ok = bool(torch.equal(out, ref))
    print(f"{name}: {min(ts):.3f} / {sorted(ts)[len(ts) // 2]:.3f} / {max(ts):.3f} ms  {nq / min(ts) / 1e3:.0f} Mcount/s  same answers: {ok}  "
          f"index {csa.device_bytes() / 2**20:.0f} MiB (k-mer table k = {csa.kmer_table_depth()}, {csa.kmer_table_bytes() / 2**20:.0f} MiB)",
          flush=True)


for v in variants:
    if v == "default":
        run("default")
    elif v == "none":
        csa.set_kmer_table(0, 0)
        run("no k-mer table")
    elif v.startswith("k"):
        t0 = time.time()
        csa.set_kmer_table(int(v[1:]), 64 << 30)
        torch.cuda.synchronize()
        print(f"  table k <= {v[1:]} built in {time.time() - t0:.2f} s")
        run(f"k-mer table k <= {v[1:]}")
    elif v == "dropped":
        csa.drop_sa()
        run("suffix array and text dropped")
    elif v == "lean":  # 1.5 x the bytes of the reference's csa_wt<wt_huff<>, 32, 64> stream (bench_extras.py: fm_count_lean)
        blob = len(csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL))
        csa.set_footprint(int(1.5 * blob))
        print(f"  footprint {csa.device_bytes()} B = {csa.device_bytes() / blob:.3f} x the SDSL stream ({blob} B): {csa.footprint_parts()}")
        run("lean (1.5 x SDSL's bytes)")
    elif v.startswith("dk"):
        csa.set_kmer_table(int(v[2:]), 64 << 30) if csa.sampling()[2] else None
        run(f"dropped, k = {csa.kmer_table_depth()}")
print(f"PROBE_UNITS {units}")
