"""One operation of the secondary kernels per run, for the profiler (tools/prof.sh) and as a quick rate check: the bench's text and
sizes, PROBE_UNITS = what tools/pmc_json.py divides the counters by.
usage: kernel_probe.py <sa|extract|locate|rrr_count|rrr_count_lean|sd_rank|sd_select0|sd_select1|wt_select> [text MiB = 1024] [queries]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
what = sys.argv[1]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1007)
pkg.set_timing(True)
REPS = 4


def run(name, fn, units, unit_name):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        fn(); ts.append(pkg.last_kernel_ms())
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name}: {min(ts):.3f} / {ms:.3f} / {max(ts):.3f} ms  {units / ms / 1e6:.3f} G{unit_name}/s", flush=True)
    print(f"PROBE_UNITS {units * (REPS + 1)}")


if what in ("sa", "extract", "locate", "rrr_count", "rrr_count_lean", "wt_select"):
    nt = mib << 20
    text = torch.from_numpy(pkg.english_text(nt, 1234)).to(dev)
if what in ("sa", "extract", "locate"):
    csa = pkg.csa_wt(text=text)
    if what == "locate":
        m, npat = 20, int(float(sys.argv[3])) if len(sys.argv) > 3 else 10_000_000
        st = bench.to_dev(pkg.rnd_positions(15, npat, nt - m, 0), dev)
        pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
        lq, rq = csa.interval(pats, m)
    csa.drop_sa()
    if what == "sa":
        nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000_000
        idx = torch.randint(0, nt + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
        run("csa[i], SA samples every 32nd suffix", lambda: csa.sa(idx), nq, "sa")
    elif what == "extract":
        nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10_000_000
        eb = torch.randint(0, nt - 64, (nq,), device=dev, dtype=torch.int64, generator=g)
        ee = eb + 63
        run("extract, 64-byte snippets, ISA samples every 64th position", lambda: csa.extract(eb, ee), nq * 64, "B")
    else:
        off, pos = csa.sa_range(lq, rq)
        run("locate (sa_range of %d intervals, %d occurrences)" % (npat, pos.numel()), lambda: csa.sa_range(lq, rq), pos.numel(), "occ")
elif what in ("rrr_count", "rrr_count_lean"):
    crrr = pkg.csa_wt(text=text, rrr=True)
    if what == "rrr_count_lean":  # 1.5 x the bytes of the real library's csa_wt<wt_huff<rrr_vector<63>>, 32, 64> stream (bench_extras.py: fm_count_rrr63_lean)
        crrr.set_footprint(int(1.5 * len(crrr.serialize(32, 64, pkg.capi.LAYOUT_RRR63))))
        print(f"  footprint {crrr.device_bytes()} B: {crrr.footprint_parts()}")
    m, npat = 20, int(float(sys.argv[3])) if len(sys.argv) > 3 else 20_000_000
    st = bench.to_dev(pkg.rnd_positions(15, npat, nt - m, 0), dev)
    pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
    out = torch.empty(npat, dtype=torch.int64, device=dev)
    run("count on csa_wt<wt_huff<rrr_vector<63>>>", lambda: crrr.count(pats, m, out), npat, "count")
elif what == "wt_select":
    wt = pkg.wt_huff(text=text)
    nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000_000
    gc = text[bench.to_dev(pkg.rnd_positions(14, nq, nt, 0), dev)]
    occ_c = torch.bincount(text, minlength=256)[gc.long()]
    ks = 1 + bench.to_dev(pkg.rnd_positions(16, nq, 1 << 62, 0), dev) % occ_c
    out = torch.empty(nq, dtype=torch.int64, device=dev)
    run("wt.select", lambda: wt.select(ks, gc, out), nq, "q")
elif what.startswith("sd_"):
    N_sd = 1 << 40
    pos = torch.unique(torch.randint(0, N_sd, (1 << 28,), device=dev, dtype=torch.int64, generator=g))
    sd = pkg.sd_vector(positions=pos, n_bits=N_sd)
    nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000_000
    out = torch.empty(nq, dtype=torch.int64, device=dev)
    if what == "sd_rank":
        xi = torch.randint(0, N_sd + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
        run("sd_vector rank_1", lambda: sd.rank(xi, 1, out), nq, "q")
    elif what == "sd_select1":
        si = torch.randint(1, pos.numel() + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
        run("sd_vector select_1", lambda: sd.select(si, 1, out), nq, "q")
    else:
        zi = torch.randint(1, N_sd - pos.numel() + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
        run("sd_vector select_0", lambda: sd.select(zi, 0, out), nq, "q")
else:
    sys.exit("unknown operation " + what)
