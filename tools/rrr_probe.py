"""rrr_vector<63> on the configs[2] vector (2^log_n bits, 5 % dense): ONE operation, default dispatch, a few launches — the
command tools/collect_profiles.sh puts under the counters so that every k_sw_* / k_rs_* dispatch of the run belongs to that
operation.  usage: rrr_probe.py rank|select [log_n = 34] [queries = 1e9]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("sdsl-lite_amd")
op = sys.argv[1] if len(sys.argv) > 1 else "rank"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 34
nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**9
dev = torch.device("cuda", 0)
n = 1 << logn
G = bench.golden().get("c3", {})
ckp = os.path.join(bench.ROOT, "tests", "golden", "mt9_checkpoints.bin")
if os.path.exists(ckp) and G and logn == G.get("log_n"):
    w = pkg.density_bits(n, 9, 5, np.fromfile(ckp, dtype=np.uint64).reshape(-1, 313), G["checkpoint_stride"])
else:
    w = pkg.density_bits(n, 9, 5)
rv = pkg.rrr_vector(bench.to_dev(w, dev), n)
del w
if op == "rank":
    arg = bench.to_dev(pkg.rnd_positions(7, nq, n + 1, 0), dev)
else:
    arg = bench.to_dev(pkg.rnd_positions(11, nq, rv.ones(), 1), dev)
out = torch.empty_like(arg)
pkg.set_timing(True)
calls, ts = 5, []
for _ in range(calls):
    (rv.rank if op == "rank" else rv.select)(arg, 1, out)
    ts.append(pkg.last_kernel_ms())
print(f"rrr {op}: {min(ts):.3f} / {sorted(ts)[calls // 2]:.3f} / {max(ts):.3f} ms per 10^{np.log10(nq):.0f} queries, {nq / min(ts) / 1e6:.1f} Gq/s, "
      f"{rv.device_bytes() * 8 / n:.3f} bits/bit")
print(f"PROBE_UNITS {calls * nq}")
