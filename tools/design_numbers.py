"""The figures DESIGN.md §7 quotes, from a bench.py line: python tools/design_numbers.py profiles/bench_r04_full_line.json"""
import json
import sys

d = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
r, e = d["roofline"], d.get("extras", {})
ks = r["kernel_ms_per_step"]
print(f"headline {d['value']:.2f} Grank/s  {d['ms_per_step']:.3f} ms/step  kernel {ks['min']:.2f}/{ks['median']:.2f}/{ks['max']:.2f}  frac {r['frac']:.3f}  traffic {r.get('traffic')}")
print("phases", {k: round(v, 2) for k, v in r["phases_ms"].items()})
dk = r["direct_kernel"]
print(f"direct {dk['Gq/s']:.1f} G/s {dk['kernel_ms']:.2f} ms frac {dk.get('frac')}")
cb = d["cpu_baseline"]
print(f"cpu {cb['value']:.4f} {cb['unit']} {cb['ns_per_query']:.1f} ns kind {cb['kind']}; all cores {d['cpu_baseline_all_cores']['value']:.3f}")
ee = d.get("end_to_end")
if ee:
    print("e2e pageable", round(ee["pageable"]["Grank/s"], 2), "pinned", round(ee["pinned"]["Grank/s"], 2))
for k, v in e.items():
    if k == "batch_sweep":
        for row in v:
            print("sweep", row["n_bits_log2"], f"{row['queries']:.0e}", {n: (round(row[n]["Grank/s"], 1), row[n].get("route")) if row.get(n) else None for n in ("default", "direct", "bucketed")}, row["same_answers"])
    elif isinstance(v, dict):
        keep = {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if isinstance(b, (int, float, bool, str)) and len(str(b)) < 60}
        print(k, keep)
        for sub in ("roofline", "end_to_end", "cpu_baseline", "direct_kernel"):
            if isinstance(v.get(sub), dict):
                print("   ", sub, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v[sub].items() if isinstance(b, (int, float, bool, str)) and len(str(b)) < 80})
