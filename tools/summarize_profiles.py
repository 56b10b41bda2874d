#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed profiles/ files:
   profiles/bench_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels + top others)
   profiles/bench_<tag>_pmc.md             per-kernel PMC table, per query
   profiles/pmc_latest.json                HBM bytes per launch for bench.py's roofline.traffic
"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
Q = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name[:80]


rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join(dst, f"bench_{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:14]:
        w.writerow([short(r["Name"])[:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"], r["StdDev"]])

try:
    rows2 = list(csv.DictReader(open(os.path.join(src, "trace_select", "bench_kernel_stats.csv"))))
    with open(os.path.join(dst, f"bench_{tag}_kernel_stats_with_select.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows2[:14]:
            w.writerow([short(r["Name"])[:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
except Exception:
    pass

try:  # the full default bench: every kernel of this library (sdslhip::*), whatever its share
    rows3 = list(csv.DictReader(open(os.path.join(src, "trace_full", "bench_kernel_stats.csv"))))
    with open(os.path.join(dst, f"bench_{tag}_kernel_stats_full.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows3:
            if "sdslhip" in r["Name"]:
                w.writerow([short(r["Name"])[:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
except Exception:
    pass

per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_rd", "pmc_wr", "pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if "sdslhip" in k:
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"# PMC summary of `bench.py --steps 5 --extras select` ({tag}), values per query (launch of {Q:.0e} queries)", "",
         "Kernels launched once per step; counters averaged over the launches that processed the full batch.", ""]
traffic = {}
for k, d in sorted(per.items()):
    if not any(x in k for x in ("k_rank", "k_select")):
        continue
    lines.append(f"## {k}")
    lines.append("| counter | per query |")
    lines.append("|---|---|")
    vals = {}
    for c, v in sorted(d.items()):
        big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v  # drop the tiny validation launches
        vals[c] = sum(big) / len(big) / Q
        lines.append(f"| {c} | {vals[c]:.4f} |")
    if "TCC_EA0_RDREQ_sum" in vals:
        n32 = vals.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        n128 = vals.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        n64 = vals["TCC_EA0_RDREQ_sum"] - n32 - n128
        rd = 32 * n32 + 64 * n64 + 128 * n128
        w64 = vals.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        wr = 64 * w64 + 32 * (vals.get("TCC_EA0_WRREQ_sum", 0.0) - w64)
        lines.append("")
        lines.append(f"HBM-side read bytes/query = 32*{n32:.3f} + 64*{n64:.3f} + 128*{n128:.3f} = **{rd:.1f} B**; "
                     f"write bytes/query = **{wr:.1f} B**; fabric requests/query = "
                     f"{vals['TCC_EA0_RDREQ_sum'] + vals.get('TCC_EA0_WRREQ_sum', 0.0):.3f}")
        key = "k_rank_bytes_per_launch" if "k_rank" in k else "k_select_bytes_per_launch"
        traffic[key] = (rd + wr) * Q
    lines.append("")
open(os.path.join(dst, f"bench_{tag}_pmc.md"), "w").write("\n".join(lines))
traffic["source"] = f"profiles/bench_{tag}_pmc.md (TCC_EA0_RDREQ by request size + TCC_EA0_WRREQ, rocprofv3 --pmc, separate passes)"
json.dump(traffic, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
try:
    open(os.path.join(dst, f"bench_{tag}_line_under_rocprof.json"), "w").write(
        open(os.path.join(src, "bench_line_under_trace.json")).read())
except Exception:
    pass
print("\n".join(lines))
