#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/prof.sh) into one table per kernel: launches, average duration (kernel trace), and per
launch the fabric read / write requests and bytes, VALU / SALU / LDS wave-instructions, the VALU's share of the issue cycles and
the LDS bank-conflict share of the LDS-active cycles.

Bytes: a read request is 128, 64 or 32 bytes as the *_128B / _64B / _32B counters say (what is in none of them counts 64);
a write request 64 bytes if in WRREQ_64B, else 32 (MI355X_MICROARCH.md: HBM section — every L2 miss of these kernels is a
128-byte read; rocprofv3's FETCH_SIZE would tally it at 64).  Every dispatch of the run is counted: `launches` is the number of
dispatches and the per-launch figures are totals / launches — no dispatch is selected or dropped, so two collections of the same
command agree by construction.  --units N adds "per unit" columns (N = units one launch processes).

usage: prof_summary.py <tag> [--match substr[,substr...]] [--units N] [--md out.md] [--json out.json]"""
import argparse, collections, csv, glob, json, os, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--match", default="")
ap.add_argument("--units", type=float, default=0.0)
ap.add_argument("--md", default=None)
ap.add_argument("--json", default=None)
ap.add_argument("--min-ms", type=float, default=0.0, help="drop kernels whose average launch is shorter")
ap.add_argument("--stats-csv", default=None, help="rewrite the tracer's kernel_stats.csv with short kernel names, ALL eight columns of every "
                "row kept (round 4 cut the lines at 400 characters and lost the figures of the kernels with the longest signatures)")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + a.tag)


def short(name):
    m = re.search(r"k_\w+(<[^(]*>)?", name)
    s = m.group(0) if m else name.split("(")[0][:70]
    return s.replace("sdslhip::", "").replace("(anonymous namespace)::", "")


if a.stats_csv:
    with open(a.stats_csv, "w", newline="") as fo:
        wr = None
        for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if wr is None:
                    wr = csv.DictWriter(fo, fieldnames=list(r.keys()), quoting=csv.QUOTE_NONNUMERIC)
                    wr.writeheader()
                if "sdslhip" in r["Name"] or "rocprim" in r["Name"]:
                    r["Name"] = short(r["Name"]) if "sdslhip" in r["Name"] else r["Name"].split("(")[0][:120]
                    wr.writerow(r)

want = [w for w in a.match.split(",") if w]
trace = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        trace[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for sub in ("rd", "wr", "sq", "sq2"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])


def pl(k, c):
    n = len(disp[k][c])
    return tot[k][c] / n if n else 0.0


rows = []
for k in sorted(set(trace) | set(tot), key=lambda k: -(trace.get(k, (0, 0, 0, 0))[0] * trace.get(k, (0, 0, 0, 0))[1])):
    if want and not any(w in k for w in want):
        continue
    calls, avg, mn, mx = trace.get(k, (0, 0.0, 0.0, 0.0))
    if avg < a.min_ms:
        continue
    rd, r128, r64, r32 = pl(k, "TCC_EA0_RDREQ_sum"), pl(k, "TCC_EA0_RDREQ_128B_sum"), pl(k, "TCC_EA0_RDREQ_64B_sum"), pl(k, "TCC_EA0_RDREQ_32B_sum")
    wr, w64 = pl(k, "TCC_EA0_WRREQ_sum"), pl(k, "TCC_EA0_WRREQ_64B_sum")
    rb = r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, rd - r128 - r64 - r32) * 64
    wb = w64 * 64 + max(0.0, wr - w64) * 32
    valu, salu, lds = pl(k, "SQ_INSTS_VALU"), pl(k, "SQ_INSTS_SALU"), pl(k, "SQ_INSTS_LDS")
    gui = pl(k, "GRBM_GUI_ACTIVE")  # summed over the 8 XCDs: / 8 = cycles of the launch
    # 256 CUs x 4 SIMDs issue one VALU wave-instruction per 4 cycles each: share of that capacity the kernel used
    valu_share = valu / (gui / 8 * 1024 / 4) if gui else 0.0
    bc, ia = pl(k, "SQ_LDS_BANK_CONFLICT"), pl(k, "SQ_LDS_IDX_ACTIVE")
    wave_cyc = pl(k, "SQ_WAVE_CYCLES")
    rows.append({"kernel": k, "launches": calls or len(disp[k].get("TCC_EA0_RDREQ_sum", ())), "avg_ms": avg, "min_ms": mn, "max_ms": mx,
                 "read_req": rd, "read_bytes": rb, "write_req": wr, "write_bytes": wb, "valu": valu, "salu": salu, "lds": lds,
                 "vmem_rd": pl(k, "SQ_INSTS_VMEM_RD"), "waves": pl(k, "SQ_WAVES"), "valu_issue_share": valu_share,
                 "lds_conflict_over_active": bc / ia if ia else 0.0,
                 "wait_any_share": pl(k, "SQ_WAIT_ANY") / wave_cyc if wave_cyc else 0.0,
                 "tbps": (rb + wb) / (avg * 1e-3) / 1e12 if avg else 0.0})
hdr = "| kernel | launches | avg ms (min / max) | read req | read MB | write req | write MB | fabric TB/s | VALU | SALU | LDS | VALU share of issue | LDS conflict / active |"
lines = [hdr, "|" + "---|" * 13]
for r in rows:
    lines.append(f"| {r['kernel']} | {r['launches']} | {r['avg_ms']:.3f} ({r['min_ms']:.3f} / {r['max_ms']:.3f}) | {r['read_req']:.4g} | {r['read_bytes'] / 1e6:.1f} | "
                 f"{r['write_req']:.4g} | {r['write_bytes'] / 1e6:.1f} | {r['tbps']:.2f} | {r['valu']:.3g} | {r['salu']:.3g} | {r['lds']:.3g} | "
                 f"{r['valu_issue_share']:.2f} | {r['lds_conflict_over_active']:.2f} |")
if a.units:
    lines += ["", f"per unit ({a.units:g} units per launch):", "", "| kernel | read req | bytes (read + written) | VALU wave-instr x 64 lanes |", "|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r['kernel']} | {r['read_req'] / a.units:.3f} | {(r['read_bytes'] + r['write_bytes']) / a.units:.1f} | {r['valu'] * 64 / a.units:.1f} |")
text = "\n".join(lines) + "\n"
print(text)
if a.md:
    open(a.md, "w").write(text)
if a.json:
    json.dump(rows, open(a.json, "w"), indent=1)
