#!/usr/bin/env python3
"""profiles/pmc_latest.json (what bench.py's roofline blocks quote) and the committed per-kernel tables, from gpurun_out/prof_*/
(tools/collect_profiles.sh).  The json carries the sha256 of the kernel sources it was measured on; bench.py ignores it when
the sources have changed since.

Every figure is a TOTAL over all dispatches of the named kernels in the profiled command divided by the units that command
processed (the probes print PROBE_UNITS; the headline's steps are counted by their k_sr_rank_lds dispatches) — no dispatch is
picked or dropped, except for the direct kernel k_rank, which the default dispatch also enqueues as an empty launch with every
bucketed step: there the dispatches above half the largest count."""
import collections, csv, glob, json, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def short(name):
    m = re.search(r"k_\w+(<[^(]*>)?", name)
    s = m.group(0) if m else name.split("(")[0][:70]
    return s.replace("sdslhip::", "").replace("(anonymous namespace)::", "")


def load(d):
    src = os.path.join(root, "gpurun_out", "prof_" + d)
    if not os.path.isdir(src):
        return None
    P = {"tot": collections.defaultdict(lambda: collections.defaultdict(float)),
         "per": collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float))), "trace": {}, "units": None}
    for sub in ("rd", "wr", "sq", "sq2"):
        for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                P["tot"][k][r["Counter_Name"]] += float(r["Counter_Value"])
                P["per"][k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            P["trace"][short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    try:
        for line in open(os.path.join(src, "stdout.txt")):
            if line.startswith("PROBE_UNITS"):
                P["units"] = float(line.split()[1])
    except OSError:
        pass
    return P


def bytes_of(c):
    rd, r128, r64, r32 = c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_32B_sum"]
    wr, w64 = c["TCC_EA0_WRREQ_sum"], c["TCC_EA0_WRREQ_64B_sum"]
    return r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, rd - r128 - r64 - r32) * 64, w64 * 64 + max(0.0, wr - w64) * 32


def group(P, pred):
    """totals over every kernel `pred` accepts: read bytes, written bytes, read requests, VALU wave-instructions, GUI-active cycles"""
    rb = wb = rq = valu = gui = ms = 0.0
    for k, c in P["tot"].items():
        if pred(k):
            r, w = bytes_of(c)
            rb += r; wb += w; rq += c["TCC_EA0_RDREQ_sum"]; valu += c["SQ_INSTS_VALU"]; gui += c["GRBM_GUI_ACTIVE"]
    for k, (calls, tot_ms) in P["trace"].items():
        if pred(k):
            ms += tot_ms
    return {"read": rb, "written": wb, "requests": rq, "valu": valu, "gui": gui, "ms": ms}


out = {"source": "tools/collect_profiles.sh (tools/prof.sh: rocprofv3 --pmc, separate passes) + tools/pmc_json.py: TCC_EA0_RDREQ (32/64/128 B) and "
                 "TCC_EA0_WRREQ (64 B, else 32 B), totals over all dispatches / units processed"}
sha = subprocess.run([sys.executable, "-c", "import bench; print(bench.kernel_sources_sha())"], cwd=root, capture_output=True, text=True)
out["kernel_sources_sha"] = sha.stdout.strip()
try:  # a partial re-collection keeps what an earlier pass measured on the same sources
    prev = json.load(open(os.path.join(root, "profiles", "pmc_latest.json")))
    if prev.get("kernel_sources_sha") == out["kernel_sources_sha"]:
        out = {**prev, **out}
except (OSError, ValueError):
    pass

B = load("bench")
if B:
    is_pass = lambda k: k.startswith("k_sr_") or k.startswith("k_sw_")
    lds = [k for k in B["per"] if k.startswith("k_sr_rank_lds")]  # (a template since the wide slices: k_sr_rank_lds<false>)
    steps = sum(len(B["per"][k]["TCC_EA0_RDREQ_sum"]) for k in lds) or 1
    g = group(B, is_pass)
    out.update({"rank_bucketed_bytes_per_step": (g["read"] + g["written"]) / steps, "rank_bucketed_read_bytes_per_step": g["read"] / steps,
                "rank_bucketed_write_bytes_per_step": g["written"] / steps, "rank_bucketed_steps_profiled": steps,
                "rank_bucketed_kernel_ms_per_step_under_tracer": g["ms"] / max(1, sum(c for k, (c, _) in B["trace"].items() if k.startswith("k_sr_rank_lds")) or steps)})
    for k in B["per"]:
        if k.startswith("k_rank<"):
            d = B["per"][k]["TCC_EA0_RDREQ_sum"]
            big = [i for i, v in d.items() if v > 0.5 * max(d.values())]
            sel = {c: sum(B["per"][k][c].get(i, 0.0) for i in big) / len(big) for c in B["per"][k] if c.startswith("TCC_EA0_RD")}
            dw = B["per"][k]["TCC_EA0_WRREQ_sum"]
            bigw = [i for i, v in dw.items() if v > 0.5 * max(dw.values())] if dw else []
            for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
                sel[c] = sum(B["per"][k][c].get(i, 0.0) for i in bigw) / len(bigw) if bigw else 0.0
            r, w = bytes_of(collections.defaultdict(float, sel))
            out["k_rank_bytes_per_launch"] = r + w
for variant in ("default", "k8", "dropped", "lean"):
    F = load("fm_" + variant)
    if F and F["units"]:
        g = group(F, lambda k: k.startswith("k_fm_start") or k.startswith("k_fm_count_flat") or k.startswith("k_fm_verify2"))
        u = F["units"]
        out["fm_count_%s_bytes_per_pattern" % variant] = (g["read"] + g["written"]) / u
        out["fm_count_%s_requests_per_pattern" % variant] = g["requests"] / u
        out["fm_count_%s_valu_issue_share" % variant] = g["valu"] / (g["gui"] / 8 * 1024 / 4) if g["gui"] else None
        out["fm_count_%s_kernel_ms_per_1e8_under_tracer" % variant] = g["ms"] / u * 1e8
for op in ("rank", "select"):
    Rr = load("rrr_" + op)
    if Rr and Rr["units"]:
        g = group(Rr, lambda k: k.startswith("k_sr_") or k.startswith("k_sw_") or k.startswith("k_rs_") or k.startswith("k_rrr_"))
        out["rrr_%s_bucketed_bytes_per_query" % op] = (g["read"] + g["written"]) / Rr["units"]
        out["rrr_%s_bucketed_kernel_ms_per_1e9_under_tracer" % op] = g["ms"] / Rr["units"] * 1e9
# the secondary kernels (tools/kernel_probe.py): everything the probe launched except the index build, per unit of PROBE_UNITS
PROBES = {"walk_sa": ("fm_sa", lambda k: k.startswith("k_fm_walk")), "walk_extract": ("fm_extract", lambda k: k.startswith(("k_fm_walk", "k_fm_piece", "k_fm_lengths"))),
          "walk_locate": ("fm_locate", lambda k: k.startswith(("k_fm_walk", "k_fm_expand", "k_fm_lengths"))),
          "rrr_count": ("fm_count_rrr63", lambda k: k.startswith("k_fm_count_rrr") or k.startswith("k_fm_verify") or k.startswith("k_fm_keys")),
          "rrr_count_lean": ("fm_count_rrr63_lean", lambda k: k.startswith("k_fm_count_rrr") or k.startswith("k_fm_verify") or k.startswith("k_fm_keys")),
          "wt_select": ("wt_select", lambda k: k.startswith(("k_wt_sel", "k_sw_", "k_sr_"))),
          "sd_rank": ("sd_rank", lambda k: k.startswith(("k_sd_rank<", "k_sd_rank_lane", "k_sd_redo")) or k in ("k_sd_rank",)),
          "sd_select1": ("sd_select1", lambda k: k.startswith("k_sd_select") and "select0" not in k),
          "sd_select0": ("sd_select0", lambda k: k.startswith(("k_sd_select", "k_sd_redo")))}
for tag_, (key, pred) in PROBES.items():
    Pp = load(tag_)
    if Pp and Pp["units"]:
        g = group(Pp, pred)
        u = Pp["units"]
        out[key + "_bytes_per_unit"] = (g["read"] + g["written"]) / u
        out[key + "_requests_per_unit"] = g["requests"] / u
        out[key + "_valu_issue_share"] = g["valu"] / (g["gui"] / 8 * 1024 / 4) if g["gui"] else None
        out[key + "_kernel_ms_per_1e8_under_tracer"] = g["ms"] / u * 1e8
        out[key + "_kernels"] = sorted(k for k in Pp["trace"] if pred(k))
W = load("wt")
if W:
    for k in W["per"]:
        if k.startswith("k_wt_rank"):
            d = W["per"][k]["TCC_EA0_RDREQ_sum"]
            big = [i for i, v in d.items() if v > 0.5 * max(d.values())]
            sel = collections.defaultdict(float, {c: sum(W["per"][k][c].get(i, 0.0) for i in big) / len(big) for c in W["per"][k] if c.startswith("TCC_EA0_RD")})
            dw = W["per"][k]["TCC_EA0_WRREQ_sum"]
            bigw = [i for i, v in dw.items() if v > 0.5 * max(dw.values())] if dw else []
            for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
                sel[c] = sum(W["per"][k][c].get(i, 0.0) for i in bigw) / len(bigw) if bigw else 0.0
            r, w = bytes_of(sel)
            out["k_wt_rank_bytes_per_query"] = (r + w) / 1e8
json.dump(out, open(os.path.join(root, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
