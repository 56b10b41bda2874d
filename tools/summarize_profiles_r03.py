#!/usr/bin/env python3
"""Condenses gpurun_out/prof_r03/ (tools/collect_profiles_r03.sh) into the committed profiles/ files:
   profiles/bench_r03_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the headline command (this library's kernels)
   profiles/bench_r03_kernel_stats_full.csv   the same for the full default bench
   profiles/bench_r03_pmc.md                  per-kernel PMC table of one step (fabric requests, bytes, SQ counters)
   profiles/pmc_latest.json                   HBM bytes per step for bench.py's roofline.traffic, stamped with the sha256 of
                                              the kernel sources it was measured on
"""
import collections, csv, glob, hashlib, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r03")
dst = os.path.join(root, "profiles")
STEPS, WARM = 5, 1          # of the profiled command
LAUNCHES = STEPS + WARM + max(3, min(STEPS, 20))  # + the traced calls bench.py makes after the timed loop (phase spread)


def is_pass(k):  # a kernel of the bucketed step
    return k.startswith("k_sr_") or k.startswith("k_sw_")


def short(name):
    m = re.search(r"k_\w+(<[^>(]*>)?", name)
    return m.group(0) if m else name.split("(")[0][:70]


def stats(sub, out):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return {}
    rows = list(csv.DictReader(open(f[0])))
    d = {}
    with open(os.path.join(dst, out), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if "sdslhip" in r["Name"]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"], r["StdDev"]])
                d[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
    return d


head = stats("trace", "bench_r03_kernel_stats.csv")
# the direct kernels are also enqueued with every default-mode step, where they return at once: their average over ALL
# dispatches says nothing — take the dispatches that did the work from the per-dispatch trace
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k.startswith("k_rank<") or k.startswith("k_select_wq"):
            dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for k, v in dur.items():
        big = [x for x in v if x > 0.5 * max(v)]
        head[k] = (len(big), sum(big) / len(big))
stats("trace_full", "bench_r03_kernel_stats_full.csv")
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
        vals[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])


def per_launch(k, c):
    if k.startswith("k_rank<") or k.startswith("k_select_wq"):
        # the default dispatch also enqueues the direct kernel with every bucketed step, where it returns at once:
        # only the dispatches that did the work count
        d = vals[k][c]
        if not d:
            return 0.0
        top = max(d.values())
        big = [v for v in d.values() if v > 0.5 * top]
        return sum(big) / len(big) if big else 0.0
    n = len(calls[k][c])
    return acc[k][c] / n if n else 0.0


def bytes_of(k):
    rd, r128, r64, r32 = (per_launch(k, "TCC_EA0_RDREQ_sum"), per_launch(k, "TCC_EA0_RDREQ_128B_sum"),
                          per_launch(k, "TCC_EA0_RDREQ_64B_sum"), per_launch(k, "TCC_EA0_RDREQ_32B_sum"))
    wr, w64 = per_launch(k, "TCC_EA0_WRREQ_sum"), per_launch(k, "TCC_EA0_WRREQ_64B_sum")
    rb = r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, rd - r128 - r64 - r32) * 64
    wb = w64 * 64 + max(0.0, wr - w64) * 32
    return rb, wb, rd, wr


sr = [k for k in acc if is_pass(k)]
lines = ["# PMC per kernel, one launch (rocprofv3 --pmc, separate passes; tools/collect_profiles_r03.sh)", "",
         "| kernel | launches/step | avg ms (kernel trace) | fabric read req | read MB | fabric write req | write MB | VALU wave-instr | LDS wave-instr | LDS bank-conflict / active cycles |",
         "|---|---|---|---|---|---|---|---|---|---|"]
tot_r = tot_w = tot_ms = 0.0
for k in sorted(sr, key=lambda k: -head.get(k, (0, 0))[1]) + [k for k in acc if k.startswith("k_rank")]:
    rb, wb, rd, wr = bytes_of(k)
    n_calls, avg = head.get(k, (0, 0.0))
    per_step = n_calls / LAUNCHES if is_pass(k) and LAUNCHES else 1
    if is_pass(k):
        tot_r += rb * per_step
        tot_w += wb * per_step
        tot_ms += avg / 1e6 * per_step
    bc, ia = per_launch(k, "SQ_LDS_BANK_CONFLICT"), per_launch(k, "SQ_LDS_IDX_ACTIVE")
    lines.append(f"| {k} | {per_step:g} | {avg / 1e6:.3f} | {rd:.4g} | {rb / 1e6:.1f} | {wr:.4g} | {wb / 1e6:.1f} | "
                 f"{per_launch(k, 'SQ_INSTS_VALU'):.3g} | {per_launch(k, 'SQ_INSTS_LDS'):.3g} | {bc:.3g} / {ia:.3g} |")
lines += ["", f"bucketed rank, one step (10^9 queries): {tot_r / 1e9:.2f} GB read + {tot_w / 1e9:.2f} GB written = "
              f"{(tot_r + tot_w) / 1e9:.2f} GB of fabric traffic in {tot_ms:.3f} ms of kernel time (sum of the per-kernel averages) "
              f"= {(tot_r + tot_w) / tot_ms / 1e9 if tot_ms else 0:.2f} TB/s; algorithmic bytes (SURVEY 8(d)): 96 GB."]
try:
    line = json.load(open(os.path.join(src, "bench_line_under_trace.json")))
    lines.append(f"bench.py under the tracer: kernel_ms {line['roofline']['kernel_ms']:.3f} (HIP events), value {line['value']:.2f} Grank/s, "
                 f"direct kernel {line['roofline']['direct_kernel']['kernel_ms']:.3f} ms.")
except Exception as e:
    lines.append(f"(bench line under the tracer not available: {e})")
open(os.path.join(dst, "bench_r03_pmc.md"), "w").write("\n".join(lines) + "\n")
h = hashlib.sha256()
for f in ("bv.hip", "bv_device.hpp", "bv_sorted.hip", "bv_sorted_dev.hpp", "bv_swc.hip", "bits.hpp", "wt.hip", "wt_device.hpp", "fm.hip",
          "fm_device.hpp"):
    h.update(open(os.path.join(root, "sdsl-lite_amd", "csrc", f), "rb").read())
kr = [k for k in acc if k.startswith("k_rank<")]
# fused-layout kernels of the wt + fm extras (two extra PMC passes: pmcfull_rd, pmcfull_wr).  The extras launch each
# kernel on the English-class index first (three timed launches of 10^8 queries), later on other indexes / small batches:
# only the first three dispatches of a kernel are averaged.
full = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(os.path.join(src, "pmcfull_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        full[short(row["Kernel_Name"])][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])


def full_bytes(k):
    def pl(c):
        # (the dispatches that did the work: a default-mode batch also enqueues the direct kernel, which returns at once, and
        # small side batches — the k-mer table, bucket plans — run the same kernels; of the full-size ones the first three)
        d = full[k][c]
        ref = full[k]["TCC_EA0_RDREQ_sum"]
        top = max(ref.values()) if ref else 0.0
        ids = [i for i in sorted(d) if ref.get(i, 0.0) > 0.5 * top][:3] if c.startswith("TCC_EA0_RD") else None
        if ids is None:  # the write pass has its own dispatch numbering: judge by its own counter
            wref = full[k]["TCC_EA0_WRREQ_sum"]
            wtop = max(wref.values()) if wref else 0.0
            ids = [i for i in sorted(d) if wref.get(i, 0.0) > 0.5 * wtop][:3]
        return sum(d[i] for i in ids) / len(ids) if ids else 0.0
    rd, r128, r64, r32 = pl("TCC_EA0_RDREQ_sum"), pl("TCC_EA0_RDREQ_128B_sum"), pl("TCC_EA0_RDREQ_64B_sum"), pl("TCC_EA0_RDREQ_32B_sum")
    wr, w64 = pl("TCC_EA0_WRREQ_sum"), pl("TCC_EA0_WRREQ_64B_sum")
    return r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, rd - r128 - r64 - r32) * 64 + w64 * 64 + max(0.0, wr - w64) * 32


fused = {}
for k, per in (("k_fm_count", 1e8), ("k_wt_rank", 1e8), ("k_rrr_rank", 1e9), ("k_rrr_select_flat", 1e9)):
    cand = sorted((x for x in full if x == k or x.startswith(k + "<")), key=full_bytes, reverse=True)
    if cand and full_bytes(cand[0]) > 0:  # (the instantiation with the most traffic: the others serve small side batches)
        fused[k] = full_bytes(cand[0]) / per
        lines.append(f"{cand[0]}: {full_bytes(cand[0]) / 1e9:.2f} GB of fabric traffic per launch = {fused[k]:.1f} B per query")
open(os.path.join(dst, "bench_r03_pmc.md"), "w").write("\n".join(lines) + "\n")
out = {"kernel_sources_sha": h.hexdigest()[:16], "rank_bucketed_bytes_per_step": tot_r + tot_w,
       "rank_bucketed_read_bytes_per_step": tot_r, "rank_bucketed_write_bytes_per_step": tot_w,
       "k_rank_bytes_per_launch": sum(bytes_of(kr[0])[:2]) if kr else None,
       "k_fm_count_bytes_per_pattern": fused.get("k_fm_count"), "k_wt_rank_bytes_per_query": fused.get("k_wt_rank"),
       "k_rrr_rank_bytes_per_query": fused.get("k_rrr_rank"), "k_rrr_select_bytes_per_query": fused.get("k_rrr_select_flat"),
       "source": "tools/collect_profiles_r03.sh: TCC_EA0_RDREQ (32/64/128 B) and TCC_EA0_WRREQ (64 B, else 32 B) per kernel launch"}
json.dump(out, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
print("\n".join(lines[-3:]))
print(json.dumps(out))
