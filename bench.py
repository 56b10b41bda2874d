#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X rank/select engine (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: `--queries` (default 10^9) batched rank_1
queries on a 2^34-bit random bit vector resident in HBM together with the query and result arrays.
Inputs are SURVEY.md 8(d)'s streams (vector words = std::mt19937_64(42), positions = mt19937_64(7 + rank) % (n + 1),
generated on the host by the library's integer-only generators), so that the first 10^7 answers can be compared with
the digests the REAL sdsl-lite produced for them in the build container (tests/golden/golden_large.json).  N > 1: one process per GPU (torchrun / torch.distributed, backend nccl =
RCCL); the index is replicated, every rank answers its own resident shard of the same size (weak
scaling, no data-path collective); the step time is the MAX over ranks and `value` is the whole-job
aggregate.  Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (dominant kernel
against the HBM roofline, algorithmic bytes per SURVEY.md §8(d)) and `cpu_baseline` (the reference's CPU
path on a bounded sample of the same workload, timed on this box's host cores, N=1 only).

The line is COMPACT (< 4 KiB, asserted by tests/test_bench_line.py) and is the LAST line of stdout.  The secondary measurements
(`--extras e2e,sweep,select,rrr,sd,shapes,wt,fm,...`: bench_extras.py) never enter the timed region and never enter the line: they go
to the sidecar `bench_extras.json` next to this script (path in the line's `extras_file`, content echoed on stderr); the line carries
only `secondary` (the metric's second half: Mcount/s of count()), `end_to_end` and a few headline figures of the other legs.  A
watchdog prints the line and leaves if the legs outlive `--extras-budget-s`: a hung leg cannot cost the headline.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bench_common import (ALG_BYTES, HBM_PEAK_GBS, box_facts, digest_matches, digests_match, golden, host_cpu_limits, kernel_sources_sha,  # noqa: E402
                          pmc_traffic, set_mempolicy_interleave, spread_of, time_steps)
from bench_common import cpu_time, synthetic_text, to_dev  # noqa: E402,F401  (the probes under tools/ reach them through this module)

LINE_LIMIT = 4096  # bytes; the driver keeps a bounded tail of stdout (round 4: a 23 KB line came back unparsed)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--log-n", type=int, default=34, help="bit vector length = 2^log_n (BASELINE: 34)")
    p.add_argument("--queries", type=float, default=1e9, help="queries per step per GPU (BASELINE: 1e9)")
    p.add_argument("--extras", type=str, default=None,
                   help="comma list of: e2e,sweep,select,rrr,sd,shapes,wt,fm,big,fm_sharded,group (or 'none'); big = an index of 2^32 + 777 symbols "
                        "(opt-in: 172 GB of working memory); default: the first eight on one "
                        "GPU, fm_sharded (configs[4]: 10^8 patterns sharded over the ranks) on several; group = rank 0 also drives all "
                        "GPUs through the C ABI's device group (RCCL point-to-point between devices: opt-in, it has not run on "
                        "hardware with more than one device yet and must not endanger the scaling run)")
    p.add_argument("--text-mib", type=int, default=1024, help="synthetic text size for the wt/fm extras (BASELINE: 1 GiB)")
    p.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU time budget per cpu_baseline sample")
    p.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only "
                   "for exercising the multi-rank control flow on a single GPU)")
    p.add_argument("--mode", type=str, default="ranks", choices=["ranks", "group"],
                   help="ranks: one process per GPU (torch.distributed; --gpus N > 1 without WORLD_SIZE launches the ranks itself); "
                        "group: ONE process drives all N GPUs through the C ABI's device group (sdsl_hip_group_*: RCCL broadcast at "
                        "load time, scatter / kernels / gather per batch)")
    p.add_argument("--extras-budget-s", type=float, default=None, help="wall-clock budget of all the legs together (default: 1500 s on one GPU, "
                   "600 s on several: a collective that never completes must not hold the scaling run); when it runs out the line is printed "
                   "with what is finished and the process leaves")
    p.add_argument("--sidecar", type=str, default=None, help="where the legs' blocks go (default: bench_extras.json next to bench.py)")
    p.add_argument("--text-file", type=str, default=None, help="a text for the wt / fm extras instead of the synthetic stand-in "
                   "(e.g. Pizza&Chili english.1GB; the first --text-mib MiB are used, zero bytes are dropped)")
    return p.parse_args()


def self_launch(a):
    """--gpus N > 1 without a launcher: start the N ranks ourselves (torch.distributed.run on 127.0.0.1).  Never returns."""
    import socket
    if a.backend == "nccl" and torch.cuda.device_count() < a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} device(s) are visible\n")
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)



def pin_to_gpu_numa_node(local):
    """One rank per GPU: keep the rank's host threads (the launch thread, the generators' workers) on the NUMA node the GPU hangs off —
    on a two-socket 8-GPU node a rank scheduled on the far socket pays the inter-socket hop on every launch and every host-array copy.
    Best effort: sysfs says which node (pci_bus_id of the device -> /sys/bus/pci/devices/<id>/numa_node -> its cpulist); where any of
    it is missing (containers, single-socket hosts report -1) nothing changes.  Returns what was done, for the line's config."""
    try:
        p = torch.cuda.get_device_properties(local)
        bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"numa_node": node, "cpus": len(allowed)}
    except Exception:
        return None


def cpu_baseline(pkg, words_dev, n_bits, idx_dev, gpu_out_dev, seconds):
    """The reference's CPU path on this host: real sdsl-lite (oracle/_ref, kind 'reference') if the
    prebuilt library travelled with the repo, else the C restatement (kind 'port').  One thread, scalar
    loop over distinct queries like the reference's harnesses; results are compared with the GPU's."""
    import oracle_lib as ol
    words = words_dev.cpu().numpy().view(np.uint64)
    kind = "reference" if ol.have_ref() else "port"
    t0 = time.perf_counter()
    if kind == "reference":
        words_p = ol.padded(words, n_bits)
        h = ol.ref().L.ref_bv_create(words_p.ctypes.data, n_bits)

        def run(idx):
            out = np.empty(idx.size, dtype=np.uint64)
            ol.ref().L.ref_bv_rank(h, 1, idx.ctypes.data, idx.size, out.ctypes.data)
            return out
    else:
        words_p = ol.padded(words, n_bits)
        h = ol.oracle().L.orc_rank_v5_build(words_p.ctypes.data, n_bits, 1)

        def run(idx):
            out = np.empty(idx.size, dtype=np.uint64)
            ol.oracle().L.orc_rank_v5_batch(h, idx.ctypes.data, idx.size, out.ctypes.data)
            return out
    build_s = time.perf_counter() - t0
    probe = idx_dev[:1_000_000].cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    run(probe)
    per_q = (time.perf_counter() - t0) / probe.size
    n_s = int(min(idx_dev.numel(), max(1_000_000, seconds / per_q)))
    sample = idx_dev[:n_s].cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    res = run(sample)
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(res, gpu_out_dev[:n_s].cpu().numpy().view(np.uint64)))
    one = {
        "value": n_s / dt / 1e9, "unit": "Grank/s", "cores": 1, "kind": kind,
        "sample": f"first {n_s} of the step's queries, scalar loop, 1 thread; "
                  f"rank_support_v5 on the same 2^{int(np.log2(n_bits))}-bit vector (build {build_s:.1f}s)",
        "ns_per_query": dt / n_s * 1e9, "matches_gpu": same,
    }
    # the same loop on every host core: threads pinned one per logical CPU of the affinity mask, output pre-faulted,
    # released together, >= 10^7 queries per thread (several passes over its slice if the step has fewer), clock stopped
    # when the slowest thread is done — thread creation and page faults are outside the measurement.  What the box really
    # grants is reported beside it: the cgroup's CPU quota (a container may see 256 CPUs and be allowed a few of them) and
    # the NUMA placement of the index (interleaved over all nodes for this leg, so that no socket reads it remotely)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    limits = host_cpu_limits()
    n_m = int(min(idx_dev.numel(), threads * 10_000_000))
    reps = max(1, int(np.ceil(10_000_000 / max(1, n_m // threads))))
    sample = idx_dev[:n_m].cpu().numpy().view(np.uint64)
    outm = np.zeros(n_m, dtype=np.uint64)
    if kind == "reference":
        interleaved = set_mempolicy_interleave(True)
        h2 = ol.ref().L.ref_bv_create(words_p.ctypes.data, n_bits) if interleaved else h
        set_mempolicy_interleave(False)
        dtm = ol.ref().L.ref_bv_rank_mt_timed(h2, 1, sample.ctypes.data, n_m, outm.ctypes.data, threads, reps)
        if h2 is not h:
            ol.ref().L.ref_bv_destroy(h2)
    else:
        interleaved = False
        t0 = time.perf_counter()
        for _ in range(reps):
            ol.oracle().L.orc_rank_v5_batch_mt(h, sample.ctypes.data, n_m, outm.ctypes.data, threads)
        dtm = time.perf_counter() - t0
    model, smt = "unknown", "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = f"{len(sib.replace('-', ',').split(','))} hardware threads per core (cpu0 siblings: {sib})"
    except OSError:
        pass
    eff = limits.get("effective_cpus")
    allc = {"value": n_m * reps / dtm / 1e9, "unit": "Grank/s", "cores": threads, "kind": kind, "cpu_model": model,
            "smt": smt, "pinned": True, "queries_per_thread": n_m * reps // threads,
            "ns_per_query_per_thread": dtm / (n_m * reps / threads) * 1e9,
            "host_limits": limits, "index_interleaved_over_numa_nodes": interleaved,
            "note": (f"{threads} threads were started (one per CPU of the affinity mask) but the cgroup grants {eff:.1f} CPUs' worth of time: "
                     "the figure is what this container may use, not what the machine can do") if eff and eff < 0.9 * threads else None,
            "sample": f"first {n_m} of the step's queries, contiguous slices, {threads} pinned threads, {reps} pass(es), "
                      f"output pre-faulted, timed from a common start to the slowest thread",
            "matches_gpu": bool(np.array_equal(outm, gpu_out_dev[:n_m].cpu().numpy().view(np.uint64)))}
    return one, allc



def sig(x, digits=5):
    """floats of the compact line: `digits` significant figures (value / ms_per_step keep full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, digits) for v in x]
    return x


def pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def secondary_of(ex, text_mib):
    """the metric's second half ("... + Mcount/s english.1GB FM-index") out of the sidecar's blocks"""
    name = "Mcount/s, count() of 20-byte patterns, FM-index of a %d MiB text" % text_mib
    if "fm_count" in ex:
        f = ex["fm_count"]
        s = {"metric": name, "value": f["Mcount/s"], "unit": "Mcount/s", "n_gpus": 1, "index_bytes": f.get("index_bytes"),
             "route": f.get("route"), "reference_digest_match": f.get("reference_digest_match"),
             "roofline_frac": (f.get("roofline") or {}).get("frac"),
             "sdsl_stream_bytes": (ex.get("text") or {}).get("sdsl_stream_bytes"), "source": "extras.fm_count"}
        lean = ex.get("fm_count_lean")
        if lean:
            s["lean"] = pick(lean, "Mcount/s", "index_bytes", "route", "reference_digest_match")
            s["lean"]["roofline_frac"] = (lean.get("roofline") or {}).get("frac")
        cb = f.get("cpu_baseline")
        if cb:
            s["cpu_baseline"] = pick(cb, "value", "unit", "cores", "kind", "matches_gpu")
        return s
    if "fm_count_sharded" in ex:
        f = ex["fm_count_sharded"]
        return {"metric": name, "value": f["resident_shards"]["Mcount/s"], "unit": "Mcount/s", "n_gpus": f.get("n_gpus"),
                "scaling": "strong (one batch of %d patterns split over the ranks)" % f["patterns_total"],
                "index_bytes_per_gpu": f.get("index_bytes_per_gpu"),
                "root_owned_batch_Mcount/s": (f.get("root_owned_batch_pipelined") or f.get("root_owned_batch") or {}).get("Mcount/s"),
                "source": "extras.fm_count_sharded.resident_shards"}
    return None


def summary_of(ex):
    """a few figures of the other legs for the line (everything else: the sidecar)"""
    s = {}
    for key, fields in (("select_1", ("Gq/s", "roofline_frac", "reference_digest_match")),
                        ("rrr63_rank_1", ("Gq/s", "bits_per_bit", "traffic_frac", "reference_digest_match")),
                        ("rrr63_select_1", ("Gq/s", "traffic_frac", "reference_digest_match")),
                        ("wt_huff_rank", ("Gq/s", "roofline_frac", "reference_digest_match")),
                        ("wt_huff_select", ("Gq/s", "reference_digest_match")),
                        ("sd_vector", ("rank_1_Gq/s", "select_1_Gq/s", "select_0_Gq/s")),
                        ("fm_sa_access_dens32", ("Msa/s",)), ("fm_extract_64B", ("GB/s", "long_ranges_GB/s", "with_text_resident_GB/s")),
                        ("fm_locate_dens32", ("Gocc/s",)), ("fm_count_rrr63", ("Mcount/s", "x_sdsl_stream_bytes")),
                        ("fm_count_rrr63_lean", ("Mcount/s", "x_sdsl_stream_bytes", "same_answers_as_plain_index"))):
        if key in ex:
            s[key] = pick(ex[key], *fields)
    return s or None


def compact_line(result, ex, sidecar, text_mib):
    """The ONE line: contract keys + config + roofline + cpu_baseline + secondary + end_to_end summary, below LINE_LIMIT bytes
    whatever the legs produced (optional blocks are dropped in a fixed order if it ever grows past the limit)."""
    clip = lambda t, n: t if not isinstance(t, str) or len(t) <= n else t[: n - 3] + "..."  # noqa: E731
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                       "vs_baseline", "dtype", "data", "reference_digest_match")}
    line["config"] = pick(result["config"], "workload", "n_bits", "queries_per_step_per_gpu", "parallelism", "index_bytes_per_gpu",
                          "batch_scratch_bytes_per_gpu", "rank0_host_affinity")
    line["config"]["workload"] = clip(line["config"]["workload"], 320)
    rf = result["roofline"]
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms",
                                               "kernel_ms_per_step", "algorithmic_bytes_per_query")}
    line["roofline"]["kernel"] = clip(rf.get("kernel"), 220)
    line["roofline"]["traffic_source"] = clip(rf.get("traffic_source"), 160)
    if rf.get("direct_kernel"):
        line["roofline"]["direct_kernel"] = pick(rf["direct_kernel"], "Gq/s", "frac", "same_answers")
    cb = result.get("cpu_baseline")
    line["cpu_baseline"] = pick(cb, "value", "unit", "cores", "kind", "sample", "ns_per_query", "matches_gpu") if cb else None
    if cb:
        line["cpu_baseline"]["sample"] = clip(cb.get("sample"), 200)
    ca = result.get("cpu_baseline_all_cores")
    if ca:
        line["cpu_baseline_all_cores"] = pick(ca, "value", "unit", "cores", "kind", "matches_gpu")
        line["cpu_baseline_all_cores"]["cgroup_quota_cpus"] = (ca.get("host_limits") or {}).get("cgroup_quota_cpus", ca.get("cgroup_quota_cpus"))
    if result.get("scaling_columns"):
        line["scaling_columns"] = result["scaling_columns"]
    line["secondary"] = secondary_of(ex, text_mib)
    if "end_to_end" in ex:
        e = ex["end_to_end"]
        line["end_to_end"] = {"what": "same entry point on HOST arrays, wall clock (16 B per query over PCIe); never `value`", "queries": e["queries"],
                              "pageable_Grank/s": e["pageable"]["Grank/s"], "pinned_Grank/s": e["pinned"]["Grank/s"],
                              "same_answers": e["pageable"]["same_answers"] and e["pinned"]["same_answers"]}
    line["summary"] = summary_of(ex)
    if "rank_root_owned_batch" in ex:
        line["scaling_columns"] = {"kernel_only_resident_shards_Grank/s": result["value"],
                                   "end_to_end_root_owned_batch_Grank/s": ex["rank_root_owned_batch"]["Grank/s"],
                                   "device_group": pick(ex.get("device_group") or {}, "kernel_only_resident_shards_Grank/s",
                                                        "end_to_end_root_owned_batch_Grank/s", "error") or None,
                                   "note": "root-owned: rank 0 holds the batch, scatter -> kernels -> gather over RCCL in 8 pieces, 16 B per query cross xGMI"}
    line["extras_file"] = os.path.relpath(sidecar, ROOT) if ex else None
    line["extras_legs"] = [k for k in ex if k not in ("error", "leg_seconds")] or None
    line["extras_error"] = ex.get("error")
    keep_exact = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
    rf_exact = {k: line["roofline"][k] for k in ("achieved", "frac", "kernel_ms")}  # frac == achieved / peak must hold to the last digit
    line = sig(line)
    line.update(keep_exact)
    line["roofline"].update(rf_exact)
    for drop in (None, "extras_legs", "summary", "cpu_baseline_all_cores", "end_to_end", "scaling_columns"):
        if drop:
            line.pop(drop, None)
        txt = json.dumps(line, separators=(",", ":"))
        if len(txt.encode()) < LINE_LIMIT:
            return txt
    raise AssertionError("compact line over the limit even without its optional blocks")


def main_group(a):
    """--mode group: the whole line from one process driving N GPUs."""
    import bench_extras
    N = a.gpus
    if torch.cuda.device_count() < N:
        sys.stderr.write(f"bench.py --mode group: --gpus {N} but only {torch.cuda.device_count()} device(s) are visible\n")
        sys.exit(2)
    pkg = importlib.import_module("sdsl-lite_amd")
    n_bits, nq = 1 << a.log_n, int(a.queries)
    cols = bench_extras.group_leg(pkg, a, N, n_bits, nq, a.steps, a.warmup)
    kernel_ms = cols["resident_shards"]["kernel_ms"]
    achieved = ALG_BYTES["rank"] * nq / (kernel_ms * 1e-3) / 1e9
    traffic = pmc_traffic("rank_bucketed_bytes_per_step")
    result = {
        "metric": "Grank/s, batched rank_1 on a 2^%d-bit vector" % a.log_n, "value": cols["kernel_only_resident_shards_Grank/s"],
        "unit": "Grank/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": cols["resident_shards"]["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[1]: batched rank_1, 2^%d-bit random bit_vector (mt19937_64(42)), %d queries per step per GPU at "
                               "mt19937_64(7 + device) %% (n + 1), index replicated by one RCCL broadcast per buffer, queries and results "
                               "resident in each GPU's HBM" % (a.log_n, nq),
                   "n_bits": n_bits, "queries_per_step_per_gpu": nq,
                   "parallelism": "one process, device group of %d (sdsl_hip_group_*), replicated index, query shards" % N},
        "reference_digest_match": cols["resident_shards"]["reference_digest_match_device0"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": "committed PMC (profiles/pmc_latest.json, same kernel sources)" if traffic else None,
                     "kernel": "bucketed batch rank, slowest device of the group",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_query": ALG_BYTES["rank"]},
        "cpu_baseline": None,
        "scaling_columns": {k: cols[k] for k in ("kernel_only_resident_shards_Grank/s", "end_to_end_root_owned_batch_Grank/s", "replicate_s",
                                                 "index_bytes_per_gpu")},
    }
    sidecar = a.sidecar or os.path.join(ROOT, "bench_extras.json")
    with open(sidecar, "w") as f:
        json.dump({"bench_argv": sys.argv[1:], "n_gpus": N, "extras": {"device_group": cols}}, f)
    result["scaling_columns"]["root_owned_batch"] = pick(cols["root_owned_batch"], "Grank/s", "ms", "queries", "matches_single_gpu")
    keep = {"value": result["value"], "ms_per_step": result["ms_per_step"]}
    result = sig(result)
    result.update(keep)
    result["extras_file"] = os.path.relpath(sidecar, ROOT)
    print(json.dumps(result, separators=(",", ":")), flush=True)


def main():
    a = parse()
    if a.mode == "group":
        return main_group(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import bench_extras
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    cpu_group = None
    if a.backend == "gloo":
        local = local % max(1, n_dev)  # test mode: several ranks may share one GPU
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            cpu_group = dist.new_group(backend="gloo")  # a rendezvous that keeps no kernel spinning on the waiting GPUs

            def barrier():
                dist.barrier(device_ids=[local])
        else:
            dist.init_process_group(a.backend)

            def barrier():
                dist.barrier()
    else:
        def barrier():
            pass
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: the line would report a device count that was not asked for"
    if a.backend == "nccl":
        assert n_dev >= world, f"{world} ranks but {n_dev} visible device(s)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pinned = pin_to_gpu_numa_node(local) if world > 1 and a.backend == "nccl" else None
    comm_dev = dev if a.backend == "nccl" else torch.device("cpu")
    pkg = importlib.import_module("sdsl-lite_amd")
    n_bits = 1 << a.log_n
    nq = int(a.queries)

    # index: replicated (same seed on every rank); queries: this rank's resident shard.  SURVEY.md 8(d) streams.
    G = golden()
    t0 = time.perf_counter()
    # (generated ON the device from generator checkpoints: a rank holds no host copy of its 2 GiB of words and 8 GB of positions —
    # eight ranks on one node would hold 80 GB before the first kernel; the host only walks each generator once, a few seconds)
    words = pkg.rnd_positions_device(42, (n_bits + 63) // 64, 0, 0, local)  # = util::set_random_bits (util.hpp:467-485)
    if a.extras is None:
        a.extras = "e2e,sweep,select,rrr,sd,shapes,wt,fm" if world == 1 else "fm_sharded"
    if a.extras_budget_s is None:
        a.extras_budget_s = 1500.0 if world == 1 else 600.0
    extras = [] if a.extras in ("", "none") else a.extras.split(",")
    bv = pkg.bit_vector(words, n_bits, device=local, select1="select" in extras, select0=False)
    index_bytes = bv.device_bytes()
    idx = pkg.rnd_positions_device(7 + rank, nq, n_bits + 1, 0, local)
    out = torch.empty_like(idx)
    gq = torch.Generator(device=dev).manual_seed(1007 + rank)  # secondary measurements without a reference digest
    setup_s = time.perf_counter() - t0

    # the headline: whatever sdsl_hip_bv_rank_batch does with a device-resident batch by default (large batch over a
    # large vector: the bucketed path, bv_sorted.hip); the direct kernel (one rank line per query) is timed next to it
    step_ms = []
    wall, kernel_ms = time_steps(lambda: bv.rank(idx, 1, out), a.steps, a.warmup, barrier, per_step=step_ms)
    if world > 1:
        wall = pkg.dist.max_over_ranks(wall, comm_dev)
        kernel_ms = pkg.dist.max_over_ranks(kernel_ms, comm_dev)
    value = nq * world * a.steps / wall / 1e9
    achieved = ALG_BYTES["rank"] * nq / (kernel_ms * 1e-3) / 1e9
    ref_ok = None
    if rank == 0 and a.log_n == G.get("c2", {}).get("log_n") and nq >= G["c2"]["rank_1"]["n"]:
        ref_ok = digests_match(out, G["c2"], "rank_1") and bv.ones() == G["c2"]["ones"]  # first 10^7 answers + every 100th of the 10^9
    # the passes of the step, one by one, over as many traced steps as were timed (tracing synchronises after every step, so
    # these runs are not the timed ones)
    pkg.set_option("trace_phases", 1)
    traced = []
    for _ in range(max(3, min(a.steps, 20))):
        bv.rank(idx, 1, out)
        torch.cuda.synchronize()
        ph = pkg.last_phases()
        ph.pop("select", None)
        traced.append(ph)
    pkg.set_option("trace_phases", 0)
    phases = {k: spread_of([t[k] for t in traced if k in t])["median"] for k in traced[0]} if traced[0] else {}
    phases_spread = {k: spread_of([t[k] for t in traced if k in t]) for k in traced[0]} if traced[0] else None
    bucketed = bool(phases)
    scratch_bytes = pkg.device_scratch_bytes(dev.index if dev.index is not None else 0)  # one pool per device, shared by all handles
    # the direct kernel and its access skeleton (read a position, fetch its 64-byte rank line, write a word), same table,
    # same positions, same run
    pkg.set_option("rank_sorted", 0)
    out_d = torch.empty_like(idx)
    _, direct_ms = time_steps(lambda: bv.rank(idx, 1, out_d), max(2, a.steps // 2), 1, barrier)
    same = bool(torch.equal(out, out_d))
    _, probe_ms = time_steps(lambda: bv.gather_probe(idx, out_d), max(2, a.steps // 2), 1, barrier)
    del out_d
    pkg.set_option("rank_sorted", -1)
    traffic = pmc_traffic("rank_bucketed_bytes_per_step" if bucketed else "k_rank_bytes_per_launch")
    result = {
        "metric": "Grank/s, batched rank_1 on a 2^%d-bit vector" % a.log_n, "value": value, "unit": "Grank/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[1]: batched rank_1, 2^%d-bit random bit_vector (words = mt19937_64(42)), %d queries per step "
                               "per GPU at mt19937_64(7 + rank) %% (n + 1), index + queries + results resident in HBM" % (a.log_n, nq),
                   "n_bits": n_bits, "queries_per_step_per_gpu": nq, "parallelism": "replicated index, query shards x%d" % world,
                   "index_bytes_per_gpu": index_bytes, "batch_scratch_bytes_per_gpu": scratch_bytes, "rank0_host_affinity": pinned},
        "reference_digest_match": ref_ok,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("committed PMC collection (profiles/pmc_latest.json; valid for these kernel sources, sha "
                                        + kernel_sources_sha() + "), not re-measured in this run") if traffic else None,
                     "kernel": ("bucketed batch rank: all kernels of one step (bv_swc.hip + bv_sorted.hip: sample, histogram, 2 partition "
                                "passes, k_sr_rank_lds, 2 un-permute passes)") if bucketed else "sdslhip::k_rank<4,false,true>",
                     "kernel_ms": kernel_ms, "kernel_ms_per_step": spread_of(step_ms),
                     "algorithmic_bytes_per_query": ALG_BYTES["rank"],
                     # (the line keeps the keys above and direct_kernel's rate; the rest of this block is for the sidecar)
                     "phases_ms": phases or None, "phases_ms_spread": phases_spread,
                     "direct_kernel": {"kernel": "sdslhip::k_rank<4,false,true>", "kernel_ms": direct_ms, "Gq/s": nq / direct_ms / 1e6,
                                       "frac": ALG_BYTES["rank"] * nq / (direct_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "same_answers": same,
                                       "access_skeleton_ms": probe_ms, "kernel_over_skeleton": probe_ms / direct_ms}},
        "setup_s": setup_s, "box": box_facts(local) if rank == 0 else None,
    }
    if rank == 0 and world == 1 and not a.no_cpu:
        result["cpu_baseline"], result["cpu_baseline_all_cores"] = cpu_baseline(pkg, words, n_bits, idx, out, a.cpu_seconds)
    else:
        result["cpu_baseline"] = None

    sidecar = a.sidecar or os.path.join(ROOT, "bench_extras.json")
    c = bench_extras.Ctx(a=a, pkg=pkg, dev=dev, local=local, rank=rank, world=world, barrier=barrier, comm_dev=comm_dev, G=G, gq=gq,
                         nq=nq, n_bits=n_bits, bv=bv, words=words, idx=idx, out=out, extras=extras, sidecar=sidecar, cpu_group=cpu_group,
                         done=[])
    c.ex["headline"] = result  # in full: the line keeps a whitelist of it (compact_line)
    del bv, words, idx, out
    printed = threading.Lock()

    def emit():
        """prints the line exactly once (the watchdog and the normal end race for it)"""
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            bench_extras.write_sidecar(c)
            try:
                sys.stderr.write("bench_extras: " + json.dumps(c.ex) + "\n")
                sys.stderr.flush()
            except Exception:
                pass
            print(compact_line(result, c.ex, sidecar, a.text_mib), flush=True)

    def watchdog():
        c.ex["error"] = (c.ex.get("error") or "") + " watchdog: the legs outlived --extras-budget-s %g; line printed, process left" % a.extras_budget_s
        emit()
        os._exit(0)

    wd = threading.Timer(a.extras_budget_s, watchdog)
    wd.daemon = True
    if extras:
        wd.start()
        bench_extras.run_extras(c)
        if world > 1 and "group" in extras and a.backend == "nccl":
            # the other driver of 8(e): rank 0 alone drives all the GPUs through the C ABI's device group while the other ranks (their
            # memory released) wait at the barrier.  Opt-in; every failure lands in the sidecar, never on the line
            try:
                import torch.distributed as dist
                c.bv = c.idx = c.out = None
                torch.cuda.empty_cache()
                torch.cuda.synchronize()
                dist.barrier(group=cpu_group)
                if rank == 0:
                    c.ex["device_group"] = bench_extras.group_leg(pkg, a, world, n_bits, min(nq, 250_000_000), max(3, a.steps // 2), 1)
                    torch.cuda.set_device(local)
                dist.barrier(group=cpu_group)
            except Exception as e:
                c.ex["device_group"] = {"error": f"{type(e).__name__}: {e}"}
        wd.cancel()
    emit()
    if world > 1:
        import torch.distributed as dist
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
