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

Optional secondary measurements (`--extras select,rrr,sd,shapes,wt,fm`) are reported under "extras"; they never
enter the timed region.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy achieves
FUSED_NOTE = ("survey_8d_model_frac prices the query as SURVEY.md 8(d) does (80 bytes per tree level: the REFERENCE's level-by-level walk); "
              "the kernel walks the fused layout (one 128-byte line per level of its own 8-ary tree) and does not move those bytes, so that "
              "figure can exceed 1 and is no roofline fraction.  roofline_frac is: measured fabric traffic of the kernel (PMC, "
              "profiles/pmc_latest.json, when it was collected on these kernel sources) over the 8 TB/s peak, else line_fetch_frac — the "
              "lines the fused walk addresses x 128 B, an upper bound on its HBM traffic (small nodes stay in cache, the k-mer table "
              "skips the first characters of a pattern)")


def fused_frac(key, n, ms, line_frac):
    """HBM-roofline fraction of a fused-layout kernel: measured fabric bytes per launch (per query x n) if available"""
    per_q = pmc_traffic(key)
    if per_q:
        return per_q * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "pmc"
    return line_frac, "lines_addressed"
ALG_BYTES = {"rank": 96, "select": 112, "rrr": 144}  # SURVEY.md §8(d), bytes per query


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


def time_steps(fn, steps, warmup, barrier, per_step=None):
    """barrier + synchronize on both sides of exactly `steps` calls; HIP events on the launch stream
    give the average kernel duration of the same region (and, with `per_step`, every step's own duration: an event between
    consecutive steps costs nothing — the stream is in order anyway)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    if per_step is not None:
        per_step.extend(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return wall, ev[0].elapsed_time(ev[steps]) / steps


def spread_of(xs):
    xs = sorted(xs)
    return {"min": xs[0], "median": xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2]), "max": xs[-1]}


def kernel_sources_sha():
    """sha256 over the kernel sources a PMC measurement is valid for (every .hip / .hpp of the library)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sdsl-lite_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key):
    """HBM bytes per step from the committed PMC summary (profiles/pmc_latest.json) — only if that summary was
    collected on these very kernel sources (it carries their sha256); otherwise None: a stale number is not a
    measurement of this run."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
        if d.get("kernel_sources_sha") != kernel_sources_sha():
            return None
        return d.get(kernel_key)
    except Exception:
        return None


def golden():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json")))
    except Exception:
        return {}


def digest_matches(ans_dev, want):
    """The first want['n'] answers against the reference's sum / xor / sha256 / first answers (golden_large.json)."""
    import hashlib
    a = ans_dev[: want["n"]].cpu().numpy().view(np.uint64)
    if a.size != want["n"]:
        return None
    first = np.array(want["first"], dtype=np.uint64)
    return bool(np.array_equal(a[: first.size], first) and int(np.add.reduce(a, dtype=np.uint64)) == want["sum"]
                and int(np.bitwise_xor.reduce(a)) == want["xor"] and hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"])


def to_dev(host_u64, dev):
    """uint64 numpy array -> int64 device tensor (chunked: no second full-size pinned copy on the host)."""
    t = torch.empty(host_u64.size, dtype=torch.int64, device=dev)
    step = 1 << 27
    for s0 in range(0, host_u64.size, step):
        t[s0:s0 + step].copy_(torch.from_numpy(host_u64[s0:s0 + step].view(np.int64)))
    return t


def box_facts(dev_index):
    """Clocks, power cap and memory of the GPU this run landed on (box-to-box spread of the same binary is +-8 %)."""
    import subprocess
    out = {}
    try:
        p = torch.cuda.get_device_properties(dev_index)
        out.update(name=p.name, cus=p.multi_processor_count, total_mem_gib=round(p.total_memory / 2**30, 1))
    except Exception:
        pass
    try:
        r = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showclocks", "--showpower", "--showmaxpower", "--showperflevel",
                            "--showmemuse", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        card = next(iter(j.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "power", "performance level", "memory")):
                keep[k] = v
        out["rocm_smi"] = keep
    except Exception as e:
        out["rocm_smi"] = f"unavailable: {type(e).__name__}"
    return out


def host_cpu_limits():
    """What the container may really use: affinity mask, cgroup CPU quota (v2 cpu.max / v1 cfs quota), cpuset, NUMA nodes."""
    out = {}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        out["affinity_cpus"] = os.cpu_count()
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        out["cgroup_cpu_max"] = f"{q} {per}"
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_cfs_quota_us"] = q
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    for f in ("/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.effective_cpus"):
        try:
            out["cpuset_effective"] = open(f).read().strip()
            break
        except OSError:
            pass
    try:
        out["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        out["numa_nodes"] = None
    out["cgroup_quota_cpus"] = quota
    out["effective_cpus"] = min(out["affinity_cpus"], quota) if quota else float(out["affinity_cpus"])
    return out


def set_mempolicy_interleave(on):
    """MPOL_INTERLEAVE over all NUMA nodes for this thread's next allocations (off: back to the default policy); False if the
    kernel refuses (no NUMA, no permission)."""
    import ctypes
    try:
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
        if nodes < 2 and on:
            return False
        libc = ctypes.CDLL(None, use_errno=True)
        mask = ctypes.c_ulong((1 << nodes) - 1)
        r = libc.syscall(238, 3 if on else 0, ctypes.byref(mask) if on else None, nodes + 1 if on else 0)  # set_mempolicy
        return r == 0
    except Exception:
        return False


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


def cpu_time(run, args_dev, gpu_out_dev, seconds, unit_scale, what):
    """Times `run(*host_arrays)` (a scalar CPU loop of the reference / its restatement) on a bounded prefix of
    the step's arguments and checks the answers against the GPU's."""
    probe = [a[:200_000].cpu().numpy() for a in args_dev]
    t0 = time.perf_counter()
    run(*probe)
    per_q = (time.perf_counter() - t0) / 200_000
    n_s = int(min(args_dev[0].shape[0], max(200_000, seconds / per_q)))
    host = [a[:n_s].cpu().numpy() for a in args_dev]
    t0 = time.perf_counter()
    res = run(*host)
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(np.asarray(res).view(np.uint64), gpu_out_dev[:n_s].cpu().numpy().view(np.uint64)))
    return {"value": n_s / dt / unit_scale, "ns_per_query": dt / n_s * 1e9, "cores": 1, "sample": f"first {n_s} {what}",
            "matches_gpu": same}


def synthetic_text(n_bytes, seed, device):
    """English-like stand-in for Pizza&Chili english (not available offline): words drawn from a fixed
    4096-word vocabulary with a Zipf-like distribution, separated by spaces.  Built on the device."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    vocab_n, max_len = 4096, 12
    lens = torch.randint(2, max_len + 1, (vocab_n,), generator=g)
    letters = torch.tensor(list(b"etaoinshrdlcumwfgypbvkjxqz"), dtype=torch.uint8)
    lw = torch.arange(1, 27, dtype=torch.float64).pow(-0.9)
    vocab = torch.zeros(vocab_n, max_len + 1, dtype=torch.uint8)
    pick = torch.multinomial(lw, vocab_n * max_len, replacement=True, generator=g).view(vocab_n, max_len)
    vocab[:, :max_len] = letters[pick]
    for i in range(vocab_n):
        vocab[i, lens[i]:] = 0
        vocab[i, lens[i]] = 32
    zipf = torch.arange(1, vocab_n + 1, dtype=torch.float64).pow(-1.0)
    cdf = torch.cumsum(zipf / zipf.sum(), 0).to(device)
    gd = torch.Generator(device=device).manual_seed(seed)
    vocab_d = vocab.to(device)
    out = torch.empty(n_bytes, dtype=torch.uint8, device=device)
    filled, chunk_words = 0, 1 << 24  # chunked: boolean compaction of > 2^31 elements is not safe in torch
    while filled < n_bytes:
        u = torch.rand(chunk_words, device=device, dtype=torch.float64, generator=gd)
        ids = torch.searchsorted(cdf, u).clamp_(max=vocab_n - 1)
        piece = vocab_d[ids].reshape(-1)
        piece = piece[piece != 0]
        take = min(piece.numel(), n_bytes - filled)
        out[filled:filled + take] = piece[:take]
        filled += take
    return out


def group_leg(pkg, a, N, n_bits, nq, steps, warmup):
    """ONE process, N GPUs, through the C ABI's device group (sdsl_hip_group_*, csrc/group.cpp) — the path a header-only C++ caller
    has.  Both columns of SURVEY.md 8(e): resident shards (every device answers nq positions that already live in its HBM, no
    collective) and a root-owned batch (device 0 holds all the positions: scatter -> kernels -> gather over RCCL in 8 pieces)."""
    G = golden()
    devs = list(range(N))
    d0 = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    words = to_dev(pkg.set_random_bits(n_bits, 42), d0)
    bv0 = pkg.bit_vector(words, n_bits, device=0, select1=False, select0=False)
    del words
    t0 = time.perf_counter()
    grp = pkg.device_group(devs)
    reps = grp.replicate(bv0)
    for r in devs:
        torch.cuda.synchronize(r)
    repl_s = time.perf_counter() - t0
    idx_d, out_d = [], []
    for r in devs:
        idx_d.append(pkg.rnd_positions_device(7 + r, nq, n_bits + 1, 0, r))
        out_d.append(torch.empty_like(idx_d[r]))

    def step():
        for r in devs:
            torch.cuda.set_device(r)
            reps[r].rank(idx_d[r], 1, out_d[r])

    def sync_all():
        for r in devs:
            torch.cuda.synchronize(r)

    for _ in range(warmup):
        step()
    sync_all()
    ev = []
    for r in devs:
        torch.cuda.set_device(r)
        ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
    t0 = time.perf_counter()
    for r in devs:
        torch.cuda.set_device(r)
        ev[r][0].record()
    for _ in range(steps):
        step()
    for r in devs:
        torch.cuda.set_device(r)
        ev[r][1].record()
    sync_all()
    wall = time.perf_counter() - t0
    kernel_ms = max(ev[r][0].elapsed_time(ev[r][1]) for r in devs) / steps
    ref_ok = None
    if a.log_n == G.get("c2", {}).get("log_n") and nq >= G["c2"]["rank_1"]["n"]:
        ref_ok = digest_matches(out_d[0], G["c2"]["rank_1"])
    resident = {"Grank/s": nq * N * steps / wall / 1e9, "ms_per_step": wall / steps * 1e3, "kernel_ms": kernel_ms,
                "reference_digest_match_device0": ref_ok}
    # root-owned batch
    torch.cuda.set_device(0)
    del idx_d[1:], out_d[1:]
    nro = min(nq, 250_000_000) * N
    gq = torch.Generator(device=d0).manual_seed(1007)
    allq = torch.randint(0, n_bits + 1, (nro,), device=d0, dtype=torch.int64, generator=gq)
    ro = torch.empty_like(allq)
    grp.rank(reps, allq, 1, ro, chunks=8)
    sync_all()
    t0 = time.perf_counter()
    reps_ro = max(2, steps // 4)
    for _ in range(reps_ro):
        grp.rank(reps, allq, 1, ro, chunks=8)
    sync_all()
    dt = (time.perf_counter() - t0) / reps_ro
    chk = bv0.rank(allq[:1_000_000].clone(), 1)
    root = {"Grank/s": nro / dt / 1e9, "ms": dt * 1e3, "queries": nro, "pieces": 8, "bytes_over_links_per_query": 16,
            "matches_single_gpu": bool(torch.equal(chk, ro[:1_000_000]))}
    index_bytes = bv0.device_bytes()
    del allq, ro, chk, idx_d, out_d
    for o in reps[1:]:
        o.close()
    bv0.close()
    grp.close()
    for r in devs:
        with torch.cuda.device(r):
            torch.cuda.empty_cache()
    return {"driver": "device group (one process, C ABI sdsl_hip_group_*)", "n_gpus": N, "replicate_s": repl_s,
            "index_bytes_per_gpu": index_bytes,
            "kernel_only_resident_shards_Grank/s": resident["Grank/s"], "resident_shards": resident,
            "end_to_end_root_owned_batch_Grank/s": root["Grank/s"], "root_owned_batch": root}


def main_group(a):
    """--mode group: the whole line from one process driving N GPUs."""
    N = a.gpus
    if torch.cuda.device_count() < N:
        sys.stderr.write(f"bench.py --mode group: --gpus {N} but only {torch.cuda.device_count()} device(s) are visible\n")
        sys.exit(2)
    pkg = importlib.import_module("sdsl-lite_amd")
    n_bits, nq = 1 << a.log_n, int(a.queries)
    cols = group_leg(pkg, a, N, n_bits, nq, a.steps, a.warmup)
    kernel_ms = cols["resident_shards"]["kernel_ms"]
    achieved = ALG_BYTES["rank"] * nq / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "Grank/s, batched rank_1 on a 2^%d-bit vector" % a.log_n, "value": cols["kernel_only_resident_shards_Grank/s"],
        "unit": "Grank/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": cols["resident_shards"]["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[1]: batched rank_1 on a 2^%d-bit random bit_vector (words = mt19937_64(42)), %d queries per step "
                               "per GPU at mt19937_64(7 + device) %% (n + 1), index replicated by one RCCL broadcast per buffer, queries and "
                               "results resident in each GPU's HBM" % (a.log_n, nq),
                   "n_bits": n_bits, "queries_per_step_per_gpu": nq,
                   "parallelism": "one process, device group of %d (sdsl_hip_group_*), replicated index, query shards" % N},
        "reference_digest_match": cols["resident_shards"]["reference_digest_match_device0"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("rank_bucketed_bytes_per_step"), "kernel": "bucketed batch rank, slowest device of the group",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_query": ALG_BYTES["rank"]},
        "cpu_baseline": None,
        "scaling_columns": cols,
    }
    print(json.dumps(result))


def main():
    a = parse()
    if a.mode == "group":
        return main_group(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
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
        ref_ok = digest_matches(out, G["c2"]["rank_1"]) and bv.ones() == G["c2"]["ones"]
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
    result = {
        "metric": "Grank/s, batched rank_1 on a 2^%d-bit vector" % a.log_n, "value": value, "unit": "Grank/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[1]: batched rank_1 on a 2^%d-bit random bit_vector (words = mt19937_64(42), density "
                               "0.5), %d queries per step per GPU at mt19937_64(7 + rank) %% (n + 1), index+queries+results "
                               "resident in HBM" % (a.log_n, nq),
                   "n_bits": n_bits, "queries_per_step_per_gpu": nq, "parallelism": "replicated index, query shards x%d" % world,
                   "index_bytes_per_gpu": index_bytes, "batch_scratch_bytes_per_gpu": scratch_bytes, "setup_s": setup_s},
        "reference_digest_match": ref_ok,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("rank_bucketed_bytes_per_step" if bucketed else "k_rank_bytes_per_launch"),
                     "kernel": ("bucketed batch rank (bv_swc.hip + bv_sorted.hip): k_sr_sample_spread, k_sw_hist, k_sw_partition<1>, "
                                "k_sw_partition<2>, k_sr_rank_lds, k_sw_unpermute_dma<2>, k_sw_unpermute_dma<1> + 7 table kernels and two "
                                "memsets (and the direct kernel, which returns at once when the sample says 'spread'); kernel_ms = all of "
                                "them, one step") if bucketed else "sdslhip::k_rank<4,false,true>",
                     "kernel_ms": kernel_ms, "kernel_ms_per_step": spread_of(step_ms), "phases_ms": phases or None,
                     "phases_ms_spread": phases_spread,
                     "algorithmic_bytes_per_query": ALG_BYTES["rank"],
                     "direct_kernel": {"kernel": "sdslhip::k_rank<4,false,true>", "kernel_ms": direct_ms,
                                       "Gq/s": nq / direct_ms / 1e6,
                                       "frac": ALG_BYTES["rank"] * nq / (direct_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "same_answers": same,
                                       "access_skeleton_ms": probe_ms, "kernel_over_skeleton": probe_ms / direct_ms}},
        "box": box_facts(local) if rank == 0 else None,
    }
    if rank == 0 and world == 1 and not a.no_cpu:
        result["cpu_baseline"], result["cpu_baseline_all_cores"] = cpu_baseline(pkg, words, n_bits, idx, out,
                                                                                 a.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None

    ex = {}
    try:
        if "e2e" in extras and world == 1 and rank == 0:
            # kernel-only (the headline) AND end-to-end.  The same entry point handed HOST arrays: 16 bytes per
            # query cross PCIe (8 up, 8 down); the library cuts the batch into chunks that travel on two streams, so upload,
            # kernel and download overlap (common.hpp: host_pipeline_u64).  Never `value`.
            ne = min(nq, 250_000_000)
            want_e = out[:ne].cpu().numpy().view(np.uint64)
            legs = {}
            for kind in ("pageable", "pinned"):
                if kind == "pageable":
                    h_idx = idx[:ne].cpu().numpy().view(np.uint64)
                    h_out = np.zeros(ne, dtype=np.uint64)
                else:
                    t_idx = torch.empty(ne, dtype=torch.int64).pin_memory()
                    t_idx.copy_(idx[:ne])
                    t_out = torch.zeros(ne, dtype=torch.int64).pin_memory()
                    h_idx, h_out = t_idx.numpy().view(np.uint64), t_out.numpy().view(np.uint64)
                bv.rank(h_idx, 1, h_out)  # warm-up (first touch of the result pages, the pipeline's staging buffers)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    bv.rank(h_idx, 1, h_out)  # returns when the answers are in h_out
                    ts.append(time.perf_counter() - t0)
                sec = sorted(ts)[1]
                legs[kind] = {"Grank/s": ne / sec / 1e9, "seconds": spread_of(ts), "pcie_GB/s_both_directions": 16 * ne / sec / 1e9,
                              "same_answers": bool(np.array_equal(h_out, want_e))}
            result["end_to_end"] = {"what": "sdsl_hip_bv_rank_batch on HOST arrays (positions in, answers out), wall clock around the call",
                                    "queries": ne, "bytes_over_pcie_per_query": 16, **legs,
                                    "pcie_note": "PCIe 5.0 x16: 64 GB/s per direction on paper, ~55 achievable; the kernel-only rate is `value`"}
            del h_idx, h_out, want_e
            if kind == "pinned":
                del t_idx, t_out
        if "sweep" in extras and world == 1 and rank == 0:
            # where the routes cross: batch size x vector size, default dispatch / direct kernel / bucketed passes forced.
            # (vector words and positions from the device's generator: no reference digest at these sizes, the routes check each other)
            sweep = []
            for ln in (30, a.log_n, 36):
                nb = 1 << ln
                if ln == a.log_n:
                    bs_ = bv
                else:
                    w_ = torch.randint(-2**63, 2**63 - 1, (nb // 64,), device=dev, dtype=torch.int64, generator=gq)
                    bs_ = pkg.bit_vector(w_, nb, device=local, select1=False, select0=False)
                    del w_
                for nqs in (10**5, 10**6, 10**7, 10**8, 10**9):
                    if nqs > nq:
                        continue
                    qi = torch.randint(0, nb + 1, (nqs,), device=dev, dtype=torch.int64, generator=gq)
                    o_ = [torch.empty_like(qi) for _ in range(3)]
                    row = {"n_bits_log2": ln, "queries": nqs}
                    for j, (route, opt) in enumerate((("default", -1), ("direct", 0), ("bucketed", 1))):
                        pkg.set_option("rank_sorted", opt)
                        pkg.set_option("trace_phases", 1)
                        bs_.rank(qi, 1, o_[j])
                        torch.cuda.synchronize()
                        took_passes = bool(pkg.last_phases())
                        pkg.set_option("trace_phases", 0)
                        if route == "bucketed" and not took_passes:
                            row[route] = None  # the passes do not apply to this vector / batch (bv_sorted.hip: bv_sorted_rank_possible)
                            continue
                        _, ms_ = time_steps(lambda: bs_.rank(qi, 1, o_[j]), 5 if nqs >= 10**8 else 20, 1, barrier)
                        row[route] = {"Grank/s": nqs / ms_ / 1e6, "kernel_ms": ms_}
                        if route == "default":
                            row[route]["route"] = "bucketed" if took_passes else "direct"
                    row["same_answers"] = bool(torch.equal(o_[0], o_[1]) and (row["bucketed"] is None or torch.equal(o_[0], o_[2])))
                    sweep.append(row)
                    del qi, o_
                pkg.set_option("rank_sorted", -1)
                if bs_ is not bv:
                    del bs_
                    torch.cuda.empty_cache()
            ex["batch_sweep"] = sweep
        if "select" in extras:
            ones = bv.ones()
            si = to_dev(pkg.rnd_positions(11, nq, ones, 1), dev)  # 8(d): 1 + mt19937_64(11) % ones
            _, ms = time_steps(lambda: bv.select(si, 1, out), max(2, a.steps // 2), 1, barrier)
            pkg.set_option("trace_phases", 1)
            bv.select(si, 1, out)
            torch.cuda.synchronize()
            sph = pkg.last_phases()
            pkg.set_option("trace_phases", 0)
            sel_bucketed = sph.pop("select", 0) == 1
            ex["select_1"] = {"Gq/s": nq / ms / 1e6, "kernel_ms": ms,
                              "path": "bucketed (bv_sorted.hip, DESIGN.md 3.5b)" if sel_bucketed else "direct kernel",
                              "phases_ms": sph if sel_bucketed else None,
                              "roofline_frac": ALG_BYTES["select"] * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if a.log_n == G.get("c2", {}).get("log_n") and nq >= G["c2"]["select_1"]["n"]:
                ex["select_1"]["reference_digest_match"] = digest_matches(out, G["c2"]["select_1"])
            # the default above is the bucketed path for a batch of this size; the direct kernel beside it
            pkg.set_option("select_sorted", 0)
            out_d = torch.empty_like(out)
            _, ms_d = time_steps(lambda: bv.select(si, 1, out_d), max(2, a.steps // 2), 1, barrier)
            pkg.set_option("select_sorted", -1)
            ex["select_1"]["direct_kernel"] = {"Gq/s": nq / ms_d / 1e6, "kernel_ms": ms_d, "same_answers": bool(torch.equal(out, out_d)),
                                               "roofline_frac": ALG_BYTES["select"] * nq / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del out_d
            pos = out[: 1 << 20].clone()
            assert bool((bv.rank(pos, 1) == si[: 1 << 20] - 1).all()), "select/rank round trip failed"
            if rank == 0 and world == 1 and not a.no_cpu:
                import oracle_lib as ol
                if ol.have_ref():  # the real select_support_mcl<1>; ref_bv_create builds it together with the rank supports
                    wp = ol.padded(words.cpu().numpy().view(np.uint64), n_bits)
                    hh = ol.ref().L.ref_bv_create(wp.ctypes.data, n_bits)

                    def run_sel(i):
                        o = np.empty(i.size, dtype=np.uint64)
                        ii = np.ascontiguousarray(i).view(np.uint64)
                        ol.ref().L.ref_bv_select(hh, 1, ii.ctypes.data, ii.size, o.ctypes.data)
                        return o
                    bv.select(si, 1, out)
                    cb = cpu_time(run_sel, [si], out, a.cpu_seconds, 1e9, "select_1 arguments, select_support_mcl<1>")
                    cb.update(unit="Gselect/s", kind="reference")
                    ex["select_1"]["cpu_baseline"] = cb
                    ol.ref().L.ref_bv_destroy(hh)
            del si
        del words
        if "rrr" in extras:
            del bv
            torch.cuda.empty_cache()
            # 5 % dense 2^log_n-bit vector (BASELINE.json configs[2]): bit i = (mt19937_64(9)_i % 100 < 5), produced by all
            # host threads from the committed generator checkpoints (tests/golden/mt9_checkpoints.bin)
            c3 = G.get("c3", {})
            ckp = os.path.join(ROOT, "tests", "golden", "mt9_checkpoints.bin")
            if os.path.exists(ckp) and c3:
                ck = np.fromfile(ckp, dtype=np.uint64).reshape(-1, 313)
                w5h_all = pkg.density_bits(n_bits, 9, 5, ck, c3["checkpoint_stride"])
            else:
                w5h_all = pkg.density_bits(n_bits, 9, 5)
            w5 = to_dev(w5h_all, dev)
            t0 = time.perf_counter()
            rv = pkg.rrr_vector(w5, n_bits, device=local)
            build = time.perf_counter() - t0
            w5h = w5h_all if (rank == 0 and world == 1 and not a.no_cpu) else None
            del w5, w5h_all
            rrr_bytes = rv.device_bytes()
            _, ms = time_steps(lambda: rv.rank(idx, 1, out), max(2, a.steps // 2), 1, barrier)
            ex["rrr63_rank_1"] = {"Gq/s": nq / ms / 1e6, "kernel_ms": ms, "build_s": build,
                                  "bits_per_bit": rrr_bytes * 8 / n_bits,
                                  "path": "default dispatch (a spread batch of this size: the passes of bv_swc.hip around the slice-wise "
                                          "decoder of rrr_sorted.hip)",
                                  "roofline_frac": ALG_BYTES["rrr"] * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            bq = pmc_traffic("rrr_rank_bucketed_bytes_per_query")
            ex["rrr63_rank_1"]["fabric_traffic"] = {
                "bytes_per_query": bq, "frac_of_hbm_peak": bq * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bq else None,
                "note": "measured fabric bytes of ALL kernels of a bucketed step (tools/rrr_probe.py under the counters, "
                        "profiles/pmc_latest.json) over the 8 TB/s peak — the honest fraction of this path: roofline_frac prices every "
                        "query at SURVEY 8(d)'s 144 bytes, which a batch that reads each record once does not move (it can exceed 1)"}
            # the direct kernel (one record fetch and one block decode per query) beside it, same answers
            pkg.set_option("rrr_sorted", 0)
            out_d = torch.empty_like(out)
            _, ms_d = time_steps(lambda: rv.rank(idx, 1, out_d), 2, 1, barrier)
            pkg.set_option("rrr_sorted", -1)
            ex["rrr63_rank_1"]["direct_kernel"] = {"Gq/s": nq / ms_d / 1e6, "kernel_ms": ms_d, "same_answers": bool(torch.equal(out, out_d)),
                                                   "roofline_frac": ALG_BYTES["rrr"] * nq / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del out_d
            c3ok = a.log_n == c3.get("log_n") and nq >= c3.get("rank_1", {}).get("n", 1 << 62) and rank == 0
            if c3ok:
                ex["rrr63_rank_1"]["reference_digest_match"] = digest_matches(out, c3["rank_1"]) and rv.ones() == c3["ones"]
            si = to_dev(pkg.rnd_positions(11, nq, rv.ones(), 1), dev)
            _, ms = time_steps(lambda: rv.select(si, 1, out), max(2, a.steps // 2), 1, barrier)
            ex["rrr63_select_1"] = {"Gq/s": nq / ms / 1e6, "kernel_ms": ms,
                                    "roofline_frac": ALG_BYTES["rrr"] * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            bq = pmc_traffic("rrr_select_bucketed_bytes_per_query")
            ex["rrr63_select_1"]["fabric_traffic"] = {"bytes_per_query": bq,
                                                      "frac_of_hbm_peak": bq * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bq else None}
            pkg.set_option("rrr_sorted", 0)
            out_d = torch.empty_like(out)
            _, ms_d = time_steps(lambda: rv.select(si, 1, out_d), 2, 1, barrier)
            pkg.set_option("rrr_sorted", -1)
            ex["rrr63_select_1"]["direct_kernel"] = {"Gq/s": nq / ms_d / 1e6, "kernel_ms": ms_d, "same_answers": bool(torch.equal(out, out_d)),
                                                     "roofline_frac": ALG_BYTES["rrr"] * nq / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del out_d
            if c3ok:
                ex["rrr63_select_1"]["reference_digest_match"] = digest_matches(out, c3["select_1"])
            assert bool((rv.rank(out[: 1 << 20].clone(), 1) == si[: 1 << 20] - 1).all())
            if rank == 0 and world == 1 and not a.no_cpu:
                import oracle_lib as ol
                if ol.have_ref():  # the real rrr_vector<63> built from the same bits
                    wp = ol.padded(w5h, n_bits)
                    t0 = time.perf_counter()
                    hh = ol.ref().L.ref_rrr_create(wp.ctypes.data, n_bits)
                    cpu_build = time.perf_counter() - t0

                    def run_rank(i):
                        o = np.empty(i.size, dtype=np.uint64)
                        ii = np.ascontiguousarray(i).view(np.uint64)
                        ol.ref().L.ref_rrr_rank(hh, 1, ii.ctypes.data, ii.size, o.ctypes.data)
                        return o

                    def run_sel(i):
                        o = np.empty(i.size, dtype=np.uint64)
                        ii = np.ascontiguousarray(i).view(np.uint64)
                        ol.ref().L.ref_rrr_select(hh, 1, ii.ctypes.data, ii.size, o.ctypes.data)
                        return o
                    rv.rank(idx, 1, out)
                    cb = cpu_time(run_rank, [idx], out, a.cpu_seconds, 1e9, "rank_1 arguments, rank_support_rrr<1,63>")
                    cb.update(unit="Grank/s", kind="reference", build_s=cpu_build)
                    ex["rrr63_rank_1"]["cpu_baseline"] = cb
                    rv.select(si, 1, out)
                    cb = cpu_time(run_sel, [si], out, a.cpu_seconds, 1e9, "select_1 arguments, select_support_rrr<1,63>")
                    cb.update(unit="Gselect/s", kind="reference")
                    ex["rrr63_select_1"]["cpu_baseline"] = cb
                    ol.ref().L.ref_rrr_destroy(hh)
            del rv, si
        if "sd" in extras:
            # sd_vector<> (Elias-Fano): 2^28 ones over a universe of 2^40 — the plain vector would need 128 GiB
            torch.cuda.empty_cache()
            N_sd = 1 << 40
            pos = torch.unique(torch.randint(0, N_sd, (1 << 28,), device=dev, dtype=torch.int64, generator=gq))
            t0 = time.perf_counter()
            sd = pkg.sd_vector(positions=pos, n_bits=N_sd, device=local)
            torch.cuda.synchronize()
            sd_build = time.perf_counter() - t0
            nq_sd = min(nq, 100_000_000)
            xi = torch.randint(0, N_sd + 1, (nq_sd,), device=dev, dtype=torch.int64, generator=gq)
            o_sd = torch.empty(nq_sd, dtype=torch.int64, device=dev)
            _, ms_r = time_steps(lambda: sd.rank(xi, 1, o_sd), 3, 1, barrier)
            assert torch.equal(o_sd[:1_000_000], torch.searchsorted(pos, xi[:1_000_000], right=False))
            si = torch.randint(1, pos.numel() + 1, (nq_sd,), device=dev, dtype=torch.int64, generator=gq)
            _, ms_s = time_steps(lambda: sd.select(si, 1, o_sd), 3, 1, barrier)
            assert torch.equal(o_sd, pos[si - 1])
            zi = torch.randint(1, N_sd - pos.numel() + 1, (nq_sd,), device=dev, dtype=torch.int64, generator=gq)
            _, ms_z = time_steps(lambda: sd.select(zi, 0, o_sd), 3, 1, barrier)
            # the i-th zero sits at p with p - rank_1(p) == i - 1 and bit p clear
            zr = sd.rank(o_sd[:1_000_000], 1)
            assert torch.equal(o_sd[:1_000_000] - zr, zi[:1_000_000] - 1)
            assert bool((sd.access(o_sd[:1_000_000]) == 0).all())
            ex["sd_vector"] = {"ones": pos.numel(), "universe_log2": 40, "low_width": sd.low_width(),
                               "bits_per_one": sd.device_bytes() * 8 / pos.numel(), "build_s": sd_build,
                               "lane_kernels": {"rank": bool(sd.lane_kernels() & 1), "select_0": bool(sd.lane_kernels() & 2)},
                               "rank_1_Gq/s": nq_sd / ms_r / 1e6, "select_1_Gq/s": nq_sd / ms_s / 1e6,
                               "select_0_Gq/s": nq_sd / ms_z / 1e6, "queries": nq_sd}
            del sd, pos, xi, si, zi, zr, o_sd
        if "shapes" in extras and world == 1:
            # select_1 where the ones are NOT spread evenly (select_support_mcl's long blocks,
            # select_support_mcl.hpp:242-252): clustered in 1 % of the range, 2^20-bit dense/empty stripes, isolated
            # ones every 2^16 bits; plain, rrr_vector<63>, sd_vector; of_uniform = rate relative to the 50 % vector
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import select_shapes_bench
            ex["select_shapes"] = {"n_bits_log2": a.log_n, "queries": 10**8,
                                   "shapes": select_shapes_bench.run(pkg, a.log_n, 10**8, emit=lambda s: None, device=local)}
            pkg.set_timing(False)
        if "wt" in extras or "fm" in extras:
            torch.cuda.empty_cache()
            nt = a.text_mib << 20
            if a.text_file:
                # a real corpus (Pizza&Chili english.1GB the day it is on the box): first --text-mib MiB, zero bytes dropped
                # (SDSL's byte alphabet reserves 0 for the sentinel, construct.hpp:127-193)
                raw = np.fromfile(a.text_file, dtype=np.uint8, count=nt)
                text_h = np.ascontiguousarray(raw[raw != 0])
                nt = int(text_h.size)
                del raw
            else:
                text_h = pkg.english_text(nt, 1234)
            text = torch.from_numpy(text_h).to(dev)
            t0 = time.perf_counter()
            csa = pkg.csa_wt(text=text, device=local)
            build = time.perf_counter() - t0
            c4 = G.get("c4", {})
            c4ok = rank == 0 and not a.text_file and nt == (1 << c4.get("text_log", -1)) and "wt_rank" in c4
            wt = csa.wavelet_tree
            lens = torch.from_numpy(wt.code_lengths().astype(np.int64)).to(dev)
            fsteps = torch.from_numpy(wt.fused_steps().astype(np.int64)).to(dev)
            nq2 = min(nq, 100_000_000)
            # 8(d): i = mt19937_64(13) % (size() + 1), c = text[mt19937_64(14) % n] — symbols as the text distributes them
            gi = to_dev(pkg.rnd_positions(13, nq2, nt + 2, 0), dev)
            gc = text[to_dev(pkg.rnd_positions(14, nq2, nt, 0), dev)]
            out2 = torch.empty(nq2, dtype=torch.int64, device=dev)
            hbar = float(lens[gc.long()].double().mean())
            cnt_b = np.bincount(text_h, minlength=256)
            p_b = cnt_b[cnt_b > 0] / nt
            ex["text"] = {"bytes": nt, "kind": ("file " + os.path.basename(a.text_file)) if a.text_file else "English-class stand-in for Pizza&Chili english (sdsl_hip_util_english_text, seed 1234: Zipf "
                                               "words over a 65536-word vocabulary, mixed case, digits, punctuation, rare Latin-1 / control "
                                               "bytes; integer-only, reproduced bit for bit in the build container)",
                          "sigma": csa.sigma(), "H0": float(-(p_b * np.log2(p_b)).sum()), "index_build_s": build,
                          "mean_code_length_of_queried_symbols": hbar,
                          "wt_bits": wt.bv_size(), "index_bytes": csa.device_bytes()}
            del text_h
            ocsa = rcsa = None
            if rank == 0 and world == 1 and not a.no_cpu:
                import oracle_lib as ol
            if rank == 0 and world == 1 and not a.no_cpu and ol.have_ref():
                # CPU side, kind "reference": the index built on the GPU is written out as the bytes of
                # csa_wt<wt_huff<bit_vector, rank_support_v5<>>> (32 / 64) and LOADED BY THE REAL sdsl-lite — the unmodified
                # library then answers the same queries on it (a round trip of the whole index on every run)
                t0 = time.perf_counter()
                blob = csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL)
                t1 = time.perf_counter()
                rcsa = ol.RCsa(sdsl_bytes=blob)
                ex["text"]["sdsl_stream_bytes"] = len(blob)
                ex["text"]["gpu_serialize_s"] = t1 - t0
                ex["text"]["sdsl_load_s"] = time.perf_counter() - t1
                del blob
            elif rank == 0 and world == 1 and not a.no_cpu:
                # CPU side: the C restatement of wt_huff / backward_search (kind "port") over the SAME BWT, which is
                # reconstructed from the device index with wt[i] (access) so that no CPU suffix sorting is needed
                t0 = time.perf_counter()
                bwt = torch.empty(nt + 1, dtype=torch.uint8, device=dev)
                for s0 in range(0, nt + 1, 1 << 27):
                    e0 = min(nt + 1, s0 + (1 << 27))
                    wt.access(torch.arange(s0, e0, device=dev, dtype=torch.int64), bwt[s0:e0])
                ocsa = ol.OCsa(bwt=bwt.cpu().numpy())
                ex["text"]["cpu_index_build_s"] = time.perf_counter() - t0
                del bwt
            if "wt" in extras:
                _, ms = time_steps(lambda: wt.rank(gi, gc, out2), max(2, a.steps // 2), 1, barrier)
                alg = 17 + 80 * hbar
                steps = float(fsteps[gc.long()].double().mean())  # fused layout: depth in its own 8-ary tree
                lf = (17 + 128 * steps) * nq2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                rf, how = fused_frac("k_wt_rank_bytes_per_query", nq2, ms, lf)
                ex["wt_huff_rank"] = {"Gq/s": nq2 / ms / 1e6, "kernel_ms": ms, "queries": nq2,
                                      "reference_digest_match": digest_matches(out2, c4["wt_rank"])
                                      if c4ok and nq2 >= c4["wt_rank"]["n"] else None,
                                      "roofline_frac": rf, "roofline_frac_source": how,
                                      "survey_8d_model_frac": alg * nq2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_query": alg,
                                      "fused_steps_per_query": steps,
                                      "line_fetch_frac": lf,
                                      "note": FUSED_NOTE}
                if rcsa is not None:
                    cb = cpu_time(lambda i, c: rcsa.wt_rank(np.ascontiguousarray(i).view(np.uint64), c), [gi, gc], out2,
                                  a.cpu_seconds, 1e9, "(i,c) pairs, csa.wavelet_tree.rank of the real sdsl-lite on the "
                                  "index the GPU built and serialised")
                    cb.update(unit="Grank/s", kind="reference")
                    ex["wt_huff_rank"]["cpu_baseline"] = cb
                elif ocsa is not None:
                    owt = ocsa.wt()
                    cb = cpu_time(lambda i, c: owt.rank(np.ascontiguousarray(i).view(np.uint64), c), [gi, gc], out2,
                                  a.cpu_seconds, 1e9, "(i,c) pairs, wt_huff<bit_vector,rank_support_v5<>>::rank")
                    cb.update(unit="Grank/s", kind="port")
                    ex["wt_huff_rank"]["cpu_baseline"] = cb
            if "wt" in extras:
                # select(k, c) for symbols drawn from the text (the stream of rank) and k = 1 + mt19937_64(16) % occ(c): checked
                # through rank and against the real library's answers (golden_large.json, c4.wt_select)
                # (on the wavelet tree of the TEXT — same size and symbol distribution as the index's tree over the BWT — because that
                # is the sequence the reference's digest was made on: wt_huff<> constructed from the text by the real library)
                wt_t = pkg.wt_huff(text=text, device=local)
                occ_c = torch.bincount(text, minlength=256)[gc.long()]
                ks = 1 + to_dev(pkg.rnd_positions(16, nq2, 1 << 62, 0), dev) % occ_c
                _, ms = time_steps(lambda: wt_t.select(ks, gc, out2), 2, 1, barrier)
                chk = wt_t.rank(out2[:1_000_000], gc[:1_000_000])
                assert torch.equal(chk, ks[:1_000_000] - 1), "rank(select(k, c), c) != k - 1"
                ex["wt_huff_select"] = {"Gq/s": nq2 / ms / 1e6, "kernel_ms": ms, "queries": nq2,
                                        "path": "bucketed by place in symbol order, one lane per key (wt_sorted.hip)" if nq2 >= (1 << 23)
                                        else "direct fused select",
                                        "reference_digest_match": digest_matches(out2, c4["wt_select"])
                                        if c4ok and "wt_select" in c4 and nq2 >= c4["wt_select"]["n"] else None}
                del occ_c, ks, chk, wt_t
            if "fm" in extras:
                m = 20
                st = to_dev(pkg.rnd_positions(15, nq2, nt - m, 0), dev)  # 8(d): patterns cut at mt19937_64(15) % (n - m)
                pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
                sum_l = float(lens[pats.view(-1, m)[:, :m - 1].long()].double().sum(dim=1).mean())
                alg = 28 + 160 * sum_l
                sum_steps = float(fsteps[pats.view(-1, m)[:, :m - 1].long()].double().sum(dim=1).mean())

                def count_leg(variant, what):
                    """one timed leg of count(): >= 6 batches, every answer compared with the reference's digest; `roofline` from
                    the PMC collection of exactly this variant (tools/collect_profiles.sh -> profiles/pmc_latest.json), valid
                    only for these kernel sources"""
                    steps_ms = []
                    _, ms = time_steps(lambda: csa.count(pats, m, out2), max(6, a.steps // 2), 1, barrier, per_step=steps_ms)
                    assert bool((out2 >= 1).all()), "every pattern was cut from the text"
                    bpp = pmc_traffic("fm_count_%s_bytes_per_pattern" % variant)
                    rpp = pmc_traffic("fm_count_%s_requests_per_pattern" % variant)
                    return {"Mcount/s": nq2 / ms / 1e3, "kernel_ms": ms, "kernel_ms_per_batch": spread_of(steps_ms),
                            "spread": (max(steps_ms) - min(steps_ms)) / ms, "patterns": nq2, "m": m, "path": what,
                            "reference_digest_match": digest_matches(out2, c4["count"]) if c4ok and nq2 >= c4["count"]["n"] else None,
                            "index_bytes": csa.device_bytes(), "kmer_table": {"k": csa.kmer_table_depth(), "bytes": csa.kmer_table_bytes()},
                            "jump_depth": csa.jump_depth(),
                            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                         "traffic_bytes_per_pattern": bpp, "fabric_requests_per_pattern": rpp,
                                         "achieved": bpp * nq2 / (ms * 1e-3) / 1e9 if bpp else None,
                                         "frac": bpp * nq2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bpp else None,
                                         "valu_issue_share": pmc_traffic("fm_count_%s_valu_issue_share" % variant),
                                         "kernels": "k_fm_start + k_fm_count_flat + k_fm_verify2 (fm_count2.hip), summed",
                                         "source": "PMC (TCC_EA0_RDREQ/WRREQ, SQ_INSTS_VALU) of tools/fm_probe.py on this text and these "
                                                   "patterns, profiles/pmc_latest.json; null when it was not collected on these kernel sources"},
                            "survey_8d_model_frac": alg * nq2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_pattern": alg, "fused_steps_per_pattern_without_table": sum_steps, "note": FUSED_NOTE}

                verify_on = os.environ.get("SDSL_HIP_FM_VERIFY", "1") != "0"
                ex["fm_count"] = count_leg("default", "k-mer hash table (k = %d: one 128-byte bucket instead of k LF steps) -> flat search kernel "
                                           "until one suffix is left%s (fm_count2.hip); no sort, patterns in the caller's order"
                                           % (csa.kmer_table_depth(), " -> the remaining characters compared with the text at SA[l]: the whole "
                                              "suffix array and the text are resident" if verify_on else ""))
                ms = ex["fm_count"]["kernel_ms"]
                if rank == 0 and world == 1:
                    # end to end: patterns and answers in HOST memory (28 bytes per pattern over PCIe), pieces of 2^20 patterns
                    # on two streams (fm.hip: host_pipeline_bytes)
                    want_c = out2.cpu().numpy().view(np.uint64)
                    h_p = pats.cpu().numpy()
                    h_o = np.zeros(nq2, dtype=np.uint64)
                    csa.count(h_p, m, h_o)
                    ts = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        csa.count(h_p, m, h_o)
                        ts.append(time.perf_counter() - t0)
                    sec = sorted(ts)[1]
                    ex["fm_count"]["end_to_end"] = {"what": "sdsl_hip_fm_count_batch on HOST arrays (pageable), wall clock around the call",
                                                    "Mcount/s": nq2 / sec / 1e6, "seconds": spread_of(ts), "bytes_over_pcie_per_pattern": m + 8,
                                                    "pcie_GB/s_both_directions": (m + 8) * nq2 / sec / 1e9,
                                                    "same_answers": bool(np.array_equal(h_o, want_c))}
                    del h_p, h_o, want_c
                # the same index with the deepest table (HBM is there to be used: 32 bytes per distinct 8-mer)
                wt_budget = wt.device_bytes()
                csa.set_kmer_table(8, 64 << 30)
                ex["fm_count_kmer8"] = count_leg("k8", "as fm_count with the k-mer table at its deepest (k = %d)" % csa.kmer_table_depth())
                csa.set_kmer_table(8, wt_budget)  # back to the default depth
                if rcsa is not None:
                    cb = cpu_time(lambda p: rcsa.count_batch(p.reshape(-1), m), [pats.view(-1, m)], out2, a.cpu_seconds,
                                  1e6, "20-byte patterns, sdsl::count of the real sdsl-lite on the index the GPU built "
                                  "and serialised")
                    cb.update(unit="Mcount/s", kind="reference")
                    ex["fm_count"]["cpu_baseline"] = cb
                elif ocsa is not None:
                    cb = cpu_time(lambda p: ocsa.count_batch(p.reshape(-1), m), [pats.view(-1, m)], out2, a.cpu_seconds,
                                  1e6, "20-byte patterns, count(csa_wt<wt_huff<>>)")
                    cb.update(unit="Mcount/s", kind="port")
                    ex["fm_count"]["cpu_baseline"] = cb
                # locate / extract / SA access, on the whole suffix array the build left in HBM
                # and on SDSL's default samples (32 / 64) after drop_sa
                npat = 100_000
                lq, rq = csa.interval(pats[: npat * m], m)
                off, pos = csa.sa_range(lq, rq)
                _, ms = time_steps(lambda: csa.sa_range(lq, rq), 2, 1, barrier)
                ex["fm_locate_whole_sa"] = {"Gocc/s": pos.numel() / ms / 1e6, "ms": ms, "patterns": npat,
                                            "occurrences": pos.numel()}
                del off, pos
                sidx = torch.randint(0, nt + 1, (20_000_000,), device=dev, dtype=torch.int64, generator=gq)
                want = csa.sa(sidx)
                csa.drop_sa()
                _, ms = time_steps(lambda: csa.sa(sidx), 2, 1, barrier)
                assert torch.equal(csa.sa(sidx), want), "sampled SA walk != whole SA"
                ex["fm_sa_access_dens32"] = {"Msa/s": sidx.numel() / ms / 1e3, "ms": ms, "queries": sidx.numel()}
                # count() at the footprint of csa_wt<wt_huff<>, 32, 64> plus the k-mer table: no suffix array, no text, every
                # character after the table's k is an LF step (suffix_array_algorithm.hpp:228-248)
                ex["fm_count_sa_dropped"] = count_leg("dropped", "k-mer hash table (k = %d) -> flat search kernel over ALL remaining "
                                                      "characters; suffix array and text released (SDSL's default samples kept)"
                                                      % csa.kmer_table_depth())
                eb = torch.randint(0, nt - 64, (10_000_000,), device=dev, dtype=torch.int64, generator=gq)
                ee = eb + 63
                eoff, etxt = csa.extract(eb, ee)
                assert torch.equal(etxt.view(-1, 64)[:4096],
                                   text[(eb[:4096].view(-1, 1) + torch.arange(64, device=dev).view(1, 64))])
                _, ms = time_steps(lambda: csa.extract(eb, ee), 2, 1, barrier)
                ex["fm_extract_64B"] = {"GB/s": etxt.numel() / ms / 1e6, "ms": ms, "snippets": eb.numel()}
                del eoff, etxt, want
                # the compressed flavour csa_wt<wt_huff<rrr_vector<63>>> on the same patterns
                del csa, wt
                torch.cuda.empty_cache()
                t0 = time.perf_counter()
                crrr = pkg.csa_wt(text=text, device=local, rrr=True)
                rb = time.perf_counter() - t0
                nq3 = min(nq2, 20_000_000)
                _, ms = time_steps(lambda: crrr.count(pats[: nq3 * m], m, out2[:nq3]), 2, 1, barrier)
                ex["fm_count_rrr63"] = {"Mcount/s": nq3 / ms / 1e3, "kernel_ms": ms, "patterns": nq3, "m": m,
                                        "index_bytes": crrr.device_bytes(), "index_build_s": rb}
                del crrr
                # second data point: the sigma = 28 lowercase text round 1 reported on (an easier alphabet: shorter codes, a
                # deeper k-mer table)
                torch.cuda.empty_cache()
                t28 = synthetic_text(nt, 1234, dev)
                c28 = pkg.csa_wt(text=t28, device=local)
                st28 = torch.randint(0, nt - m, (nq2,), device=dev, generator=gq)
                p28 = t28[(st28.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
                _, ms = time_steps(lambda: c28.count(p28, m, out2), 2, 1, barrier)
                ex["fm_count_sigma28"] = {"Mcount/s": nq2 / ms / 1e3, "kernel_ms": ms, "patterns": nq2, "m": m,
                                          "sigma": c28.sigma(), "jump_depth": c28.jump_depth(),
                                          "text": "Zipf over a 4096-word lowercase vocabulary (round 1's stand-in)"}
                del c28, t28, p28, st28

        if "big" in extras and world == 1 and rank == 0:
            # An index of more than 2^32 symbols (opt-in: 172 GB of working memory in the suffix sorter): csa_wt from a
            # synthetic text of 2^32 + 777 symbols — 64-bit suffix sorter, fused lines with the 2^32-crossing list, SA / ISA
            # samples instead of the whole array (answers checked by tests/test_gpu_beyond_2_32.py).
            torch.cuda.empty_cache()
            nb, sg = (1 << 32) + 777, 40
            gb = torch.Generator(device=dev).manual_seed(1)
            tb = torch.empty(nb, dtype=torch.uint8, device=dev)
            for a0 in range(0, nb, 1 << 28):
                b0 = min(nb, a0 + (1 << 28))
                uu = torch.rand(b0 - a0, device=dev, generator=gb)
                tb[a0:b0] = (1 + (uu * uu * sg).to(torch.int64).clamp_(max=sg - 1)).to(torch.uint8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cb = pkg.csa_wt(text=tb, device=local)
            torch.cuda.synchronize()
            build_s = time.perf_counter() - t0
            npb, mb = 10_000_000, 20
            stb = torch.randint(0, nb - mb, (npb,), device=dev, dtype=torch.int64, generator=gb)
            pb = tb[(stb.view(-1, 1) + torch.arange(mb, device=dev).view(1, mb)).reshape(-1)].contiguous()
            ob = torch.empty(npb, dtype=torch.int64, device=dev)
            _, ms_c = time_steps(lambda: cb.count(pb, mb, ob), 2, 1, barrier)
            nrb = 100_000_000
            ib = torch.randint(0, nb + 1, (nrb,), device=dev, dtype=torch.int64, generator=gb)
            sb = tb[torch.randint(0, nb, (nrb,), device=dev, dtype=torch.int64, generator=gb)]
            orb = torch.empty(nrb, dtype=torch.int64, device=dev)
            wtb = cb.wavelet_tree
            _, ms_r = time_steps(lambda: wtb.rank(ib, sb, out=orb), 2, 1, barrier)
            ex["beyond_2_32"] = {"symbols": nb, "sigma": sg, "build_from_text_s": build_s, "resident_GB": cb.device_bytes() / 1e9,
                                 "sampling": list(cb.sampling()), "count_Mcount/s": npb / ms_c / 1e3, "patterns": npb, "m": mb,
                                 "every_pattern_found": bool((ob >= 1).all()), "wt_rank_Gq/s": nrb / ms_r / 1e6,
                                 "note": "fused lines with the 2^32-crossing list; the flat count kernel and its k-mer table are 32-bit and not used"}
            del wtb, cb, tb, pb, ob, ib, sb, orb, stb
            torch.cuda.empty_cache()

        if "fm_sharded" in extras and world > 1:
            # the headline queries as a ROOT-OWNED batch (the end-to-end column): rank 0 holds all
            # world * nq positions, scatter -> rank kernel -> gather in eight pipelined pieces.  16 bytes per query cross
            # xGMI, so this column is link-bound by construction; the resident-shard figure above is the kernel column.
            stage0 = (lambda t: t) if a.backend == "nccl" else (lambda t: t.cpu())
            nro = min(nq, 250_000_000) * world
            allq = stage0(torch.randint(0, n_bits + 1, (nro,), device=dev, dtype=torch.int64, generator=gq)) if rank == 0 \
                else stage0(torch.empty(1, dtype=torch.int64, device=dev))

            def local_rank(x):
                xd = x.to(dev)
                return stage0(bv.rank(xd, 1))

            pkg.dist.sharded_query(local_rank, (allq,), nro, chunks=8)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            ro = pkg.dist.sharded_query(local_rank, (allq,), nro, chunks=8)
            torch.cuda.synchronize()
            barrier()
            dt = pkg.dist.max_over_ranks(time.perf_counter() - t0, comm_dev)
            ex["rank_root_owned_batch"] = {"Grank/s": nro / dt / 1e9, "ms": dt * 1e3, "queries": nro, "pieces": 8,
                                           "bytes_over_links_per_query": 16}
            if rank == 0:
                chk = bv.rank(allq[:1_000_000].to(dev), 1)
                ex["rank_root_owned_batch"]["matches_local"] = bool(torch.equal(stage0(chk), ro[:1_000_000]))
            del allq, ro
            # configs[4]: count() on a 1 GiB text, 10^8 20-byte patterns sharded across the ranks (strong scaling).
            # The FM-index is replicated (every rank builds it from the same text on its own GPU); (a) resident
            # shards: every rank answers its slice of the batch, no collective; (b) root-owned batch: rank 0 holds the
            # whole batch, one scatter + one gather over RCCL/xGMI around the same local call (dist.sharded_query).
            import torch.distributed as dist
            torch.cuda.empty_cache()
            nt = a.text_mib << 20
            stage = (lambda t: t) if a.backend == "nccl" else (lambda t: t.cpu())
            # load time: rank 0 owns the text, one broadcast hands it to every rank, every rank lays out its own index
            t0 = time.perf_counter()
            text = pkg.dist.replicate(stage(torch.from_numpy(pkg.english_text(nt, 1234)).to(dev)) if rank == 0 else None,
                                      stage(torch.empty(0, dtype=torch.uint8, device=dev))).to(dev)
            torch.cuda.synchronize()
            bcast = time.perf_counter() - t0
            t0 = time.perf_counter()
            csa = pkg.csa_wt(text=text, device=local)
            build = time.perf_counter() - t0
            m, total = 20, min(int(a.queries) // 10, 100_000_000)
            gp = torch.Generator(device=dev).manual_seed(99)  # the same batch on every rank; each takes its slice
            st = torch.randint(0, nt - m, (total,), device=dev, generator=gp)
            lo, hi = pkg.dist.shard_bounds(total, world, rank)
            mine = text[(st[lo:hi].view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
            res = torch.empty(hi - lo, dtype=torch.int64, device=dev)
            wall_s, _ = time_steps(lambda: csa.count(mine, m, res), 3, 1, barrier)
            wall_s = pkg.dist.max_over_ranks(wall_s, comm_dev)
            ok = bool((res >= 1).all())
            fs = {"patterns_total": total, "m": m, "text_bytes": nt, "index_build_s": build, "index_bytes_per_gpu": csa.device_bytes(),
                  "kmer_table_k": csa.kmer_table_depth(),
                  "text_broadcast_s": bcast,
                  "resident_shards": {"Mcount/s": total * 3 / wall_s / 1e6, "ms_per_batch": wall_s / 3 * 1e3,
                                      "all_patterns_found": ok, "scaling": "strong"}}
            # (b) root-owned batch
            if rank == 0:
                allp = stage(text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous())
            else:
                allp = stage(torch.empty(1, dtype=torch.uint8, device=dev))

            def local_count(p):
                r = torch.empty(p.numel() // m, dtype=torch.int64, device=dev)
                csa.count(p.to(dev), m, r)
                return stage(r)

            full = pkg.dist.sharded_query(local_count, (allp,), total, widths=(m,))  # warm-up + check
            if rank == 0:
                fs["root_owned_batch_matches"] = bool(torch.equal(full[lo:hi].to(dev), res))
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            full = pkg.dist.sharded_query(local_count, (allp,), total, widths=(m,))
            torch.cuda.synchronize()
            barrier()
            dt = pkg.dist.max_over_ranks(time.perf_counter() - t0, comm_dev)
            fs["root_owned_batch"] = {"Mcount/s": total / dt / 1e6, "ms_per_batch": dt * 1e3,
                                      "collectives": "1 scatter (patterns) + 1 gather (counts)"}
            # the same in four pipelined pieces: scatter of piece c+1 and gather of piece c-1 overlap the kernels of piece c
            t0 = time.perf_counter()
            full4 = pkg.dist.sharded_query(local_count, (allp,), total, widths=(m,), chunks=4)
            torch.cuda.synchronize()
            barrier()
            dt = pkg.dist.max_over_ranks(time.perf_counter() - t0, comm_dev)
            fs["root_owned_batch_pipelined"] = {"Mcount/s": total / dt / 1e6, "ms_per_batch": dt * 1e3, "pieces": 4,
                                                "matches": bool(torch.equal(full, full4)) if rank == 0 else None}
            ex["fm_count_sharded"] = fs
            del csa, text
    except Exception as e:  # the secondary measurements must never cost the headline line
        ex["error"] = f"{type(e).__name__}: {e}"
    if ex:
        result["extras"] = ex
    # the second half of BASELINE.json's metric ("... + Mcount/s english.1GB FM-index"), surfaced next to the headline
    if "fm_count" in ex:
        result["secondary"] = {"metric": "Mcount/s, count() of 20-byte patterns, FM-index of a %d MiB text" % a.text_mib,
                               "value": ex["fm_count"]["Mcount/s"], "unit": "Mcount/s", "n_gpus": 1,
                               "source": "extras.fm_count"}
    elif "fm_count_sharded" in ex:
        result["secondary"] = {"metric": "Mcount/s, count() of 20-byte patterns, FM-index of a %d MiB text" % a.text_mib,
                               "value": ex["fm_count_sharded"]["resident_shards"]["Mcount/s"], "unit": "Mcount/s",
                               "n_gpus": world, "scaling": "strong (one batch of %d patterns split over the ranks)"
                                                           % ex["fm_count_sharded"]["patterns_total"],
                               "source": "extras.fm_count_sharded.resident_shards"}
    if world > 1 and "group" in extras and a.backend == "nccl":
        # the other driver of 8(e): rank 0 alone drives all the GPUs through the C ABI's device group while the other ranks (their
        # memory released) wait at the barrier — both drivers' columns in one line
        try:
            import torch.distributed as dist
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
            if rank == 0:
                ex["device_group"] = group_leg(pkg, a, world, n_bits, min(nq, 250_000_000), max(3, a.steps // 2), 1)
                torch.cuda.set_device(local)
            dist.barrier(group=cpu_group)
        except Exception as e:
            ex["device_group"] = {"error": f"{type(e).__name__}: {e}"}
        result["extras"] = ex
    if world > 1 and "rank_root_owned_batch" in ex:
        # both columns of the multi-GPU report, side by side
        result["scaling_columns"] = {"kernel_only_resident_shards_Grank/s": value,
                                     "device_group": ex.get("device_group"),
                                     "end_to_end_root_owned_batch_Grank/s": ex["rank_root_owned_batch"]["Grank/s"],
                                     "note": "resident shards: every rank answers its own HBM-resident shard, no collective in the "
                                             "timed region; root-owned: rank 0 holds the batch, scatter -> kernels -> gather over "
                                             "RCCL in 8 pipelined pieces, 16 bytes per query cross xGMI"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
