"""bench_common.py — helpers shared by bench.py (the headline) and bench_extras.py (the secondary legs): timing with HIP events,
the committed-PMC lookup, the digest check against the real library's answers, host facts."""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy achieves
FUSED_NOTE = ("survey_8d_model_frac prices the query as SURVEY.md 8(d) does (80 bytes per tree level: the REFERENCE's level-by-level walk); "
              "the kernel walks the fused layout (one 128-byte line per level of its own 16-ary tree) and does not move those bytes, so that "
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


def pmc_roofline(key, units, ms):
    """roofline block of a secondary kernel from the committed PMC collection (tools/kernel_probe.py under tools/prof.sh ->
    profiles/pmc_latest.json: <key>_bytes_per_unit / _requests_per_unit / _valu_issue_share), priced at THIS run's rate: fabric bytes the
    kernels move per unit x units / measured time, over the HBM peak.  All None when the collection is not of these kernel sources."""
    b = pmc_traffic(key + "_bytes_per_unit")
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic_bytes_per_unit": b,
            "fabric_requests_per_unit": pmc_traffic(key + "_requests_per_unit"),
            "achieved": b * units / (ms * 1e-3) / 1e9 if b else None, "frac": b * units / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if b else None,
            "valu_issue_share": pmc_traffic(key + "_valu_issue_share"), "kernels": pmc_traffic(key + "_kernels"),
            "traffic_source": "committed PMC collection (profiles/pmc_latest.json), measured fabric bytes — these kernels have no closed-form "
                              "byte model: walk lengths are geometric, lines per tree step depend on the symbol"}


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


def digests_match(ans_dev, block, key):
    """block[key] (the first n answers) AND block[key + "_strided"] (every stride-th answer of the first `count`: the whole BASELINE batch)
    when the batch is long enough for them; None when neither applies"""
    res = None
    w = block.get(key)
    if w and ans_dev.numel() >= w["n"]:
        res = digest_matches(ans_dev, w)
    ws = block.get(key + "_strided")
    if ws and ans_dev.numel() >= ws["count"] and res is not False:
        res = digest_matches(ans_dev[: ws["count"]: ws["stride"]].contiguous(), ws)
    return res


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

