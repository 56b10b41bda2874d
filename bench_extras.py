"""bench_extras.py — the secondary legs of bench.py (`--extras ...`).  None of them enters the timed region of the headline; each leg
writes its block into the SIDECAR (bench_extras.json next to bench.py, rewritten after every leg so that a leg that dies leaves the
finished ones behind), never into the one JSON line bench.py prints.  A leg is a function of the run's context `c` (class Ctx)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from bench_common import (ALG_BYTES, FUSED_NOTE, HBM_PEAK_GBS, ROOT, cpu_time, digest_matches, digests_match, fused_frac, golden, pmc_roofline, pmc_traffic,
                          spread_of, synthetic_text, time_steps, to_dev)

SIDECAR = os.path.join(ROOT, "bench_extras.json")


class Ctx:
    """What a leg may use: the parsed arguments, the package, this rank's device, the headline's index / positions / answers (legs that
    need the memory release them: c.bv = None), and `ex`, the sidecar's content."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.ex = {}


def write_sidecar(c, path=None):
    """atomic rewrite of the sidecar (rank 0 only)"""
    if c.rank != 0:
        return
    path = path or c.sidecar
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump({"bench_argv": sys.argv[1:], "n_gpus": c.world, "legs_done": list(c.done), "extras": c.ex}, f)
    os.replace(tmp, path)


# leg name -> (function, runs on several ranks?, runs on one rank?)
def legs_table():
    return {"e2e": (leg_e2e, False), "sweep": (leg_sweep, False), "select": (leg_select, True), "rrr": (leg_rrr, True), "sd": (leg_sd, True),
            "shapes": (leg_shapes, False), "text": (leg_text, True), "rep": (leg_repetitive, False), "big": (leg_big, False), "fm_sharded": (leg_sharded, True)}


def run_extras(c):
    """every asked-for leg in the fixed order below; an exception ends the legs (later ones would run on a device in an unknown state) and
    is recorded, it never costs the headline"""
    T = legs_table()
    want = list(c.extras)
    if "wt" in want or "fm" in want:
        want.append("text")
    if "fm" in want:
        want.append("rep")
    c.done = []
    for name in ("e2e", "sweep", "select", "rrr", "sd", "shapes", "text", "rep", "big", "fm_sharded"):
        if name not in want:
            continue
        fn, multi = T[name]
        if c.world > 1 and not multi:
            continue
        if name == "fm_sharded" and c.world == 1:
            continue
        if name in ("e2e", "sweep", "shapes", "big", "rep") and c.rank != 0:
            continue
        t0 = time.perf_counter()
        try:
            fn(c)
        except Exception as e:
            c.ex["error"] = f"{name}: {type(e).__name__}: {e}"
            write_sidecar(c)
            break
        c.ex.setdefault("leg_seconds", {})[name] = round(time.perf_counter() - t0, 2)
        c.done.append(name)
        if name == "select":
            c.words = None  # only the select leg's CPU baseline reads the vector's words again
        write_sidecar(c)
    return c.ex


def leg_e2e(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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
    ex["end_to_end"] = {"what": "sdsl_hip_bv_rank_batch on HOST arrays (positions in, answers out), wall clock around the call",
                            "queries": ne, "bytes_over_pcie_per_query": 16, **legs,
                            "pcie_note": "PCIe 5.0 x16: 64 GB/s per direction on paper, ~55 achievable; the kernel-only rate is `value`"}
    del h_idx, h_out, want_e
    if kind == "pinned":
        del t_idx, t_out


def leg_sweep(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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


def leg_select(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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
        ex["select_1"]["reference_digest_match"] = digests_match(out, G["c2"], "select_1")
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


def leg_rrr(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
    del bv
    c.bv = None
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
    ex["rrr63_rank_1"]["traffic_frac"] = ex["rrr63_rank_1"]["fabric_traffic"]["frac_of_hbm_peak"]  # (what the line's summary carries)
    ex["rrr63_rank_1"]["survey_8d_model_frac"] = ex["rrr63_rank_1"].pop("roofline_frac")          # (> 1 is possible: not a roofline fraction)
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
        ex["rrr63_rank_1"]["reference_digest_match"] = digests_match(out, c3, "rank_1") and rv.ones() == c3["ones"]
    si = to_dev(pkg.rnd_positions(11, nq, rv.ones(), 1), dev)
    _, ms = time_steps(lambda: rv.select(si, 1, out), max(2, a.steps // 2), 1, barrier)
    ex["rrr63_select_1"] = {"Gq/s": nq / ms / 1e6, "kernel_ms": ms,
                            "roofline_frac": ALG_BYTES["rrr"] * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    bq = pmc_traffic("rrr_select_bucketed_bytes_per_query")
    ex["rrr63_select_1"]["fabric_traffic"] = {"bytes_per_query": bq,
                                              "frac_of_hbm_peak": bq * nq / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bq else None}
    ex["rrr63_select_1"]["traffic_frac"] = ex["rrr63_select_1"]["fabric_traffic"]["frac_of_hbm_peak"]
    ex["rrr63_select_1"]["survey_8d_model_frac"] = ex["rrr63_select_1"].pop("roofline_frac")
    pkg.set_option("rrr_sorted", 0)
    out_d = torch.empty_like(out)
    _, ms_d = time_steps(lambda: rv.select(si, 1, out_d), 2, 1, barrier)
    pkg.set_option("rrr_sorted", -1)
    ex["rrr63_select_1"]["direct_kernel"] = {"Gq/s": nq / ms_d / 1e6, "kernel_ms": ms_d, "same_answers": bool(torch.equal(out, out_d)),
                                             "roofline_frac": ALG_BYTES["rrr"] * nq / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del out_d
    if c3ok:
        ex["rrr63_select_1"]["reference_digest_match"] = digests_match(out, c3, "select_1")
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


def leg_sd(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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
                       "select_0_Gq/s": nq_sd / ms_z / 1e6, "queries": nq_sd,
                       "roofline": {"rank_1": pmc_roofline("sd_rank", nq_sd, ms_r), "select_1": pmc_roofline("sd_select1", nq_sd, ms_s),
                                    "select_0": pmc_roofline("sd_select0", nq_sd, ms_z)}}
    del sd, pos, xi, si, zi, zr, o_sd


def leg_shapes(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
    # select_1 where the ones are NOT spread evenly (select_support_mcl's long blocks,
    # select_support_mcl.hpp:242-252): clustered in 1 % of the range, 2^20-bit dense/empty stripes, isolated
    # ones every 2^16 bits; plain, rrr_vector<63>, sd_vector; of_uniform = rate relative to the 50 % vector
    torch.cuda.empty_cache()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import select_shapes_bench
    ex["select_shapes"] = {"n_bits_log2": a.log_n, "queries": 10**8,
                           "shapes": select_shapes_bench.run(pkg, a.log_n, 10**8, emit=lambda s: None, device=local)}
    pkg.set_timing(False)


def leg_text(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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
    if a.text_file:
        # a text of the user's: digests from `make_golden_large.py c4 c4sel c4s --text-file FILE`, if they were made (same bytes used)
        gf = os.path.join(ROOT, "tests", "golden", "golden_large_%s.json" % os.path.basename(a.text_file))
        c4 = json.load(open(gf)).get("c4", {}) if os.path.exists(gf) else {}
        c4ok = rank == 0 and c4.get("text_bytes") == nt and "wt_rank" in c4
    else:
        c4ok = rank == 0 and nt == (1 << c4.get("text_log", -1)) and "wt_rank" in c4
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
    if "wt" in c.extras:
        _, ms = time_steps(lambda: wt.rank(gi, gc, out2), max(2, a.steps // 2), 1, barrier)
        alg = 17 + 80 * hbar
        steps = float(fsteps[gc.long()].double().mean())  # fused layout: depth in its own 16-ary tree
        lf = (17 + 128 * steps) * nq2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        rf, how = fused_frac("k_wt_rank_bytes_per_query", nq2, ms, lf)
        ex["wt_huff_rank"] = {"Gq/s": nq2 / ms / 1e6, "kernel_ms": ms, "queries": nq2,
                              "reference_digest_match": digests_match(out2, c4, "wt_rank") if c4ok else None,
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
    if "wt" in c.extras:
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
                                if c4ok and "wt_select" in c4 and nq2 >= c4["wt_select"]["n"] else None,
                                "roofline": pmc_roofline("wt_select", nq2, ms)}
        # the same tree at the reference's footprint: SDSL's binary levels released (sdsl_hip_wt_release_binary_levels), rank and select on
        # the fused lines alone; bytes against the stream wt_huff<bit_vector, rank_support_v5<>, select_support_mcl<>, ...> serialises to
        full_b = wt_t.device_bytes()
        want_sel = out2.clone()
        wt_stream = len(wt_t.serialize(pkg.capi.LAYOUT_BV_MCL))
        wt_t.release_binary_levels()
        _, ms_s = time_steps(lambda: wt_t.select(ks, gc, out2), 2, 1, barrier)
        same_sel = bool(torch.equal(out2, want_sel))
        _, ms_r = time_steps(lambda: wt_t.rank(gi, gc, out2), 2, 1, barrier)
        ex["wt_huff_fused_lines_only"] = {"device_bytes": wt_t.device_bytes(), "device_bytes_with_binary_levels": full_b,
                                          "sdsl_stream_bytes": wt_stream, "x_sdsl_stream_bytes": wt_t.device_bytes() / wt_stream,
                                          "rank_Gq/s": nq2 / ms_r / 1e6, "select_Gq/s": nq2 / ms_s / 1e6, "select_same_answers": same_sel}
        del occ_c, ks, chk, wt_t, want_sel
    if "fm" in c.extras:
        m = 20
        st = to_dev(pkg.rnd_positions(15, nq2, nt - m, 0), dev)  # 8(d): patterns cut at mt19937_64(15) % (n - m)
        pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
        sum_l = float(lens[pats.view(-1, m)[:, :m - 1].long()].double().sum(dim=1).mean())
        alg = 28 + 160 * sum_l
        sum_steps = float(fsteps[pats.view(-1, m)[:, :m - 1].long()].double().sum(dim=1).mean())

        def route_of(ix):
            """which road count() takes on this index, in a few words (the line's secondary.route)"""
            p = ix.footprint_parts()
            return ("k-mer table k=%d -> flat fused-tree search%s" % (ix.kmer_table_depth(), " -> text comparison at <= 8 suffixes" if p["suffix_array"] and verify_on else "")
                    + ("; tree: fused lines only" if not p["wt_binary_levels"] else ""))

        verify_on = os.environ.get("SDSL_HIP_FM_VERIFY", "1") != "0"

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
                    "spread": (max(steps_ms) - min(steps_ms)) / ms, "patterns": nq2, "m": m, "path": what, "route": route_of(csa),
                    "resident_bytes_by_part": csa.footprint_parts(),
                    "reference_digest_match": digests_match(out2, c4, "count") if c4ok else None,
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

        ex["fm_count"] = count_leg("default", "k-mer hash table (k = %d: one 128-byte bucket instead of k LF steps) -> flat search kernel "
                                   "until at most eight suffixes are left%s (fm_count2.hip); no sort, patterns in the caller's order"
                                   % (csa.kmer_table_depth(), " -> the remaining characters compared with the text in front of each of them: the whole "
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
        eb = torch.randint(0, nt - 64, (10_000_000,), device=dev, dtype=torch.int64, generator=gq)
        ee = eb + 63
        _, ms_t = time_steps(lambda: csa.extract(eb, ee), 2, 1, barrier)  # text still resident: a copy (locate.hip: k_fm_extract_copy)
        # the sampling densities are template parameters of the reference's type (csa_wt.hpp:51-57): one denser point first
        # (csa_wt<..., 8, 16>), then the suffix array is brought back and SDSL's defaults 32 / 64 are taken
        csa.drop_sa(8, 16)
        _, ms8 = time_steps(lambda: csa.sa(sidx), 2, 1, barrier)
        assert torch.equal(csa.sa(sidx), want), "sampled SA walk (dens 8) != whole SA"
        dense = {"Msa/s": sidx.numel() / ms8 / 1e3, "ms": ms8, "sa_dens": 8, "isa_dens": 16, "index_bytes": csa.device_bytes()}
        csa.restore_suffix_array()
        csa.drop_sa(32, 64)
        _, ms = time_steps(lambda: csa.sa(sidx), 2, 1, barrier)
        assert torch.equal(csa.sa(sidx), want), "sampled SA walk != whole SA"
        ex["fm_sa_access_dens32"] = {"Msa/s": sidx.numel() / ms / 1e3, "ms": ms, "queries": sidx.numel(),
                                     "roofline": pmc_roofline("fm_sa", sidx.numel(), ms), "at_dens_8_16": dense}
        # locate on the samples: the occurrences of 10^7 patterns, every one an LF walk to the next sampled suffix (csa_wt.hpp:363-381)
        n_loc = min(nq2, 10_000_000)
        lq6, rq6 = csa.interval(pats[: n_loc * m], m)
        off6, pos6 = csa.sa_range(lq6, rq6)
        _, ms6 = time_steps(lambda: csa.sa_range(lq6, rq6), 2, 1, barrier)
        ex["fm_locate_dens32"] = {"Gocc/s": pos6.numel() / ms6 / 1e6, "ms": ms6, "patterns": n_loc, "occurrences": pos6.numel(),
                                  "roofline": pmc_roofline("fm_locate", pos6.numel(), ms6)}
        del lq6, rq6, off6, pos6
        # count() at the footprint of csa_wt<wt_huff<>, 32, 64> plus the k-mer table: no suffix array, no text, every
        # character after the table's k is an LF step (suffix_array_algorithm.hpp:228-248)
        ex["fm_count_sa_dropped"] = count_leg("dropped", "k-mer hash table (k = %d) -> flat search kernel over ALL remaining "
                                              "characters; suffix array and text released (SDSL's default samples kept)"
                                              % csa.kmer_table_depth())
        eoff, etxt = csa.extract(eb, ee)
        assert torch.equal(etxt.view(-1, 64)[:4096],
                           text[(eb[:4096].view(-1, 1) + torch.arange(64, device=dev).view(1, 64))])
        _, ms = time_steps(lambda: csa.extract(eb, ee), 2, 1, barrier)
        ex["fm_extract_64B"] = {"GB/s": etxt.numel() / ms / 1e6, "ms": ms, "snippets": eb.numel(),
                                "roofline": pmc_roofline("fm_extract", etxt.numel(), ms),
                                "with_text_resident_GB/s": etxt.numel() / ms_t / 1e6}
        # long ranges: a range is cut at the ISA samples and every piece walks exactly its own symbols (a 64-byte snippet also walks the
        # 32 steps from the sample behind it): 256 ranges of 1 MiB, compared with the text
        rl = min(1 << 20, max(64, nt // 1024))  # (1 MiB each on the bench text; a small text of the user's: shorter ones)
        lb = torch.arange(256, device=dev, dtype=torch.int64) * ((nt - rl - 18) // 256) + 17
        le = lb + rl - 1
        loff, ltxt = csa.extract(lb, le)
        assert torch.equal(ltxt[:rl], text[17:17 + rl]) and torch.equal(ltxt[-rl:], text[int(lb[-1]):int(le[-1]) + 1])
        _, ms_l = time_steps(lambda: csa.extract(lb, le), 2, 1, barrier)
        ex["fm_extract_64B"]["long_ranges_GB/s"] = ltxt.numel() / ms_l / 1e6
        ex["fm_extract_64B"]["long_ranges"] = "256 x %d bytes, cut at the ISA samples into pieces walked in parallel" % rl
        del eoff, etxt, want, loff, ltxt
        # count() against resident bytes: the index gives HBM back step by step (sdsl_hip_fm_set_footprint) down to the reference's
        # own footprint — csa_wt<wt_huff<>, 32, 64> of this text serialises to `sdsl_stream_bytes` (csa_wt.hpp:389-402) — and count()
        # is timed and digest-checked at every step.  "lean" = 1.5 x the reference's bytes: fused tree lines, 32-bit samples, the
        # k-mer table the rest of the budget holds; that row carries its own PMC roofline (variant "lean" of tools/fm_probe.py)
        sdsl_bytes = ex["text"].get("sdsl_stream_bytes")
        if not sdsl_bytes:
            sdsl_bytes = len(csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL))
            ex["text"]["sdsl_stream_bytes"] = sdsl_bytes
        rows = [("k8", ex["fm_count_kmer8"]), ("default", ex["fm_count"]), ("sa_dropped", ex["fm_count_sa_dropped"])]
        for name, budget in (("lean_k6", 2.75 * sdsl_bytes), ("lean", 1.5 * sdsl_bytes), ("lean_1.3x", 1.3 * sdsl_bytes), ("floor", 0)):
            try:
                if name == "floor":
                    p_ = csa.footprint_parts()
                    budget = sum(v for k_, v in p_.items() if k_ != "kmer_table")
                csa.set_footprint(int(budget))
            except Exception as e_:
                ex.setdefault("fm_footprint_errors", {})[name] = str(e_)
                continue
            leg = count_leg("lean" if name == "lean" else "none", "sdsl_hip_fm_set_footprint(%d): %s" % (int(budget), route_of(csa)))
            leg["x_sdsl_stream_bytes"] = leg["index_bytes"] / sdsl_bytes
            rows.append((name, leg))
            if name == "lean":
                ex["fm_count_lean"] = leg
        ex["fm_count_vs_resident_bytes"] = {
            "sdsl_stream_bytes": sdsl_bytes, "what": "count() of the same 20-byte patterns on the same index at shrinking footprints; every row "
            "compared with the real library's digest",
            "rows": [{"name": nm, "index_bytes": lg["index_bytes"], "x_sdsl_stream_bytes": lg["index_bytes"] / sdsl_bytes,
                      "Mcount/s": lg["Mcount/s"], "kmer_k": lg["kmer_table"]["k"], "route": lg["route"],
                      "reference_digest_match": lg["reference_digest_match"],
                      "roofline_frac": lg["roofline"]["frac"]} for nm, lg in sorted(rows, key=lambda r: -r[1]["index_bytes"])]}
        # csa[i] and extract on the lean index walk the same fused lines: one figure each, beside the ones above
        _, ms = time_steps(lambda: csa.sa(sidx), 2, 1, barrier)
        ex["fm_sa_access_dens32"]["lean_index_Msa/s"] = sidx.numel() / ms / 1e3
        # the compressed flavour csa_wt<wt_huff<rrr_vector<63>>> on the same patterns
        nq3 = min(nq2, 20_000_000)
        csa.count(pats[: nq3 * m], m, out2[:nq3])
        plain_counts = out2[:nq3].clone()  # (digest-checked above: every leg of the plain index compared all nq2 with the real library's)
        del csa, wt
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        crrr = pkg.csa_wt(text=text, device=local, rrr=True)
        rb = time.perf_counter() - t0
        _, ms = time_steps(lambda: crrr.count(pats[: nq3 * m], m, out2[:nq3]), 2, 1, barrier)
        ex["fm_count_rrr63"] = {"Mcount/s": nq3 / ms / 1e3, "kernel_ms": ms, "patterns": nq3, "m": m,
                                "index_bytes": crrr.device_bytes(), "index_build_s": rb, "roofline": pmc_roofline("fm_count_rrr63", nq3, ms),
                                "same_answers_as_plain_index": bool(torch.equal(out2[:nq3], plain_counts))}
        # ... and at a compressed size (round 6): suffix array and text -> SDSL's samples, the k-mer table the budget holds — 1.5 x the bytes
        # the real library's csa_wt<wt_huff<rrr_vector<63>>, 32, 64> serialises to (our serialiser writes its stream byte for byte)
        try:
            rrr_stream = len(crrr.serialize(32, 64, pkg.capi.LAYOUT_RRR63))
            crrr.set_footprint(int(1.5 * rrr_stream))
            _, ms = time_steps(lambda: crrr.count(pats[: nq3 * m], m, out2[:nq3]), 2, 1, barrier)
            lean = {"Mcount/s": nq3 / ms / 1e3, "kernel_ms": ms, "patterns": nq3, "m": m, "index_bytes": crrr.device_bytes(),
                    "sdsl_stream_bytes": rrr_stream, "x_sdsl_stream_bytes": crrr.device_bytes() / rrr_stream,
                    "resident_bytes_by_part": crrr.footprint_parts(), "kmer_table": {"k": crrr.kmer_table_depth(), "bytes": crrr.kmer_table_bytes()},
                    "roofline": pmc_roofline("fm_count_rrr63_lean", nq3, ms)}
            lean["same_answers_as_plain_index"] = bool(torch.equal(out2[:nq3], plain_counts))
            ex["fm_count_rrr63_lean"] = lean
            ex["fm_count_rrr63"]["sdsl_stream_bytes"] = rrr_stream
            ex["fm_count_rrr63"]["x_sdsl_stream_bytes"] = ex["fm_count_rrr63"]["index_bytes"] / rrr_stream
        except Exception as e_:
            ex.setdefault("fm_footprint_errors", {})["rrr63_lean"] = str(e_)
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


def leg_repetitive(c):
    """configs[3] / [4] on the REPETITIVE stand-in (round 6; VERDICT r05: the independent blocks of the first stand-in give a 20-byte pattern
    1.017 occurrences on average, and count()'s default route — ... -> text comparison at one suffix — lives off that; the real english.1GB
    concatenates books that repeat whole passages).  english_text_repetitive(2^30, 1234, 30): 30 % of the 64 KiB blocks are rotated copies of
    earlier ones; patterns drawn from the text as genpatterns.c:183-203 draws them occur 2.4 times on average, 46 % of them more than
    once.  wt.rank and count() at the default footprint and at 1.5 x the reference's stream, all 10^8 answers against the digests of the
    real library (tests/golden/make_golden_large.py c4r)."""
    a, pkg, dev, local, rank, barrier, G, ex = c.a, c.pkg, c.dev, c.local, c.rank, c.barrier, c.G, c.ex
    if a.text_file:
        return
    torch.cuda.empty_cache()
    nt = a.text_mib << 20
    c4r = G.get("c4r", {})
    ok = nt == (1 << c4r.get("text_log", -1)) and "count" in c4r
    text = torch.from_numpy(pkg.english_text_repetitive(nt, 1234, 30)).to(dev)
    t0 = time.perf_counter()
    csa = pkg.csa_wt(text=text, device=local)
    build = time.perf_counter() - t0
    nq2 = min(c.nq, 100_000_000)
    m = 20
    gi = to_dev(pkg.rnd_positions(13, nq2, nt + 2, 0), dev)
    gc = text[to_dev(pkg.rnd_positions(14, nq2, nt, 0), dev)]
    out2 = torch.empty(nq2, dtype=torch.int64, device=dev)
    wt = csa.wavelet_tree
    _, ms = time_steps(lambda: wt.rank(gi, gc, out=out2), 3, 1, barrier)
    block = {"text": "english_text_repetitive(%d, 1234, 30): 30 %% of the 64 KiB blocks are rotated copies of earlier blocks" % nt,
             "index_build_s": build, "mean_count_of_a_20_byte_pattern": c4r.get("mean_count"), "share_of_patterns_occurring_once": c4r.get("share_count_1"),
             "wt_rank": {"Gq/s": nq2 / ms / 1e6, "kernel_ms": ms, "reference_digest_match": digests_match(out2, c4r, "wt_rank") if ok else None}}
    del gi, gc
    st = to_dev(pkg.rnd_positions(15, nq2, nt - m, 0), dev)
    pats = text[(st.view(-1, 1) + torch.arange(m, device=dev).view(1, m)).reshape(-1)].contiguous()
    del st

    def leg(name):
        steps_ms = []
        _, ms_ = time_steps(lambda: csa.count(pats, m, out2), 4, 1, barrier, per_step=steps_ms)
        return {"Mcount/s": nq2 / ms_ / 1e3, "kernel_ms": ms_, "kernel_ms_per_batch": spread_of(steps_ms), "patterns": nq2, "m": m,
                "index_bytes": csa.device_bytes(), "kmer_table": {"k": csa.kmer_table_depth(), "bytes": csa.kmer_table_bytes()},
                "resident_bytes_by_part": csa.footprint_parts(), "reference_digest_match": digests_match(out2, c4r, "count") if ok else None}

    block["count_default"] = leg("default")
    sdsl_bytes = len(csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL))
    block["sdsl_stream_bytes"] = sdsl_bytes
    csa.drop_sa()
    block["count_sa_dropped"] = leg("sa_dropped")
    try:
        csa.set_footprint(int(1.5 * sdsl_bytes))
        block["count_lean"] = leg("lean")
        block["count_lean"]["x_sdsl_stream_bytes"] = block["count_lean"]["index_bytes"] / sdsl_bytes
    except Exception as e_:
        block["count_lean_error"] = str(e_)
    first = ex.get("fm_count", {}).get("Mcount/s")
    if first:
        block["default_route_vs_first_stand_in"] = block["count_default"]["Mcount/s"] / first
    ex["fm_count_repetitive"] = block
    del csa, wt, text, pats, out2


def leg_big(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
    # An index of more than 2^32 symbols (opt-in: 172 GB of working memory in the suffix sorter): csa_wt from a
    # synthetic text of 2^32 + 777 symbols — 64-bit suffix sorter, fused lines with 64-bit superblock counts, SA / ISA
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
                         "kmer_table": {"k": cb.kmer_table_depth(), "bytes": cb.kmer_table_bytes()},
                             "note": "fused 16-ary lines with 64-bit superblock counts; count: k-mer table with 40-bit intervals -> wide flat search kernel -> "
                                     "text comparison at <= 8 suffixes (fm_count2.hip, WIDE variants)"}
    del wtb, cb, tb, pb, ob, ib, sb, orb, stb
    torch.cuda.empty_cache()


def leg_sharded(c):
    a, pkg, dev, local, rank, world, barrier, comm_dev, G, gq = c.a, c.pkg, c.dev, c.local, c.rank, c.world, c.barrier, c.comm_dev, c.G, c.gq
    nq, n_bits, bv, words, idx, out, ex = c.nq, c.n_bits, c.bv, c.words, c.idx, c.out, c.ex
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
    fs = {"n_gpus": world, "patterns_total": total, "m": m, "text_bytes": nt, "index_build_s": build, "index_bytes_per_gpu": csa.device_bytes(),
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
