"""Bucketed (sorted) batch rank vs the direct kernel: equality + per-phase timing (hand tool for gpurun).

usage: sorted_rank_probe.py <log2 bits> <queries> [mode]   mode: child run with SDSL_HIP_RANK_SORTED preset
"""
import hashlib, importlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(logn, nq):
    import torch
    pkg = importlib.import_module("sdsl-lite_amd")
    n = (1 << logn) - 37  # not a multiple of anything
    g = torch.Generator(device="cuda").manual_seed(42)
    words = torch.randint(-2**63, 2**63 - 1, ((n + 63) // 64,), device="cuda", dtype=torch.int64, generator=g)
    bv = pkg.bit_vector(words, n)
    del words
    idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    idx[:7] = torch.tensor([0, n, n + 1, 2**62, 1, n - 1, 447], device="cuda")  # edges incl. out-of-range -> NPOS
    out = torch.empty_like(idx)
    pkg.set_timing(True)
    light = os.environ.get("PROBE_LIGHT") is not None
    for bit in ((1,) if light else (1, 0)):
        bv.rank(idx, bit, out); torch.cuda.synchronize()
        ts = []
        for _ in range(2 if light else 4):
            bv.rank(idx, bit, out); ts.append(pkg.last_kernel_ms())
        ms = min(ts)
        h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"rank{bit}: {ms:.3f} ms {nq/ms/1e6:.2f} G/s frac {96*nq/ms/1e6/8000:.3f} sha {h} scratch {bv.device_bytes()/2**30:.2f} GiB", flush=True)
    # select_1 / select_0 through the same switch (option select_sorted follows SDSL_HIP_RANK_SORTED here)
    pkg.set_option("select_sorted", int(os.environ.get("SDSL_HIP_RANK_SORTED", "-1")))
    for bit in ((1,) if light else (1, 0)):
        tot = bv.ones() if bit else n - bv.ones()
        si = torch.randint(1, tot + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        si[:5] = torch.tensor([1, tot, tot + 1, 0, 2], device="cuda")
        bv.select(si, bit, out); torch.cuda.synchronize()
        ts = []
        for _ in range(2 if light else 3):
            bv.select(si, bit, out); ts.append(pkg.last_kernel_ms())
        ms = min(ts)
        h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"select{bit}: {ms:.3f} ms {nq/ms/1e6:.2f} G/s frac {112*nq/ms/1e6/8000:.3f} sha {h}", flush=True)
    if light:
        return
    # skewed batch: everything in one bucket, then two values only
    idx2 = torch.randint(0, 1 << 20, (nq,), device="cuda", dtype=torch.int64, generator=g)
    bv.rank(idx2, 1, out); torch.cuda.synchronize()
    t0 = time.time(); bv.rank(idx2, 1, out); torch.cuda.synchronize(); t1 = time.time()
    print(f"skewed(2^20 window): {(t1-t0)*1e3:.2f} ms sha {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "child":
    child(int(sys.argv[1]), int(float(sys.argv[2])))
else:
    variants = [("0", {})] + [("1", {"SDSL_HIP_TRACE_SORTED": "1", "SDSL_HIP_SORTED_THREADS": a.split(":")[0], "SDSL_HIP_SORTED_VARIANT": (a.split(":") + ["0"])[1], "SDSL_HIP_SORTED_PER": (a.split(":") + ["0", "16"])[2]}) for a in (sys.argv[3:] or ["256"])]
    for mode, extra in variants:
        env = dict(os.environ, SDSL_HIP_RANK_SORTED=mode, **extra)
        print(f"--- SDSL_HIP_RANK_SORTED={mode} {extra}", flush=True)
        r = subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], "child"], env=env, capture_output=True, text=True)
        err = [l for l in r.stderr.splitlines() if "amdgpu.ids" not in l]
        print(r.stdout[-3000:]); print("\n".join(err[:2] + err[-3:]))
