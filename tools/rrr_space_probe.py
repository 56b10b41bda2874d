"""rrr_vector<63> between 1 and 50 % density: bits per bit on the device (default and option rrr_sparse_limit = 20) against the
size of SDSL's serialised vector (which the library writes byte for byte), and what the direct kernels make of either.
Usage: python tools/rrr_space_probe.py [log2 bits] [queries]"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("sdsl-lite_amd")


def rate(fn, n, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return n / best / 1e9


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
    n_bits = 1 << lg
    print(f"2^{lg} bits, {nq:.0e} uniformly random queries per call, direct kernels (option rrr_sorted = 0) unless noted")
    print("| density | SDSL bits/bit | default: bits/bit (x SDSL) | rank G/s | select_1 G/s | limit 20: bits/bit (x SDSL) | rank G/s | select_1 G/s |")
    print("|---|---|---|---|---|---|---|---|")
    idx = torch.randint(0, n_bits + 1, (nq,), device="cuda", dtype=torch.int64)
    out = torch.empty_like(idx)
    for pct in (1, 2, 5, 10, 15, 20, 30, 50):
        w = pkg.density_bits(n_bits, 7 + pct, pct)
        cells = []
        sdsl = None
        for limit in (10, 20):
            pkg.set_option("rrr_sparse_limit", limit)
            v = pkg.rrr_vector(w, n_bits)
            pkg.set_option("rrr_sparse_limit", 20)
            if sdsl is None:
                sdsl = len(v.serialize()) * 8 / n_bits
            bpb = v.device_bytes() * 8 / n_bits
            i = torch.randint(1, v.ones() + 1, (nq,), device="cuda", dtype=torch.int64)
            pkg.set_option("rrr_sorted", 0)
            r = rate(lambda: v.rank(idx, 1, out=out), nq)
            s = rate(lambda: v.select(i, 1, out=out), nq)
            pkg.set_option("rrr_sorted", -1)
            cells.append(f"{bpb:.3f} ({bpb / sdsl:.2f}) | {r:.1f} | {s:.1f}")
            v.close()
        print(f"| {pct} % | {sdsl:.3f} | " + " | ".join(cells) + " |", flush=True)


if __name__ == "__main__":
    main()
