"""rrr_vector<63> at 10-30 % density BY DEFAULT (round 6: the classes up to 20 stay enumerative and the bucketed route decodes them):
bits per bit on the device against the real library's stream (which the library writes byte for byte), bucketed and direct rates.
Usage: python tools/rrr_space_probe3.py [log2 bits = 34] [queries = 1e9]      -> profiles/rrr_space_r06.txt"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("sdsl-lite_amd")


def rate(fn, n, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        fn()
        best = min(best, pkg.last_kernel_ms())
    return n / best / 1e6


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 34
    nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
    n_bits = 1 << lg
    pkg.set_timing(True)
    print(f"2^{lg} bits, {nq:.0e} uniformly random queries per call; no option set (rrr_sparse_limit = 20 is the default), kernel time")
    print("| density | SDSL rrr_vector<63> bits/bit | device bits/bit (x SDSL) | with rrr_sparse_limit = 10: bits/bit (x SDSL) | rank_1 bucketed G/s | rank_1 direct G/s | "
          "select_1 bucketed G/s | select_1 direct G/s | rank_1 bucketed at limit 10 G/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    idx = torch.randint(0, n_bits + 1, (nq,), device="cuda", dtype=torch.int64)
    out = torch.empty_like(idx)
    for pct in (5, 10, 15, 20, 30):
        w = pkg.density_bits(n_bits, 7 + pct, pct)
        v = pkg.rrr_vector(w, n_bits)
        sdsl = len(v.serialize()) * 8 / n_bits
        bpb = v.device_bytes() * 8 / n_bits
        i = torch.randint(1, v.ones() + 1, (nq,), device="cuda", dtype=torch.int64)
        pkg.set_option("rrr_sorted", 1)
        rb = rate(lambda: v.rank(idx, 1, out=out), nq)
        sb = rate(lambda: v.select(i, 1, out=out), nq)
        chk = out[:1_000_000].clone()
        pkg.set_option("rrr_sorted", 0)
        nd = min(nq, 100_000_000)
        rd = rate(lambda: v.rank(idx[:nd], 1, out=out[:nd]), nd)
        sd = rate(lambda: v.select(i[:nd], 1, out=out[:nd]), nd)
        assert torch.equal(chk, out[:1_000_000])
        v.close()
        pkg.set_option("rrr_sparse_limit", 10)
        v10 = pkg.rrr_vector(w, n_bits)
        pkg.set_option("rrr_sparse_limit", 20)
        b10 = v10.device_bytes() * 8 / n_bits
        pkg.set_option("rrr_sorted", 1)
        r10 = rate(lambda: v10.rank(idx, 1, out=out), nq)
        pkg.set_option("rrr_sorted", -1)
        v10.close()
        print(f"| {pct} % | {sdsl:.3f} | {bpb:.3f} ({bpb / sdsl:.2f}) | {b10:.3f} ({b10 / sdsl:.2f}) | {rb:.1f} | {rd:.1f} | {sb:.1f} | {sd:.1f} | {r10:.1f} |", flush=True)
        del w


if __name__ == "__main__":
    main()
