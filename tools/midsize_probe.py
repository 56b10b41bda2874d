"""Mid-size batches (10^7 .. 10^8.5 positions on 2^34 bits): what would ONE partition pass buy?  The direct kernel on positions that are
already grouped by bin (1024 / 256 / 4096 bins of the vector; inside a bin in arrival order), and fully sorted, against the
direct kernel on the batch as it arrives: the answer stage of a single-level machine cannot be faster than this.
usage: midsize_probe.py [log2 bits = 34]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 34
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
nb = 1 << ln
w = torch.randint(-2**63, 2**63 - 1, (nb // 64,), device=dev, dtype=torch.int64, generator=g)
bv = pkg.bit_vector(w, nb, device=0, select1=False, select0=False)
del w
pkg.set_timing(True)
pkg.set_option("rank_sorted", 0)
for nq in (10**7, 3 * 10**7, 10**8, 3 * 10**8):
    idx = torch.randint(0, nb + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    out = torch.empty_like(idx)
    row = {}
    for name, bins_log in (("arrival order", None), ("256 bins", 8), ("1024 bins", 10), ("4096 bins", 12), ("16384 bins", 14), ("sorted", 99)):
        if bins_log is None:
            q = idx
        elif bins_log == 99:
            q = torch.sort(idx).values
        else:
            key = idx >> (ln - bins_log)
            q = idx[torch.sort(key, stable=True).indices]
        bv.rank(q, 1, out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            bv.rank(q, 1, out); ts.append(pkg.last_kernel_ms())
        row[name] = nq / min(ts) / 1e6
    print(f"2^{ln} bits, {nq:.0e} positions, direct kernel G/s: " + ", ".join(f"{k}: {v:.1f}" for k, v in row.items()), flush=True)
