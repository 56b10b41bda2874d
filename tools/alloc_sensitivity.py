"""Does WHERE hipMalloc puts the index explain the box-to-box spread of the direct rank kernel?
The same 2^34-bit vector is laid out six times in one process — behind dummy allocations of different sizes, so that its
rank lines start at differently aligned addresses — and the direct kernel and the bucketed path are timed on each copy."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("sdsl-lite_amd")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 34
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**9
n = 1 << logn
dev = torch.device("cuda", 0)
words = torch.from_numpy(pkg.set_random_bits(n, 42).view(np.int64)).to(dev)
idx = torch.from_numpy(pkg.rnd_positions(7, nq, n + 1, 0).view(np.int64)).to(dev)
out = torch.empty_like(idx)
pkg.set_timing(True)
keep = []
for trial, pad in enumerate([0, 4096, (1 << 21) + 4096, (1 << 30) + (1 << 20), 3 << 30, 0]):
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))  # shifts the next allocation
    bv = pkg.bit_vector(words, n, device=0, select1=False, select0=False)
    info = bv.layout_info()
    res = {}
    for mode, name in ((0, "direct"), (1, "bucketed")):
        pkg.set_option("rank_sorted", mode)
        bv.rank(idx, 1, out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            bv.rank(idx, 1, out); ts.append(pkg.last_kernel_ms())
        res[name] = min(ts)
    pkg.set_option("rank_sorted", -1)
    a = info["lines_ptr"]
    al = (a & -a).bit_length() - 1
    print(f"copy {trial}: lines at 0x{a:x} (aligned to 2^{al}) direct {res['direct']:.3f} ms = {nq/res['direct']/1e6:.2f} G/s   "
          f"bucketed {res['bucketed']:.3f} ms = {nq/res['bucketed']/1e6:.2f} G/s", flush=True)
    bv.close()
