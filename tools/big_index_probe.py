"""Rates on an index of more than 2^32 symbols (binary levels of the wavelet tree, no fused layout, no k-mer table):
Usage: python tools/big_index_probe.py [symbols] [sigma]"""
import importlib
import sys
import time

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
    return n / best


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 32) + 777
    sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    g = torch.Generator(device="cuda").manual_seed(1)
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        u = torch.rand(b - a, device="cuda", generator=g)
        text[a:b] = (1 + (u * u * sigma).to(torch.int64).clamp_(max=sigma - 1)).to(torch.uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    csa = pkg.csa_wt(text=text)
    torch.cuda.synchronize()
    print(f"{n} symbols (2^32 + {n - (1 << 32)}), sigma {sigma}: csa_wt built from text in {time.perf_counter() - t0:.1f} s, "
          f"{csa.device_bytes() / 1e9:.2f} GB resident, sampling {csa.sampling()}")
    npat, m = 10_000_000, 20
    st = torch.randint(0, n - m, (npat,), device="cuda", dtype=torch.int64, generator=g)
    pats = text[st[:, None] + torch.arange(m, device="cuda")[None, :]].contiguous().reshape(-1)
    out = torch.empty(npat, dtype=torch.int64, device="cuda")
    r = rate(lambda: csa.count(pats, m, out=out), npat)
    print(f"count, {npat} patterns of {m} bytes cut from the text: {r / 1e6:.0f} Mcount/s; every pattern found: {bool((out >= 1).all())}")
    wt = csa.wavelet_tree
    nq = 100_000_000
    i = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    c = text[torch.randint(0, n, (nq,), device="cuda", dtype=torch.int64, generator=g)]
    o = torch.empty(nq, dtype=torch.int64, device="cuda")
    print(f"wt.rank(i, c), {nq} queries, symbols drawn from the text: {rate(lambda: wt.rank(i, c, out=o), nq) / 1e9:.1f} G/s")
    idx = torch.randint(0, n + 1, (1_000_000,), device="cuda", dtype=torch.int64, generator=g)
    print(f"csa[i], 10^6 places: {rate(lambda: csa.sa(idx), 1_000_000) / 1e6:.1f} M/s")
    csa.drop_sa()
    print(f"after drop_sa (samples 32 / 64 only, {csa.device_bytes() / 1e9:.2f} GB resident):")
    r = rate(lambda: csa.count(pats, m, out=out), npat)
    print(f"count: {r / 1e6:.0f} Mcount/s; every pattern found: {bool((out >= 1).all())}")
    print(f"csa[i], 10^6 places: {rate(lambda: csa.sa(idx), 1_000_000) / 1e6:.1f} M/s")


if __name__ == "__main__":
    main()
