"""wt_huff over MORE than 2^32 symbols: does it build, and are rank / access / select right?  (No fused layout there — its counts
are 32-bit — so the binary levels answer.)  Reference: torch prefix counts on the device.
Usage: python tools/wt_big_probe.py [symbols] [sigma]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("sdsl-lite_amd")


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 32) + 1_000_003
    sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    g = torch.Generator(device="cuda").manual_seed(5)
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for a in range(0, n, step):                       # a skewed alphabet: symbol = 1 + floor(sigma * u^2)
        b = min(n, a + step)
        u = torch.rand(b - a, device="cuda", generator=g)
        text[a:b] = (1 + (u * u * sigma).to(torch.int64).clamp_(max=sigma - 1)).to(torch.uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wt = pkg.wt_huff(text=text)
    torch.cuda.synchronize()
    print(f"n = {n} symbols (2^32 + {n - (1 << 32)}), sigma {wt.sigma()}, built in {time.perf_counter() - t0:.2f} s, "
          f"{wt.device_bytes() / 1e9:.2f} GB resident, size() = {wt.size()}", flush=True)
    nq = 2_000_000
    i = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    i[:1000] = n - torch.arange(1000, device="cuda")
    i[1000:2000] = (1 << 32) - 500 + torch.arange(1000, device="cuda")
    ok = True
    for c in (1, 2, sigma // 2, sigma, sigma + 3):
        cs = torch.cumsum((text == c).to(torch.int64), 0)
        want = torch.where(i > 0, cs[(i - 1).clamp_(min=0)], torch.zeros_like(i))
        cc = torch.full((nq,), c, dtype=torch.uint8, device="cuda")
        got = wt.rank(i, cc)
        same = bool(torch.equal(got.to(torch.int64), want))
        total = int(cs[-1])
        msg = f"symbol {c}: {total} occurrences, rank equal: {same}"
        if total:
            k = torch.randint(1, total + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
            k[:1000] = total - torch.arange(1000, device="cuda").clamp_(max=total - 1)
            pos = torch.searchsorted(cs, k)           # first position with cs >= k
            sel = wt.select(k, cc)
            s_same = bool(torch.equal(sel.to(torch.int64), pos))
            msg += f", select equal: {s_same}"
            same = same and s_same
        print(msg, flush=True)
        ok = ok and same
        del cs
    j = torch.randint(0, n, (nq,), device="cuda", dtype=torch.int64, generator=g)
    j[:1000] = n - 1 - torch.arange(1000, device="cuda")
    acc = wt.access(j)
    a_same = bool(torch.equal(acc.to(torch.uint8), text[j]))
    print(f"access equal: {a_same}")
    ok = ok and a_same
    for name, fn in (("rank", lambda: wt.rank(i, cc)), ("access", lambda: wt.access(j))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(f"{name}: {nq / (time.perf_counter() - t0) / 1e9:.2f} G/s on {nq} queries")
    print("ALL EQUAL" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
