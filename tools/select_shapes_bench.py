"""select_1 under non-uniform density: 2^log_n-bit vectors whose ones are (a) clustered in 1 % of the
range, (b) in alternating dense / empty 2^20-bit stripes, (c) isolated (one per 2^16 bits, CRAFTED-SPARSE style), against
the uniform 50 % vector; plain (select_support_mcl), rrr_vector<63> and sd_vector.  Hand tool for gpurun.

usage: select_shapes_bench.py [log_n=34] [queries=1e8]      (bench.py --extras shapes calls run() below)
"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = ("uniform", "clustered_1pct", "stripes_2^20", "isolated_2^16")


def run(pkg, logn=34, nq=10**8, emit=print, device=0):
    """-> {shape: {kind: {"Gq/s", "ms", "ones", "of_uniform", "answers_ok"}}}"""
    import torch
    n = 1 << logn
    nw = n // 64
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev).manual_seed(5)

    def rnd_words(count):
        return torch.randint(-2**63, 2**63 - 1, (count,), device=dev, dtype=torch.int64, generator=g)

    def shape(name):
        w = torch.zeros(nw, dtype=torch.int64, device=dev)
        if name == "uniform":
            w = rnd_words(nw)
        elif name == "clustered_1pct":  # all ones inside the middle 1 % of the range
            lo, cnt = int(nw * 0.495), nw // 100
            w[lo:lo + cnt] = rnd_words(cnt)
        elif name == "stripes_2^20":  # 2^20-bit stripes, alternately 50 % dense and empty
            ws = (1 << 20) // 64
            v = rnd_words(nw).view(-1, ws)
            v[1::2] = 0
            w = v.reshape(-1)
        elif name == "isolated_2^16":  # one set bit per 2^16 bits, at a random offset
            k = n >> 16
            pos = torch.arange(k, device=dev, dtype=torch.int64) * (1 << 16) + torch.randint(0, 1 << 16, (k,), device=dev, generator=g)
            w.index_put_((pos >> 6,), torch.ones(1, dtype=torch.int64, device=dev) << (pos & 63), accumulate=True)
        return w

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            fn()
            ts.append(pkg.last_kernel_ms())
        return min(ts)

    pkg.set_timing(True)
    base, res = {}, {}
    try:
        for name in SHAPES:
            w = shape(name)
            for kind in ("plain", "rrr63", "sd"):
                t0 = time.time()
                if kind == "plain":
                    v = pkg.bit_vector(w, n, device=device, select1=True, select0=False)
                elif kind == "rrr63":
                    v = pkg.rrr_vector(w, n, device=device)
                else:
                    if name in ("uniform", "stripes_2^20") and logn > 32:
                        continue  # 2^33 ones: not what sd_vector is for
                    v = pkg.sd_vector(words=w, n_bits=n, device=device)
                torch.cuda.synchronize()
                build = time.time() - t0
                ones = v.ones()
                # arguments may repeat: a sparse vector has fewer ones than the batch has queries
                i = torch.randint(1, ones + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
                out = torch.empty_like(i)
                ms = timed(lambda: v.select(i, 1, out))
                # the answers: bit set there, and exactly i - 1 ones in front of it
                chk = out[: 1 << 20].clone()
                ok = bool((v.rank(chk, 1) == i[: 1 << 20] - 1).all()) and bool((v.access(chk) == 1).all())
                rate = nq / ms / 1e6
                if name == "uniform":
                    base[kind] = rate
                rel = rate / base[kind] if kind in base else None
                res.setdefault(name, {})[kind] = {"Gq/s": rate, "ms": ms, "ones": ones, "of_uniform": rel, "answers_ok": ok}
                emit(f"{name:16s} {kind:6s} ones={ones:>11d} build {build:6.2f}s  select_1 {rate:7.2f} G/s ({ms:8.3f} ms for {nq} queries) "
                     f"{'%.2f of uniform' % rel if rel else ''}  answers_ok={ok}")
                if hasattr(v, "release_scratch"):
                    v.release_scratch()
                v.close()
                del v, i, out
                torch.cuda.empty_cache()
            del w
    finally:
        pkg.set_timing(False)
    return res


if __name__ == "__main__":
    pkg = importlib.import_module("sdsl-lite_amd")
    run(pkg, int(sys.argv[1]) if len(sys.argv) > 1 else 34, int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**8,
        emit=lambda s: print(s, flush=True))
