import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("sdsl-lite_amd")
n, sigma = (1 << 32) + 1_000_003, 40
g = torch.Generator(device="cuda").manual_seed(5)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
step = 1 << 28
for a in range(0, n, step):
    b = min(n, a + step)
    u = torch.rand(b - a, device="cuda", generator=g)
    text[a:b] = (1 + (u * u * sigma).to(torch.int64).clamp_(max=sigma - 1)).to(torch.uint8)
pkg.set_timing(True)
nq = 20_000_000
j = torch.randint(0, n, (nq,), device="cuda", dtype=torch.int64, generator=g)
for mode in ("1", "0"):
    os.environ["SDSL_HIP_WT_FUSED_SELECT"] = mode
    wt = pkg.wt_huff(text=text)
    r = wt.rank(j, text[j]).to(torch.int64)
    out = wt.select(r + 1, text[j]); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        out = wt.select(r + 1, text[j]); ts.append(pkg.last_kernel_ms())
    assert torch.equal(out.to(torch.int64), j)
    print(f"fused select directory={mode}: {min(ts):.2f} ms per {nq} = {nq / min(ts) / 1e6:.2f} Gq/s, device bytes {wt.device_bytes() / 2**30:.2f} GiB", flush=True)
    wt.close(); del wt; torch.cuda.empty_cache()
