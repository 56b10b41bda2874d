"""bits per bit of the compressed bit-vector types at 2^30 bits: what the REAL library's streams take (rrr_vector<15> specialisation,
rrr_vector<63>) against what is resident on the device for the same vector handed over as the rrr_vector<15> stream — as plain rank
lines (sdsl_hip_bv_create_from_sdsl) and kept compressed (sdsl_hip_rrr_create_from_sibling).  -> profiles/rrr_space_r05.txt"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib as ol
pkg = importlib.import_module("sdsl-lite_amd")
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << ln
print(f"# 2^{ln} bits; bits per bit.  SDSL columns: bytes of the real library's serialize(); device columns: *_device_bytes()")
print("| density | SDSL rrr_vector<15> | SDSL rrr_vector<63> | device, from the rrr<15> stream: plain rank lines | device, kept compressed (x SDSL rrr<15>) (x SDSL rrr<63>) | rank_1 G/s compressed |")
print("|---|---|---|---|---|---|")
pkg.set_timing(True)
for pct in (1, 2, 5, 10, 15, 20, 30, 50):
    rng = np.random.default_rng(pct)
    bits = rng.random(n) < pct / 100
    w = np.packbits(bits, bitorder="little").view(np.uint64)
    del bits
    b15 = ol.ref_sibling_bytes(w, n, 6)
    r63 = ol.RRrr(w, n)
    b63 = len(r63.serialize())
    del r63
    plain = pkg.bit_vector(sdsl_bytes=b15, kind=pkg.capi.SIBLING_RRR15)
    comp = pkg.rrr_vector(sdsl_bytes=b15, sibling_kind=pkg.capi.SIBLING_RRR15)
    idx = torch.randint(0, n + 1, (10**8,), device="cuda", dtype=torch.int64)
    out = torch.empty_like(idx)
    comp.rank(idx, 1, out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        comp.rank(idx, 1, out); ts.append(pkg.last_kernel_ms())
    assert torch.equal(out[:10**6], plain.rank(idx[:10**6].clone(), 1))
    f = lambda b: b * 8 / n
    print(f"| {pct} % | {f(len(b15)):.3f} | {f(b63):.3f} | {f(plain.device_bytes()):.3f} | {f(comp.device_bytes()):.3f} ({comp.device_bytes() / len(b15):.2f}) "
          f"({comp.device_bytes() / b63:.2f}) | {1e8 / min(ts) / 1e6:.1f} |", flush=True)
    plain.close(); comp.close()
