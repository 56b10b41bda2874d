import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("sdsl-lite_amd")
text = torch.from_numpy(pkg.english_text(1 << 30, 1234)).cuda()
csa = pkg.csa_wt(text=text)
t0 = time.time(); b = csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL); t1 = time.time() - t0
csa.set_footprint(int(1.5 * len(b)))
t0 = time.time(); b2 = csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL); t2 = time.time() - t0
print(f"serialize 1 GiB index: {t1:.2f} s with its binary levels, {t2:.2f} s from the fused lines at the lean footprint; same bytes: {b == b2}")
