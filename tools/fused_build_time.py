"""build time of the FM-index of the 1 GiB bench text with and without the fused layout (run under rocprofv3 for the
per-kernel split)"""
import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import torch, bench
pkg = importlib.import_module("sdsl-lite_amd")
text = bench.synthetic_text(1 << 30, 1234, torch.device("cuda", 0))
for fused in (sys.argv[1:] or ["1", "0", "1"]):
    os.environ["SDSL_HIP_WT_FUSED"] = fused
    torch.cuda.synchronize()
    t0 = time.time(); csa = pkg.csa_wt(text=text); torch.cuda.synchronize(); print("fused", fused, "build", round(time.time() - t0, 3), "s", "bytes", csa.device_bytes() if hasattr(csa, "device_bytes") else "")
    csa.close()
