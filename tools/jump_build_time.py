import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import torch, bench
pkg = importlib.import_module("sdsl-lite_amd")
text = bench.synthetic_text(1 << 30, 1234, torch.device("cuda", 0))
os.environ["SDSL_HIP_FM_JUMP"] = "0"
t0 = time.time(); csa = pkg.csa_wt(text=text); torch.cuda.synchronize(); print("build without table", time.time() - t0)
for k in (4, 5, 5, 3):
    t0 = time.time(); csa.set_jump_depth(k); torch.cuda.synchronize(); print("set_jump_depth", k, time.time() - t0)
