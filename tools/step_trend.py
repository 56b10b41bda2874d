"""Per-launch duration of 30 back-to-back k_rank launches (DVFS / power-cap check)."""
import importlib, sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdsl-lite_amd")
n = 1 << 34; nq = 10**9
g = torch.Generator(device="cuda").manual_seed(42)
words = torch.randint(-2**63, 2**63 - 1, (n // 64,), device="cuda", dtype=torch.int64, generator=g)
bv = pkg.bit_vector(words, n, select1=False, select0=False)
idx = torch.randint(0, n + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
out = torch.empty_like(idx)
K = 30
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
bv.rank(idx, 1, out); torch.cuda.synchronize()
smi = []
def poll():
    for _ in range(6):
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            smi.append([l.strip() for l in r.splitlines() if "sclk" in l or "mclk" in l or "Power" in l or "fclk" in l])
        except Exception as e:
            smi.append([str(e)])
        time.sleep(0.1)
t = threading.Thread(target=poll); t.start()
ev[0].record()
for i in range(K):
    bv.rank(idx, 1, out); ev[i + 1].record()
torch.cuda.synchronize(); t.join()
print("ms per launch:", " ".join(f"{ev[i].elapsed_time(ev[i+1]):.2f}" for i in range(K)))
for s in smi[:4]: print(s)
time.sleep(2.0)
ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for i in range(3):
    ev2[i].record(); bv.rank(idx, 1, out); ev2[i + 1].record(); torch.cuda.synchronize(); time.sleep(0.5)
print("isolated (0.5 s gaps):", " ".join(f"{ev2[i].elapsed_time(ev2[i+1]):.2f}" for i in range(3)))
