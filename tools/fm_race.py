"""Two FM-index handles on one device, count() batches enqueued alternately on two streams — what a device group of two members on
one GPU does (group.cpp: group_run) — against the same batches on one stream.  Hand tool for gpurun: hunts the intermittent
mismatch of `csa_wt_multi_hip count_batch` in the adaptor parity client.  usage: fm_race.py [rounds = 400]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("sdsl-lite_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
text = (97 + rng.integers(0, 7, 300_000)).astype(np.uint8)
csa = [pkg.csa_wt(text=text), pkg.csa_wt(text=text)]
print("k-mer table depth", csa[0].kmer_table_depth(), "jump depth", csa[0].jump_depth(), flush=True)
m, npat = 6, 50_000
st = rng.integers(0, text.size - m, npat)
pats = torch.from_numpy(text[st[:, None] + np.arange(m)[None, :]].reshape(-1).copy()).to(dev)
want = csa[0].count(pats, m).clone()
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad_rounds = 0
for it in range(rounds):
    out = torch.full((npat,), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for c in range(2):  # pieces
        for r in range(2):  # members
            lo = r * 25_000 + c * 12_500
            with torch.cuda.stream(streams[r]):
                csa[r].count(pats[lo * m:(lo + 12_500) * m], m, out[lo:lo + 12_500])
    torch.cuda.synchronize()
    diff = (out != want).nonzero().flatten()
    if diff.numel():
        bad_rounds += 1
        if bad_rounds <= 5:
            i = int(diff[0])
            print(f"round {it}: {diff.numel()} of {npat} answers differ, first at {i}: got {int(out[i])} want {int(want[i])}; "
                  f"positions {diff[:8].tolist()} ... {diff[-3:].tolist()}", flush=True)
print(f"rounds {rounds}, with mismatches {bad_rounds}")
