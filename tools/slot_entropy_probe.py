"""How compressible are the fused 16-ary lines of an FM-index?  (sizing of a compressed fused line.)  CPU only, build container
only (the BWT comes from the REAL sdsl-lite through oracle/_ref).  For the first MiB-count of the bench text: a 16-ary Huffman tree over
the BWT's symbols, the root node's slot sequence, and per window of 46 / 92 / 184 positions (a section, a half line, a line): the distinct
slots (a per-window dictionary stores ceil(log2 d) bits per position) and the window's order-0 entropy (what an entropy coder could reach).
usage: python tools/slot_entropy_probe.py [MiB = 64]      -> profiles/fused_line_entropy_r06.txt"""
import heapq
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

pkg = importlib.import_module("sdsl-lite_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
text = pkg.english_text(mib << 20, 1234)
bwt = np.asarray(ol.RCsa(bytes(text)).bwt()).astype(np.uint8)
n = bwt.size
occ = np.bincount(bwt, minlength=256).astype(np.float64)
syms = [(occ[c], [c]) for c in range(256) if occ[c] > 0]
pad = (15 - (len(syms) - 1) % 15) % 15
heap = [(w, i, s) for i, (w, s) in enumerate(syms)] + [(0.0, 1000 + i, []) for i in range(pad)]
heapq.heapify(heap)
depth = np.zeros(256, dtype=np.int64)
ctr, nodes = 5000, {}
while len(heap) > 1:
    grp = [heapq.heappop(heap) for _ in range(16)]
    ss = sum((g[2] for g in grp), [])
    for c in ss:
        depth[c] += 1
    nodes[ctr] = [g[2] for g in grp]
    heapq.heappush(heap, (sum(g[0] for g in grp), ctr, ss))
    ctr += 1
root = nodes[ctr - 1]
print(f"english_text({mib} MiB, 1234): BWT of {n} symbols through the real library; 16-ary Huffman: {(occ * depth).sum() / n:.3f} fused steps per symbol; "
      f"symbols per slot of the root: {[len(g) for g in root]}")
slot = np.zeros(256, dtype=np.uint8)
for t, g in enumerate(root):
    for c in g:
        slot[c] = t
seq = slot[bwt]
for W in (46, 92, 184):
    k = n // W
    a = seq[:k * W].reshape(k, W)
    srt = np.sort(a, axis=1)
    d = 1 + (srt[:, 1:] != srt[:, :-1]).sum(axis=1)
    planes = np.where(d <= 1, 0, np.where(d <= 2, 1, np.where(d <= 4, 2, np.where(d <= 8, 3, 4))))
    ent = 0.0
    for t in range(16):
        c = (a == t).sum(axis=1).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            ent += np.where(c > 0, -c * np.log2(c / W), 0).sum()
    hist = np.bincount(d, minlength=17)[1:] / k
    print(f"windows of {W} positions of the root node's slot sequence: share with <= 2 / <= 4 / <= 8 / > 8 distinct slots {hist[:2].sum():.3f} / {hist[:4].sum():.3f} / "
          f"{hist[:8].sum():.3f} / {hist[8:].sum():.3f}; a per-window dictionary: {planes.mean():.2f} bits per position (stored today: 4); "
          f"window-local order-0 entropy: {ent / (k * W):.2f} bits per position")
print(f"runs: n / r = {n / (1 + (seq[1:] != seq[:-1]).sum()):.2f} (slots), {n / (1 + (bwt[1:] != bwt[:-1]).sum()):.2f} (symbols)")
print("a 16-ary line = 184 positions x 4 bits + sixteen 18-bit counts (288 bits = 28 % of the line, incompressible): the dictionary form saves "
      "(4 - 2.77) x 184 - 64 bits of masks = 162 of 1024 bits = 16 %")
