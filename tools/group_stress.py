"""Hunts the intermittent mismatch of count() over a device group (oracle/ref/adaptor_parity.cpp: csa_wt_multi_hip count_batch).
Phases, each `rounds` times, host arrays as the adaptors pass them:
  A  ONE group, ONE set of replicas, count over the group again and again          -> is it the batch path (staging, streams)?
  B  replicas rebuilt every round (group_fm_create_from_text: builder threads)      -> is it the index build?
  C  single handle built on the main thread, rebuilt every round                   -> is it the build itself, without threads?
usage: group_stress.py [rounds = 150] [transport = copy2|rccl1]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("sdsl-lite_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
transport = sys.argv[2] if len(sys.argv) > 2 else "copy2"
rng = np.random.default_rng(3)
text = (97 + rng.integers(0, 7, 300_000)).astype(np.uint8)
m, npat = 6, 50_000
st = rng.integers(0, text.size - m, npat)
pats = np.ascontiguousarray(text[st[:, None] + np.arange(m)[None, :]].reshape(-1))
ref = pkg.csa_wt(text=text)
want = np.asarray(ref.count(pats, m)).astype(np.uint64)
assert (want >= 1).all()
if transport == "copy2":
    os.environ["SDSL_HIP_GROUP_TRANSPORT"] = "copy"
    grp = pkg.device_group([0, 0])
    del os.environ["SDSL_HIP_GROUP_TRANSPORT"]
else:
    grp = pkg.device_group([0])


def report(tag, it, got):
    d = np.flatnonzero(got != want)
    if d.size:
        print(f"{tag} round {it}: {d.size} of {npat} differ; first {d[0]} got {got[d[0]]} want {want[d[0]]}; positions {d[:6].tolist()} .. {d[-2:].tolist()}",
              flush=True)
    return int(d.size > 0)


reps = grp.csa_from_text(text)
bad = sum(report("A", it, np.asarray(grp.count(reps, pats, m, chunks=2)).astype(np.uint64)) for it in range(rounds))
print(f"A (fixed replicas, {transport}): {bad} of {rounds} rounds with mismatches", flush=True)
bad = 0
for it in range(rounds):
    for r in reps:
        r.close()
    reps = grp.csa_from_text(text)
    bad += report("B", it, np.asarray(grp.count(reps, pats, m, chunks=2)).astype(np.uint64))
print(f"B (replicas rebuilt by the group's builder threads): {bad} of {rounds} rounds with mismatches", flush=True)
bad = 0
for it in range(rounds):
    one = pkg.csa_wt(text=text)
    bad += report("C", it, np.asarray(one.count(pats, m)).astype(np.uint64))
    one.close()
print(f"C (single handle rebuilt on the main thread): {bad} of {rounds} rounds with mismatches", flush=True)
