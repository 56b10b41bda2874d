"""Per-kernel sums of the counter csv files written by tools/sq_profile.sh."""
import csv, glob, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_\w+(<[^>(]*>)?", row["Kernel_Name"])
        k = m.group(0) if m else row["Kernel_Name"].split("(")[0][:70]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:16]:
    print(k)
    for c in sorted(acc[k]):
        n = len(calls[k][c])
        print(f"   {c:28s} {acc[k][c]:.4g}  ({n} dispatches, {acc[k][c]/n:.4g} each)")
