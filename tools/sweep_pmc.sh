#!/bin/bash
# fabric write / read requests of the pass-2 kernels with and without the one-sweep pass (hand tool for gpurun)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/sweep_pmc
mkdir -p $O
cd /tmp
for sw in 1 0; do
  SDSL_HIP_SORTED_SWEEP=$sw SDSL_HIP_RANK_SORTED=1 timeout 280 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum -d $O/sw$sw -o p --output-format csv -- python $R/tools/sorted_rank_probe.py 34 1000000000 child > $O/sw$sw.log 2>&1
  echo "sw=$sw exit=$?"
done
find $O -name "*.db" -delete
cd $R
python - <<'PY'
import csv, glob, collections
for sw in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/sweep_pmc/sw{sw}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_sr_partition" in k or "k_sr_hist" in k:
                agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(sw, k, {c: f"{sum(l)/len(l):.4g}" for c, l in v.items()})
PY
