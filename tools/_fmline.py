import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        e = json.loads(l)["extras"]
        print({k: round(v.get("GB/s") or v.get("Msa/s") or v.get("Gocc/s") or v.get("Mcount/s") or v.get("Gq/s"), 2) for k, v in e.items() if k.startswith(("fm_", "wt_"))})
