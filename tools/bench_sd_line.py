import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); e = d["extras"]
        print({k: v for k, v in e["sd_vector"].items() if "Gq" in k})
        if "select_shapes" in e:
            print({sh: {k: round(v["Gq/s"], 1) for k, v in row.items() if isinstance(v, dict) and "Gq/s" in v} for sh, row in e["select_shapes"]["shapes"].items()})
