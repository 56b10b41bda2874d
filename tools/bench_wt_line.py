import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); e = d["extras"]
        print({k: (round(e[k].get("Gq/s"), 2), e[k].get("reference_digest_match")) for k in e if k.startswith("wt")})
