"""ONE numbers table for DESIGN.md, generated — no figure in it is typed by hand.
Sources: the committed line (profiles/bench_<tag>_line.json), its sidecar (profiles/bench_extras_<tag>.json) and the condensed counter
collection the line quotes (profiles/pmc_latest.json).  usage: python tools/numbers_table.py [tag = r06] > docs/design/NUMBERS.md"""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r06"
WRITE = "--write" in sys.argv  # docs/design/NUMBERS.md and the block between the NUMBERS markers of README.md
line = json.load(open(os.path.join(ROOT, "profiles", f"bench_{tag}_line.json")))
side = json.load(open(os.path.join(ROOT, "profiles", f"bench_extras_{tag}.json")))
ex = side.get("extras", side)
pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}


def f(x, nd=1):
    if x is None:
        return "—"
    if isinstance(x, bool):
        return "yes" if x else "NO"
    if isinstance(x, (int,)) and abs(x) >= 10000:
        return f"{x:,}".replace(",", " ")
    if isinstance(x, float):
        return f"{x:.{nd}f}"
    return str(x)


def roof(d):
    """(frac, bytes per unit, requests per unit, VALU share) of a leg's roofline block, whatever its shape"""
    r = d.get("roofline") or {}
    frac = r.get("frac", d.get("roofline_frac"))
    bpu = r.get("traffic_bytes_per_unit", r.get("traffic_bytes_per_pattern"))
    rpu = r.get("fabric_requests_per_unit", r.get("fabric_requests_per_pattern"))
    ft = d.get("fabric_traffic") or {}
    bpu = bpu if bpu is not None else ft.get("bytes_per_query")
    if ft.get("frac_of_hbm_peak") is not None:
        frac = ft.get("frac_of_hbm_peak")  # (measured traffic: the 8(d) model's figure can exceed 1 for a batch that reads each record once)
    return frac, bpu, rpu, r.get("valu_issue_share")


rows = []


def row(name, workload, rate, unit, ms, d, digest=None, cpu=None):
    frac, bpu, rpu, valu = roof(d) if isinstance(d, dict) else (None, None, None, None)
    rows.append(f"| {name} | {workload} | **{f(rate, 1)} {unit}** | {f(ms, 2)} | {f(frac, 2)} | {f(bpu, 0)} | {f(rpu, 1)} | {f(valu, 2)} | {f(digest)} | {cpu or '—'} |")


def cpu_of(d):
    c = d.get("cpu_baseline") if isinstance(d, dict) else None
    if not c:
        return None
    ns = c.get("ns_per_query")
    return f"{f(c.get('value'), 4)} {c.get('unit', '')} ({f(ns, 0)} ns, {c.get('cores', 1)} thread, {c.get('kind')}, equal: {f(c.get('matches_gpu'))})"


h = ex.get("headline", line)
rf = line.get("roofline", {})
cb = line.get("cpu_baseline") or {}
rows.append(f"| **bucketed rank_1** (headline: all kernels of a step) | {line['config']['workload'][:90]}… | **{f(line['value'], 1)} {line['unit']}** | {f(line['ms_per_step'], 2)} | "
            f"{f(rf.get('frac'), 3)} (algorithmic 96 B/query) | {f((rf.get('traffic') or 0) / line['config']['queries_per_step_per_gpu'], 1) if rf.get('traffic') else '—'} | — | — | {f(line.get('reference_digest_match'))} | "
            f"{f(cb.get('value'), 4)} {cb.get('unit', '')} ({cb.get('cores')} thread, {cb.get('kind')})" + " |")
dk = rf.get("direct_kernel") or (h.get("roofline", {}) or {}).get("direct_kernel") or {}
if dk:
    rows.append(f"| direct `k_rank` (same batch, same answers) | same | **{f(dk.get('Gq/s'), 1)} Grank/s** | — | {f(dk.get('frac'), 2)} | — | — | — | {f(dk.get('same_answers'))} | — |")
for key, name, wl in (("select_1", "bucketed select_1", "10^9 select_1, 2^34 bits"), ("rrr63_rank_1", "bucketed rrr_vector<63> rank_1", "10^9, 2^34 bits, 5 %"),
                      ("rrr63_select_1", "bucketed rrr_vector<63> select_1", "same")):
    d = ex.get(key)
    if d:
        row(name, wl, d.get("Gq/s"), "G/s", d.get("kernel_ms"), d, d.get("reference_digest_match"), cpu_of(d))
        k = d.get("direct_kernel")
        if k:
            rows.append(f"| … direct kernel | same | **{f(k.get('Gq/s'), 1)} G/s** | {f(k.get('kernel_ms'), 2)} | {f(k.get('frac', k.get('roofline_frac')), 2)} | — | — | — | {f(k.get('same_answers'))} | — |")
sd = ex.get("sd_vector")
if sd:
    rows.append(f"| `sd_vector<>` rank_1 / select_1 / select_0 | {f(sd.get('queries'))} queries, 2^{sd.get('universe_log2')} universe, {f(sd.get('ones'))} ones | **{f(sd.get('rank_1_Gq/s'))} / {f(sd.get('select_1_Gq/s'))} / {f(sd.get('select_0_Gq/s'))} G/s** | — | "
                f"{' / '.join(f(v.get('frac'), 2) for v in (sd.get('roofline') or {}).values() if isinstance(v, dict)) or '—'} | — | — | — | — | — |")
t = ex.get("text", {})
wl_t = f"{f(t.get('bytes'))} B text, σ {t.get('sigma')}, H0 {f(t.get('H0'), 2)}"
d = ex.get("wt_huff_rank")
if d:
    row("`wt_huff` rank(i, c) (`k_wt_rank_flat`)", f"{f(d.get('queries'))} queries, {wl_t}", d.get("Gq/s"), "G/s", d.get("kernel_ms"), d, d.get("reference_digest_match"), cpu_of(d))
d = ex.get("wt_huff_select")
if d:
    row("`wt_huff` select (bucketed)", "same", d.get("Gq/s"), "G/s", d.get("kernel_ms"), d, d.get("reference_digest_match"))
for key, name in (("fm_count_kmer8", "`count`, k-mer table k = 8"), ("fm_count", "`count`, default footprint"), ("fm_count_sa_dropped", "`count`, suffix array and text dropped"),
                  ("fm_count_lean", "`count` at 1.5 × SDSL's stream (`set_footprint`)")):
    d = ex.get(key)
    if d:
        row(name, f"{f(d.get('patterns'))} × {d.get('m')} B; index {f(d.get('index_bytes'))} B, k = {d.get('kmer_table', {}).get('k')}", d.get("Mcount/s"), "Mcount/s", d.get("kernel_ms"), d,
            d.get("reference_digest_match"), cpu_of(d))
rep = ex.get("fm_count_repetitive")
if rep:
    for key, name in (("count_default", "`count` on the REPETITIVE text, default footprint"), ("count_sa_dropped", "… suffix array and text dropped"), ("count_lean", "… at 1.5 × SDSL's stream")):
        d = rep.get(key)
        if d:
            row(name, f"{f(d.get('patterns'))} × {d.get('m')} B; mean count {f(rep.get('mean_count_of_a_20_byte_pattern'), 2)}; index {f(d.get('index_bytes'))} B, k = {d.get('kmer_table', {}).get('k')}",
                d.get("Mcount/s"), "Mcount/s", d.get("kernel_ms"), d, d.get("reference_digest_match"))
    d = rep.get("wt_rank")
    if d:
        row("`wt_huff` rank on the repetitive text", "10^8 queries", d.get("Gq/s"), "G/s", d.get("kernel_ms"), d, d.get("reference_digest_match"))
for key, name in (("fm_count_rrr63", "`count`, `csa_wt<wt_huff<rrr_vector<63>>>` as created from text"), ("fm_count_rrr63_lean", "… at 1.5 × the real library's stream of that type")):
    d = ex.get(key)
    if d:
        row(name, f"{f(d.get('patterns'))} × {d.get('m')} B; index {f(d.get('index_bytes'))} B = {f(d.get('x_sdsl_stream_bytes'), 2)} × the stream", d.get("Mcount/s"), "Mcount/s", d.get("kernel_ms"), d,
            d.get("same_answers_as_plain_index"))
d = ex.get("fm_sa_access_dens32")
if d:
    row("`csa[i]` on SA samples at 32 (`k_fm_walk`)", f"{f(d.get('queries'))} queries", d.get("Msa/s"), "Msa/s", d.get("ms"), d)
    a = d.get("at_dens_8_16")
    if a:
        rows.append(f"| … samples at 8 / 16 | same | **{f(a.get('Msa/s'), 0)} Msa/s** | {f(a.get('ms'), 2)} | — | — | — | — | — | — |")
d = ex.get("fm_locate_dens32")
if d:
    row("`locate` on the samples", f"{f(d.get('patterns'))} patterns, {f(d.get('occurrences'))} occurrences", d.get("Gocc/s"), "Gocc/s", d.get("ms"), d)
d = ex.get("fm_extract_64B")
if d:
    row("`extract`, 64-byte snippets (walks)", f"{f(d.get('snippets'))} snippets; long ranges {f(d.get('long_ranges_GB/s'))} GB/s; text resident {f(d.get('with_text_resident_GB/s'))} GB/s", d.get("GB/s"), "GB/s", d.get("ms"), d)
d = ex.get("beyond_2_32")
if d:
    rows.append(f"| index of {f(d.get('symbols'))} symbols | built from text in {f(d.get('build_from_text_s'))} s, {f(d.get('resident_GB'))} GB resident | " + " / ".join(f"{k}: {f(v)}" for k, v in d.items() if k.endswith('/s')) + " | — | — | — | — | — | — | — |")

_real_stdout = sys.stdout
sys.stdout = buf = io.StringIO()
print(f"# Numbers (generated by tools/numbers_table.py {tag}; do not edit)")
print()
print(f"Sources: `profiles/bench_{tag}_line.json` (the ONE line of `python bench.py`), `profiles/bench_extras_{tag}.json` (its sidecar), `profiles/pmc_latest.json` "
      f"(counter collection on kernel sources `{pmc.get('kernel_sources_sha')}`; a leg's fraction is null when it was not collected on the sources of the run).  "
      "`frac` = measured fabric bytes per unit × this run's rate ÷ 8 TB/s unless the row says otherwise; bytes and requests per unit are TCC_EA0 read + write requests "
      "(`tools/pmc_json.py`); `digest` = every answer of the batch compared with digests of the real sdsl-lite's answers (first 10^6–10^7 and every 100th of the whole batch).")
print()
print("| kernel / route | workload | rate | kernel ms | frac of 8 TB/s | fabric B per unit | requests per unit | VALU issue share | digest | CPU baseline (real library unless noted) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(r)
sw = ex.get("batch_sweep")
if sw:
    print()
    print("**Batch size × vector size** (`batch_sweep`; G rank/s, the three routes' answers compared):")
    print()
    print("| bits | queries | default (route) | direct | bucketed | same answers |")
    print("|---|---|---|---|---|---|")
    for e in sw:
        print(f"| 2^{e.get('n_bits_log2')} | {f(e.get('queries'))} | {f(e['default'].get('Grank/s'))} ({e['default'].get('route')}) | {f(e['direct'].get('Grank/s'))} | "
              f"{f(e['bucketed'].get('Grank/s'))} | {f(e.get('same_answers'))} |")
fv = ex.get("fm_count_vs_resident_bytes")
if fv:
    print()
    print(f"**`count` against resident bytes** (SDSL's `csa_wt<wt_huff<>, 32, 64>` stream: {f(fv.get('sdsl_stream_bytes'))} B):")
    print()
    print("| footprint | index bytes | × SDSL's stream | Mcount/s | k | digest | frac |")
    print("|---|---|---|---|---|---|---|")
    for r in fv.get("rows", []):
        print(f"| {r['name']} | {f(r['index_bytes'])} | {f(r['x_sdsl_stream_bytes'], 2)} | {f(r['Mcount/s'], 0)} | {r['kmer_k']} | {f(r['reference_digest_match'])} | {f(r.get('roofline_frac'), 2)} |")
ph = (h.get("roofline") or {}).get("phases_ms")
if ph:
    print()
    print("**Where the headline's step goes** (ms, medians): " + ", ".join(f"{k} {f(v, 2)}" for k, v in ph.items()))

sys.stdout = _real_stdout
text = buf.getvalue()
if not WRITE:
    sys.stdout.write(text)
else:
    with open(os.path.join(ROOT, "docs", "design", "NUMBERS.md"), "w") as fo:
        fo.write(text)
    rd = os.path.join(ROOT, "README.md")
    r = open(rd).read()
    a, b = "<!-- NUMBERS:BEGIN (tools/numbers_table.py --write) -->", "<!-- NUMBERS:END -->"
    if a in r and b in r:
        main_table = text[text.index("| kernel / route |"):]
        main_table = main_table[:main_table.index("\n\n")] if "\n\n" in main_table else main_table
        r = r[:r.index(a) + len(a)] + "\n" + main_table + "\n" + r[r.index(b):]
        open(rd, "w").write(r)
    print("wrote docs/design/NUMBERS.md" + (" and README.md's table" if a in r else ""))
