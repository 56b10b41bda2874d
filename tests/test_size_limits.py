"""CPU: the size gates of the library (sdsl-lite_amd/csrc/limits.hpp, reported by sdsl_hip_limit) against the table of INTEGRATION.md 3b —
every row's "from" value must be the number the library was compiled with, every name the library knows must have a row, and every
test file a row cites must exist.  The reference has no such gates (64-bit size_type throughout: wt_pc.hpp:366-474)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("bv_bits", "bv_bucketed_bits", "rrr_bits", "rrr_bucketed_bits", "wt_fused_symbols", "wt_select_bucketed_symbols",
         "step_table_lines", "fm_fast_symbols", "sorter32_symbols", "sorter64_symbols")


def table_rows():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    sec = text[text.index("## 3b. Size limits"):text.index("## 4. Multi-GPU")]
    rows = {}
    for line in sec.splitlines():
        m = re.match(r"\| `([a-z0-9_]+)` \|", line)
        if m:
            rows[m.group(1)] = [c.strip() for c in line.strip().strip("|").split("|")]
    return rows


def test_the_table_lists_what_the_library_was_compiled_with(pkg):
    L = pkg.capi.lib()
    rows = table_rows()
    assert set(rows) == set(NAMES), "one row per gate"
    for name in NAMES:
        value = L.sdsl_hip_limit(name.encode())
        assert value, name
        numbers = [int(x) for x in re.findall(r"\b\d{6,}\b", rows[name][2])]
        assert value in numbers, f"{name}: the library says {value}, INTEGRATION.md 3b says {rows[name][2]!r}"
    assert L.sdsl_hip_limit(b"no_such_gate") == 0


def test_cited_tests_exist():
    for name, cells in table_rows().items():
        for path in re.findall(r"tests/[A-Za-z0-9_]+\.py", cells[4]):
            assert os.path.exists(os.path.join(ROOT, path)), f"{name}: {path}"
        for path, fn in re.findall(r"(tests/[A-Za-z0-9_]+\.py)`?::(test_[A-Za-z0-9_]+)", cells[4]):
            assert ("def " + fn) in open(os.path.join(ROOT, path)).read(), f"{name}: {path}::{fn}"
        for fn in re.findall(r"`::(test_[A-Za-z0-9_]+)`", cells[4]):
            assert any(("def " + fn) in open(os.path.join(ROOT, "tests", f)).read() for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py")), fn


def test_the_power_of_two_gates_are_what_the_kernels_pack():
    """a few of the constants against the field widths they stand for (so that a widened field and a forgotten gate cannot drift apart)"""
    import importlib
    pkg = importlib.import_module("sdsl-lite_amd")
    L = pkg.capi.lib()
    lim = {n: L.sdsl_hip_limit(n.encode()) for n in NAMES}
    assert lim["wt_fused_symbols"] == 1 << 36 and lim["wt_select_bucketed_symbols"] == 1 << 32
    assert lim["fm_fast_symbols"] == 1 << 39 and lim["sorter32_symbols"] == (1 << 32) - 2
    assert lim["bv_bucketed_bits"] == (1 << 29) * 448 and lim["rrr_bucketed_bits"] == (1 << 24) * 34 * 63
    assert lim["bv_bits"] == 1 << 40 == lim["rrr_bits"] == lim["sorter64_symbols"]
    assert lim["step_table_lines"] == 1 << 28
    assert lim["bv_bucketed_bits"] < lim["bv_bits"] and lim["wt_fused_symbols"] < lim["fm_fast_symbols"]
