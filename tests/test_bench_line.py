"""The line bench.py prints must stay small enough for the driver to parse (round 4: a 23 KB line came back with `parsed: null` and the
round's headline was lost).  CPU: the composer of the line on the fattest blocks a run has produced (the committed round-4 line's
extras) and on hostile ones; the GPU tests (tests/test_gpu_bench.py) assert the same on real runs, plus that it is the LAST stdout line."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _r04():
    d = json.load(open(os.path.join(ROOT, "profiles", "bench_r04_full_line.json")))
    ex = d.pop("extras")
    ex["end_to_end"] = d.pop("end_to_end")
    d.pop("secondary", None)
    return d, ex


def test_line_is_compact_and_complete():
    b = _bench()
    result, ex = _r04()
    txt = b.compact_line(result, ex, os.path.join(ROOT, "bench_extras.json"), 1024)
    assert len(txt.encode()) < b.LINE_LIMIT == 4096 and "\n" not in txt
    d = json.loads(txt)
    assert all(k in d for k in REQUIRED)
    assert d["value"] == result["value"] and d["ms_per_step"] == result["ms_per_step"]  # full precision: the driver checks one against the other
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert d["secondary"]["unit"] == "Mcount/s" and d["secondary"]["value"] > 0 and d["secondary"]["index_bytes"] > 0
    assert d["end_to_end"]["pinned_Grank/s"] > 0 and d["extras_file"] == "bench_extras.json"
    assert d["summary"]["select_1"]["Gq/s"] > 0


def test_line_sheds_optional_blocks_before_it_grows_past_the_limit():
    b = _bench()
    result, ex = _r04()
    for i in range(400):  # four hundred more legs
        ex["leg_%d" % i] = {"Gq/s": 1.0}
    result["roofline"]["kernel"] = "k" * 1500
    txt = b.compact_line(result, ex, os.path.join(ROOT, "bench_extras.json"), 1024)
    d = json.loads(txt)
    assert len(txt.encode()) < 4096 and all(k in d for k in REQUIRED) and d["secondary"]["value"] > 0


def test_no_legs_no_sidecar_pointer():
    b = _bench()
    result, _ = _r04()
    d = json.loads(b.compact_line(result, {}, "x.json", 1024))
    assert d["extras_file"] is None and d["secondary"] is None and d["extras_error"] is None
