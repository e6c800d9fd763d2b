"""The driver's line of bench.py (tools/bench_line.py): compact, strict JSON, the contract's keys -- checked on canned full records (round 5's own
19.9 KB line, which the driver could not parse, is one of them) without a GPU."""
import glob
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_line  # noqa: E402

CANNED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05zzz_bench*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r04zz_bench_as_driver.json"))
                + glob.glob(os.path.join(ROOT, "profiles", "r05*_bench_bsw*.json"))[:1] + glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_full*.json")))


def _strict(s):
    def bad(c):
        raise ValueError("non-strict constant " + c)
    return json.loads(s, parse_constant=bad)


def test_canned_records_exist():
    assert any("r05zzz_bench.json" in c for c in CANNED)


@pytest.mark.parametrize("path", CANNED, ids=[os.path.basename(c) for c in CANNED])
def test_line_of_a_real_record(path):
    full = json.load(open(path))
    s = bench_line.line(full, "/tmp/x/bench_full_pe150.json")
    assert "\n" not in s
    assert len(s.encode()) < 4096, len(s)
    d = _strict(s)
    for k in bench_line.REQUIRED:
        assert k in d, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"):
        assert d[k] == pytest.approx(full[k], rel=1e-5) if isinstance(full[k], float) else d[k] == full[k], k
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    for k in bench_line.ROOFLINE_KEYS:
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert d["roofline"]["achieved"] / d["roofline"]["peak"] == pytest.approx(d["roofline"]["frac"], rel=1e-4)
    if isinstance(full.get("cpu_baseline"), dict) and full["cpu_baseline"].get("value"):
        for k in ("value", "unit", "cores", "kind"):
            assert d["cpu_baseline"][k] == pytest.approx(full["cpu_baseline"][k], rel=1e-5) if isinstance(full["cpu_baseline"][k], float) else d["cpu_baseline"][k] == full["cpu_baseline"][k]
    if isinstance(full.get("parity"), dict) and "regs_equal" in full["parity"]:
        for k in ("regs_equal", "fin_equal", "sam_equal", "sam_records"):
            assert d["parity"][k] == full["parity"][k]
    if full.get("value_end_to_end"):
        assert d["value_end_to_end"] == pytest.approx(full["value_end_to_end"], rel=1e-5)
    assert d["ranks"] == full["n_gpus"]


def test_round5_record_shrinks_and_keeps_its_legs():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05zzz_bench.json")))
    assert len(json.dumps(full)) > 15000                     # the line the driver could not parse
    d = _strict(bench_line.line(full))
    assert d["end_to_end"]["frac_of_hot_path"] == pytest.approx(full["end_to_end"]["frac_of_hot_path"], rel=1e-5)
    assert d["end_to_end"]["host_cpu_s_per_chunk"] == pytest.approx(full["end_to_end"]["host_cpu_s_per_chunk"], rel=1e-5)
    assert d["config5"]["value"] == pytest.approx(full["config5"]["value"], rel=1e-5)
    assert d["config2"]["gcups"] == pytest.approx(full["config2"]["extend_kernel"]["gcups"], rel=1e-5)
    assert d["binding"]["sam_equal"] == full["binding"]["sam_equal"]
    assert set(d["stage_ms_per_step"]) == set(full["stage_ms_per_step"])
    assert "dropped" not in d


def test_non_finite_numbers_and_hostile_sizes():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05zzz_bench.json")))
    full["roofline"]["traffic"] = float("nan")
    full["value_end_to_end"] = float("inf")
    full["config"]["workload"] = "x" * 50000
    full["cpu_baseline"]["sample"] = "y" * 50000
    full["knobs"] = {"BM2_K%d" % i: "v" * 500 for i in range(200)}
    full["parity"]["error"] = "z" * 10000
    full["stage_ms_per_step"] = {"stage%d" % i: float(i) for i in range(400)}      # cannot fit: it leaves, the contract's keys stay
    s = bench_line.line(full)
    assert len(s.encode()) < 4096
    d = _strict(s)
    assert d["roofline"]["traffic"] is None and d["value_end_to_end"] is None
    assert "stage_ms_per_step" in d["dropped"]
    for k in bench_line.REQUIRED:
        assert k in d


def test_legs_that_did_not_run():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05zzz_bench.json")))
    full["parity"] = {"skipped": "time budget"}
    full["cpu_baseline"] = {"error": "no reference binary"}
    full["end_to_end"] = {"error": "stuck"}
    full["value_end_to_end"] = None
    full["config5"] = {"skipped": "time budget"}
    full["config2"] = None
    full["binding"] = None
    d = _strict(bench_line.line(full))
    assert d["parity"] == {"skipped": "time budget"} and d["cpu_baseline"] == {"error": "no reference binary"}
    assert d["end_to_end"] == {"error": "stuck"} and d["config5"] == {"skipped": "time budget"}
    assert "config2" not in d and "binding" not in d


def test_emit_writes_the_full_record_and_prints_the_line_last(tmp_path):
    full = json.load(open(os.path.join(ROOT, "profiles", "r05zzz_bench.json")))
    out = io.StringIO()
    p = str(tmp_path / "sub" / "bench_full.json")
    s = bench_line.emit(full, p, stream=out)
    assert out.getvalue() == s + "\n"
    back = json.load(open(p))
    assert back["roofline"]["note"] == full["roofline"]["note"] and back["config5"]["chain_kernel"]
    assert _strict(s)["full_record"] == p


def test_bench_main_ends_in_emit():
    """bench.py prints nothing on stdout but the compact line (its legs log to stderr)"""
    src = open(os.path.join(ROOT, "bench.py")).read() + open(os.path.join(ROOT, "tools", "bench_legs.py")).read()
    import re
    prints = [m.group(0) for m in re.finditer(r"^\s*print\(.*$", src, re.M)]
    assert all("file=sys.stderr" in p for p in prints), prints
    assert src.count("bench_line.emit(") == 2                # the pe150 / ont2d line and config 2's
