"""The driver's line of bench.py: a COMPACT object (< 4 KB, one line, strict JSON) assembled from the run's full record.

The full record -- every leg with its notes, the per-kernel tables, the side workloads' own lines -- goes to a file (`--full-json`, default
<workdir>/bench_full_<workload>.json; the builder's GPU scripts copy it into profiles/) and to one-line summaries on stderr.  The reference's unit
of report is one short line per chunk (bwamem.cpp:1386-1388: "Processed N reads in X CPU sec, Y real sec"); round 5's line had grown to 19.9 KB and the
driver's record of that round carries no parsed value.  tests/test_bench_line.py pins the size, the strictness and the keys on canned records.
"""
import json
import math
import os

MAX_LINE = 4096                 # hard bound of the printed line (bytes)
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_step")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _num(x, sig=6):
    """floats to `sig` significant digits (a line of measurements, not of doubles); NaN / inf -> None (strict JSON has neither)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return None
        if x == 0.0:
            return 0.0
        if abs(x) >= 1e15:
            return x
        r = float("%.*g" % (sig, x))
        return int(r) if r == int(r) and abs(r) >= 1e6 else r
    try:                                                  # numpy scalars
        import numpy as np
        if isinstance(x, np.integer):
            return int(x)
        if isinstance(x, np.floating):
            return _num(float(x), sig)
        if isinstance(x, np.bool_):
            return bool(x)
    except Exception:                                     # noqa
        pass
    return x


def _clean(o, sig=6):
    if isinstance(o, dict):
        return {str(k): _clean(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v, sig) for v in o]
    return _num(o, sig)


def _short(s, n):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    return {k: d.get(k) for k in keys if k in d} if isinstance(d, dict) else None


def compact(full, full_path=None):
    """full record of a run (the dict bench.py builds: pe150, ont2d or bsw workload) -> the object of the driver's line"""
    f = full
    cfg = f.get("config") or {}
    out = {k: f.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": _short(cfg.get("workload_short") or cfg.get("workload"), 200)}
    for k in ("reads_per_gpu_per_step", "read_len", "genome_mbp", "resident_chunks", "pairs_per_gpu_per_step", "steps_cover_reads"):
        if cfg.get(k) is not None:
            out["config"][k] = cfg[k]
    out["config"]["parallelism"] = _short(cfg.get("parallelism_short") or cfg.get("parallelism"), 120)
    out["ranks"] = f.get("n_gpus")
    if f.get("value_scope"):
        out["value_scope"] = _short(f.get("value_scope_short") or f["value_scope"], 120)
    e2 = f.get("end_to_end") if isinstance(f.get("end_to_end"), dict) else None
    if "value_end_to_end" in f or e2:
        out["value_end_to_end"] = f.get("value_end_to_end")
    if e2:
        if e2.get("value"):
            ee = {"frac_of_hot_path": e2.get("frac_of_hot_path"), "chunks": e2.get("chunks"), "host_cpu_s_per_chunk": e2.get("host_cpu_s_per_chunk"),
                  "steady_reads_per_s": (e2.get("steady_state") or {}).get("reads_per_s") if isinstance(e2.get("steady_state"), dict) else None,
                  "equal_to_serial_run": (e2.get("chunk_check") or {}).get("equal_to_serial_run")}
            out["end_to_end"] = {k: v for k, v in ee.items() if v is not None}
        else:
            out["end_to_end"] = {k: _short(e2[k], 100) for k in ("skipped", "error") if k in e2}
    if f.get("stage_ms_per_step"):
        out["stage_ms_per_step"] = {k: _num(float(v), 4) for k, v in f["stage_ms_per_step"].items()}
    r = f.get("roofline") or {}
    ro = _pick(r, ROOFLINE_KEYS) or {}
    for k in ROOFLINE_KEYS:
        ro.setdefault(k, None)
    ro["kernel"] = _short(ro.get("kernel"), 60)
    if r.get("traffic_source"):
        ro["traffic_source"] = _short(r["traffic_source"], 60)
    if isinstance(r.get("seeding_stage"), dict):
        ro["stage_frac"] = r["seeding_stage"].get("frac")
    if isinstance(r.get("fm_index_kernels"), dict):
        ro["fm_index_kernels"] = {_short(k, 30): _num(v.get("frac"), 3) for k, v in r["fm_index_kernels"].items() if isinstance(v, dict)}
    if r.get("frac_of_random_line_ceiling") is not None:
        ro["frac_of_random_line_ceiling"] = r["frac_of_random_line_ceiling"]
    out["roofline"] = ro
    cb = f.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        c = _pick(cb, CPU_KEYS)
        c["sample"] = _short(cb.get("sample_short") or cb.get("sample"), 160)
        if cb.get("hot_path_value") is not None:
            c["hot_path_value"] = cb["hot_path_value"]
        out["cpu_baseline"] = c
    elif isinstance(cb, dict):
        out["cpu_baseline"] = {k: _short(cb[k], 100) for k in ("skipped", "error") if k in cb} or None
    else:
        out["cpu_baseline"] = None
    p = f.get("parity")
    if isinstance(p, dict):
        pk = ("regs_equal", "fin_equal", "sam_equal", "sam_records", "reads", "regs", "regs_over_2p32", "pairs_equal", "pairs", "mismatches")
        out["parity"] = {k: p[k] for k in pk if k in p}
        for k in ("skipped", "error"):
            if k in p:
                out["parity"][k] = _short(p[k], 100)
    else:
        out["parity"] = None
    ek = f.get("extend_kernel")
    if isinstance(ek, dict):
        out["extend_kernel"] = {k: ek[k] for k in ("gcups", "stage_ms", "avg_launch_ms", "valu_frac", "lds_conflict_frac", "pmc_stage_ms") if ek.get(k) is not None}
    wpr = f.get("work_per_read")
    if isinstance(wpr, dict):
        out["work_per_read"] = {k: _num(float(v), 4) for k, v in wpr.items() if v is not None}
    for key in ("index_replica_gb", "pairs_per_s"):
        if f.get(key) is not None:
            out[key] = f[key]
    c5 = f.get("config5")
    if isinstance(c5, dict):
        if c5.get("value"):
            p5 = c5.get("parity") or {}
            out["config5"] = {"value": c5["value"], "unit": c5.get("unit"), "reads_per_step": (c5.get("config") or {}).get("reads_per_gpu_per_step"), "steps": c5.get("steps"),
                              "reads_in_run": (c5.get("config") or {}).get("steps_cover_reads"),
                              "ms_per_step": c5.get("ms_per_step"), "regs_equal": p5.get("regs_equal"), "fin_equal": p5.get("fin_equal"), "sam_equal": p5.get("sam_equal"),
                              "gate_reads": p5.get("reads"), "cpu_reads_per_s": (c5.get("cpu_baseline") or {}).get("value"), "exit_code": c5.get("exit_code")}
        else:
            out["config5"] = {k: _short(c5[k], 100) for k in ("skipped", "error") if k in c5}
    c2 = f.get("config2")
    if isinstance(c2, dict):
        if c2.get("value"):
            s1 = c2.get("s1_binding") or {}
            out["config2"] = {"gcups": (c2.get("extend_kernel") or {}).get("gcups"), "pairs_equal": (c2.get("parity") or {}).get("pairs_equal"),
                              "s1_bm2_chunk_s": (s1.get("bm2s1") or {}).get("chunk_real_s"), "s1_reference_chunk_s": (s1.get("reference") or {}).get("chunk_real_s"),
                              "s1_sam_equal": s1.get("sam_equal"), "exit_code": c2.get("exit_code")}
        else:
            out["config2"] = {k: _short(c2[k], 100) for k in ("skipped", "error") if k in c2}
    s1 = f.get("s1_binding")
    if isinstance(s1, dict) and (s1.get("bm2s1") or s1.get("reference")):
        out["s1_binding"] = {"bm2_chunk_s": (s1.get("bm2s1") or {}).get("chunk_real_s"), "reference_chunk_s": (s1.get("reference") or {}).get("chunk_real_s"), "sam_equal": s1.get("sam_equal")}
    bd = f.get("binding")
    if isinstance(bd, dict):
        if bd.get("bm2_wall_s"):
            out["binding"] = {"reads": bd.get("reads"), "bm2_wall_s": bd.get("bm2_wall_s"), "steady_reads_per_s": bd.get("reads_per_s_bm2_steady_chunks"),
                              "reference_reads_per_s": bd.get("reads_per_s_reference_chunks"), "sam_equal": bd.get("sam_equal"), "sam_records_compared": bd.get("sam_records_compared")}
        else:
            out["binding"] = {k: _short(bd[k], 100) for k in ("skipped", "error") if k in bd}
    if f.get("knobs"):
        out["knobs"] = {k: _short(v, 24) for k, v in list(f["knobs"].items())[:12]}
    if full_path:
        out["full_record"] = full_path
    return _clean(out)


DROP_ORDER = ("knobs", "work_per_read", "binding", "s1_binding", "config2", "config5", "extend_kernel", "index_replica_gb", "value_scope", "stage_ms_per_step", "end_to_end")


def line(full, full_path=None):
    """-> the one line (no newline inside, strict JSON, < MAX_LINE bytes).  Optional objects leave in DROP_ORDER should a run ever overflow the bound;
    the contract's keys, `roofline`, `cpu_baseline` and `parity` never do."""
    obj = compact(full, full_path)
    s = json.dumps(obj, allow_nan=False, separators=(",", ":"))
    for k in DROP_ORDER:
        if len(s.encode()) < MAX_LINE:
            break
        if k in obj:
            del obj[k]
            obj["dropped"] = obj.get("dropped", []) + [k]
            s = json.dumps(obj, allow_nan=False, separators=(",", ":"))
    if len(s.encode()) >= MAX_LINE:                      # (cannot happen with the bounded strings above; never print an unparseable line)
        obj = {k: obj.get(k) for k in REQUIRED + ("parity", "ranks", "value_end_to_end")}
        obj["roofline"] = _pick(obj.get("roofline") or {}, ROOFLINE_KEYS)
        s = json.dumps(obj, allow_nan=False, separators=(",", ":"))
    return s


def write_full(full, path):
    """the whole record, pretty enough to diff; -> path or None"""
    if not path:
        return None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as fh:
            json.dump(_clean(full, 9), fh, indent=1, allow_nan=False)
            fh.write("\n")
        os.replace(tmp, path)
        return path
    except (OSError, ValueError) as e:
        import sys
        print("[bench] full record not written to %s: %s" % (path, e), file=sys.stderr, flush=True)
        return None


def emit(full, path, stream=None):
    """write the full record to `path`, print the compact line LAST on stdout"""
    import sys
    p = write_full(full, path)
    s = line(full, p)
    print(s, file=stream or sys.stdout, flush=True)
    return s
