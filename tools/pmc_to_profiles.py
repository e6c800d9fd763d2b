#!/usr/bin/env python3
"""Turn the rocprofv3 summaries of one tools/gpu/run_full.sh run (gpurun_out/<tag>/) into the committed, judged files under profiles/:
   r02_kernel_trace.md, r02_pmc_fetch.md, r02_pmc_write.md, r02_pmc_sq1.md, r02_pmc_sq2.md   (copies of the summaries)
   r02_k_bwd_pmc.json     HBM bytes per k_bwd launch (FETCH_SIZE + WRITE_SIZE, separate passes)        -> bench.py roofline.traffic
   r02_ext_pmc_sq.json    VALU issue fraction and LDS bank-conflict fraction of k_ext_lanes (SQ counters) -> bench.py extend_kernel
   r02_bench.json / r02_bench_ont2d.json   the bench lines of the run
python tools/pmc_to_profiles.py gpurun_out/<tag> [round-prefix]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    """-> {kernel: {counter: (sum, dispatches)}} and {kernel: (calls, total_ms)} from a tools/rocpd_summary.py file"""
    pmc, calls = {}, {}
    sect = 0
    for l in open(path):
        if l.startswith("## PMC"):
            sect = 1
            continue
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        if len(c) < 4 or c[0] in ("kernel", "---"):
            continue
        try:
            if sect == 0:
                calls[c[0]] = (int(c[1]), float(c[2]))
            else:
                pmc.setdefault(c[0], {})[c[1]] = (float(c[2]), int(c[3]))
        except ValueError:
            pass
    return pmc, calls


def ont2d(src, pre):
    """config 5's own counter pass: gpurun_out/<tag>/bench_ont2d.json (the pass's own bench line) + pmc_sq1_ont2d.md -> profiles/<pre>_ont2d_pmc_sq.md and
    profiles/<pre>_ont2d_ext_pmc_sq.json (what bench.py --workload ont2d reports as extend_kernel.valu_frac) with the chaining kernels' counters beside."""
    dst = os.path.join(ROOT, "profiles")
    bench = json.load(open(os.path.join(src, "bench_ont2d_pmc.json")))
    shutil.copy(os.path.join(src, "pmc_sq1_ont2d.md"), os.path.join(dst, "%s_ont2d_pmc_sq.md" % pre))
    s1, calls = counters(os.path.join(src, "pmc_sq1_ont2d.md"))
    steps = float(os.environ.get("PMC_STEPS", 2))
    fam = {}
    for k, v in s1.items():
        name = "k_ext_wave" if "k_ext_wave" in k else "k_ext_seeds" if "k_ext_seeds" in k else "k_chain_islands" if "k_chain_islands" in k else \
               "k_chain_heavy" if "k_chain_heavy" in k else "k_seed_sw" if "k_seed_sw" in k else "k_walk" if "k_walk" in k else None
        if name:
            for cn, (val, _) in v.items():
                fam.setdefault(name, {})[cn] = fam.setdefault(name, {}).get(cn, 0.0) + val
    peak, src_peak = 256 * 4 * 2.4e9 / 4.0, "nominal: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"
    for fn in sorted(os.listdir(dst), reverse=True):
        if fn.endswith("valu_int_ubench.txt"):
            rates = [float(m.group(1)) for m in re.finditer(r"([\d.]+) G wave-instr/s", open(os.path.join(dst, fn)).read())]
            if rates:
                peak, src_peak = max(rates) * 1e9, "profiles/" + fn
                break
    stage_ms = bench["stage_ms_per_step"]["extend"]
    ext_insts = sum(fam.get(k, {}).get("SQ_INSTS_VALU", 0.0) for k in ("k_ext_wave", "k_ext_seeds")) / steps
    out = {"kernel": "k_ext_wave (sliding register window) + k_ext_seeds, config 5", "workload": {"reads_per_gpu_per_step": bench["config"]["reads_per_gpu_per_step"], "genome_mbp": bench["config"]["genome_mbp"], "read_len": None},
           "extend_stage_ms": stage_ms, "valu_wave_insts_per_step": ext_insts, "valu_frac": ext_insts / (stage_ms * 1e-3) / peak, "valu_peak_wave_insts_per_s": peak, "valu_peak_source": src_peak,
           "lds_conflict_frac": 0.0,
           "per_kernel": {k: {"valu_wave_insts_per_step": c.get("SQ_INSTS_VALU", 0.0) / steps, "valu_busy_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                              "wait_any_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None} for k, c in fam.items()},
           "source": "rocprofv3 --pmc SQ_* --kernel-trace on `bench.py --workload ont2d` (profiles/%s_ont2d_pmc_sq.md); the issue fraction over the stage's wall time of that run" % pre}
    json.dump(out, open(os.path.join(dst, "%s_ont2d_ext_pmc_sq.json" % pre), "w"), indent=1)
    print(json.dumps(out, indent=1))


def ont2d_fetch(src, pre):
    """config 5's FETCH pass: gpurun_out/<tag>/pmc_fetch_ont2d.md + bench_ont2d_fetch.json -> profiles/<pre>_ont2d_k_walk_pmc.json (HBM bytes per k_walk<1> launch,
    what `bench.py --workload ont2d` reports as roofline.traffic; FETCH_SIZE at face value: isolated 64-byte lines, profiles/r01_randline_ubench.txt)"""
    dst = os.path.join(ROOT, "profiles")
    bench = json.load(open(os.path.join(src, "bench_ont2d_fetch.json")))
    shutil.copy(os.path.join(src, "pmc_fetch_ont2d.md"), os.path.join(dst, "%s_ont2d_pmc_fetch.md" % pre))
    f, _ = counters(os.path.join(src, "pmc_fetch_ont2d.md"))
    tot, nd = 0.0, 0
    for k, v in f.items():
        if "k_walk<1>" in k and "FETCH_SIZE" in v:
            tot += v["FETCH_SIZE"][0]; nd += v["FETCH_SIZE"][1]
    out = {"kernel": "k_walk<1>", "workload": {"genome_mbp": bench["config"]["genome_mbp"], "reads_per_gpu_per_step": bench["config"]["reads_per_gpu_per_step"], "read_len": None},
           "fetch_size_kib_per_launch": tot / nd, "hbm_bytes_per_launch": tot / nd * 1024.0,
           "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace on `bench.py --workload ont2d` (profiles/%s_ont2d_pmc_fetch.md): %d dispatches (the steady-state half)" % (pre, nd)}
    json.dump(out, open(os.path.join(dst, "%s_ont2d_k_walk_pmc.json" % pre), "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--ont2d-fetch":
        return ont2d_fetch(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r05")
    if sys.argv[1] == "--ont2d":
        return ont2d(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r04")
    src = sys.argv[1]
    pre = sys.argv[2] if len(sys.argv) > 2 else "r02"
    dst = os.path.join(ROOT, "profiles")
    for a, b in (("kernel_trace.md", "kernel_trace.md"), ("pmc_fetch.md", "pmc_fetch.md"), ("pmc_write.md", "pmc_write.md"),
                 ("pmc_sq1.md", "pmc_sq1.md"), ("pmc_sq2.md", "pmc_sq2.md"), ("bench.json", "bench.json"), ("bench_ont2d.json", "bench_ont2d.json"),
                 ("bench_parity.json", "bench_parity.json"), ("sweep.json", "sweep.json"), ("pytest.log", "pytest_subset.log")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (pre, b)))
    bench = json.load(open(os.path.join(src, "bench.json")))
    wl = {"genome_mbp": bench["config"]["genome_mbp"], "reads_per_gpu_per_step": bench["config"]["reads_per_gpu_per_step"],
          "read_len": bench["config"]["read_len"]}
    f, fc = counters(os.path.join(src, "pmc_fetch.md"))
    w, _ = counters(os.path.join(src, "pmc_write.md"))
    def bwd(d, counter):                                          # k_bwd is a template since round 2: "void k_bwd<12>", "void k_bwd5<6>"
        tot, nd = 0.0, 0                                          # (k_bwd_cont -- the tasks k_bwd hands over, round 5 -- belongs to its k_bwd launch: its bytes count, its dispatches do not)
        for k, v in d.items():
            if "k_bwd" in k and "heavy" not in k and counter in v:
                tot += v[counter][0]
                if "k_bwd_cont" not in k:
                    nd += v[counter][1]
        return tot, nd
    fs, nd = bwd(f, "FETCH_SIZE")
    ws, _ = bwd(w, "WRITE_SIZE")
    out = {"kernel": "k_bwd (+ k_bwd_cont)", "workload": wl, "fetch_size_kib_per_launch": fs / nd, "write_size_kib_per_launch": ws / nd,
           "hbm_bytes_per_launch": (fs + ws) / nd * 1024.0,
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_fetch.md, %s_pmc_write.md): sums over %d "
                     "dispatches divided by %d.  FETCH_SIZE is in KiB and was calibrated at factor 1.00 on isolated 64-byte lines with "
                     "tools/ubench/randline.hip (profiles/r01_randline_ubench.txt; the x2 of the guide applies to wide streaming reads, which "
                     "this kernel does not make); WRITE_SIZE is uncalibrated (2 %% of the total)." % (pre, pre, nd, nd)}
    json.dump(out, open(os.path.join(dst, "%s_k_bwd_pmc.json" % pre), "w"), indent=1)
    s1, c1 = counters(os.path.join(src, "pmc_sq1.md"))
    tot = {}
    per = {}
    for k, v in s1.items():
        if "k_ext_lanes" in k or "k_ext_seeds" in k or "k_ext_wave" in k:             # both kernels of the stage: lane-per-seed classes and wavefront-per-seed classes
            fam = "k_ext_wave" if "k_ext_wave" in k else "k_ext_seeds"
            for cn, (val, _) in v.items():
                tot[cn] = tot.get(cn, 0.0) + val
                per.setdefault(fam, {})[cn] = per.setdefault(fam, {}).get(cn, 0.0) + val
    steps = float(os.environ.get("PMC_STEPS", 2))                # the PMC passes run `--steps 1 --warmup 1`
    stage_ms = bench["stage_ms_per_step"]["extend"]
    # (NOT the stage time of the counter pass itself: under --pmc the profiler runs one dispatch at a time, the eight concurrent launches of a phase one
    #  after the other -- 43 ms instead of 16 in round 5's pass.  The instruction count of a step does not depend on that; the time it is divided by
    #  is the stage time of the un-profiled bench run of the same call, src/bench.json.)
    # VALU issue: wave-instructions per second against the rate MEASURED on this GPU by tools/ubench/valu_int.hip (the best line of the
    # committed run: independent v_add_u32 + v_max_i32 chains), and against the nominal 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
    insts = tot["SQ_INSTS_VALU"] / steps
    peak4 = 256 * 4 * 2.4e9 / 4.0
    measured_peak, measured_src = None, None
    for fn in sorted(os.listdir(dst), reverse=True):
        if fn.endswith("valu_int_ubench.txt"):
            rates = [float(m.group(1)) for m in re.finditer(r"([\d.]+) G wave-instr/s", open(os.path.join(dst, fn)).read())]
            if rates:
                measured_peak, measured_src = max(rates) * 1e9, "profiles/" + fn
                break
    ext = {"kernel": "k_ext_seeds<P8, PF, PT, G4> + k_ext_wave (the extension stage)", "workload": wl,
           "per_kernel": {f: {"valu_wave_insts_per_step": c["SQ_INSTS_VALU"] / steps, "valu_busy_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                              "wait_any_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "lds_insts_per_step": c["SQ_INSTS_LDS"] / steps,
                              "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"]} for f, c in per.items()},
           "valu_wave_insts_per_step": insts, "extend_stage_ms": stage_ms,
           "valu_frac": insts / (stage_ms * 1e-3) / (measured_peak or peak4),
           "valu_peak_wave_insts_per_s": measured_peak or peak4, "valu_peak_source": measured_src or "nominal: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction",
           "valu_frac_nominal_4_cycles": insts / (stage_ms * 1e-3) / peak4,
           "valu_busy_of_wave_cycles": tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"],
           "wait_any_of_wave_cycles": tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
           "lds_insts_per_step": tot["SQ_INSTS_LDS"] / steps,
           "lds_conflict_frac": tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"] if tot.get("SQ_LDS_IDX_ACTIVE") else None,
           "lane_slots_per_cell": insts * 64.0 / (bench["extend_kernel"].get("cells_per_step") or bench["extend_kernel"]["cells_per_launch"]),
           "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                     "SQ_WAIT_INST_LDS (profiles/%s_pmc_sq1.md), sums over the k_ext_lanes / k_ext_wave instantiations and both steps of the pass; the "
                     "launches of a stage overlap on side streams, so the issue fraction is taken over the stage's wall time of the bench run "
                     "(%.1f ms), not over the summed kernel durations" % (pre, stage_ms)}
    json.dump(ext, open(os.path.join(dst, "%s_ext_pmc_sq.json" % pre), "w"), indent=1)
    print(json.dumps(out, indent=1)); print(json.dumps(ext, indent=1))


if __name__ == "__main__":
    main()
