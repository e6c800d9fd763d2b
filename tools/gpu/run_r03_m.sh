#!/bin/bash
# Round 3, call M: the bench exactly as the driver runs it (--steps 20 --warmup 5): four distinct resident chunks, wide gate, CPU baseline, end-to-end, binding.
TAG=${1:-r03m}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? at $(( $(date +%s) - T0 ))s"
grep -E "\[bench\]" $O/bench_full.err | tail -14
python - <<P
import json
d = json.load(open("$O/bench_full.json"))
print("value %.2f M reads/s, %.1f ms/step, resident chunks %s" % (d["value"] / 1e6, d["ms_per_step"], d["config"].get("resident_chunks")), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()})
print("roofline frac %.3f counter %.3f; extend %s" % (d["roofline"]["frac"], d["roofline"]["frac_counter"] or 0, {k: d["extend_kernel"].get(k) for k in ("gcups", "valu_frac", "lds_conflict_frac", "pmc_source")}))
e = d.get("end_to_end") or {}
print("end_to_end %.2f M reads/s (%.2f)" % (e.get("value", 0) / 1e6, e.get("frac_of_hot_path", 0)), e.get("chunk_check"))
print("binding", {k: v for k, v in (d.get("binding") or {}).items() if k != "scope"})
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
P
