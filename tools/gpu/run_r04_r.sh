#!/bin/bash
# Round 4, call R: new defaults (two parts per chunk, classes from 129 bases on the wavefront kernel): pipeline + sharded + end-to-end GPU
# tests, the whole bench as the driver runs it, then one-knob variants of the hot path (one part, staggered parts, 24 / 32 hardware queues).
TAG=${1:-r04r}; LIMIT=${2:-800}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python -m pytest tests/test_pipeline_gpu.py tests/test_sharded.py tests/test_end_to_end_gpu.py -m gpu -x -q > $O/pytest_some.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_some.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<P
import json
try:
    d = json.load(open("$O/bench.json"))
    print("value %.2f M reads/s, %.1f ms/step, parts %s, stages %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("parts_per_chunk"), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
    print("roofline frac %.3f launches %s avg ms %.2f; stage frac %.3f" % (d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["seeding_stage"]["frac"]))
    e = d.get("end_to_end") or {}
    print("e2e %.2f M (%.2f) stages %s" % ((e.get("value") or 0) / 1e6, e.get("frac_of_hot_path") or 0, e.get("stage_ms_per_chunk")))
    print("parity", json.dumps(d.get("parity"))[:200])
    for k in ("config5", "config2"):
        c = d.get(k) or {}
        print(k, c.get("value"), c.get("stage_ms_per_step"))
    print("binding", json.dumps(d.get("binding"))[:400])
except Exception as e:
    print("no bench line:", e)
P
show() { python - <<P
import json
d = json.load(open("$1"))
print("$2: value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
P
}
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding --no-parity"
BM2_N_SUB=1 timeout 200 python bench.py $Q > $O/b_sub1.json 2> $O/b_sub1.err; show $O/b_sub1.json one_part
BM2_SUB_STAGGER=1 timeout 200 python bench.py $Q > $O/b_stagger.json 2> $O/b_stagger.err; show $O/b_stagger.json staggered
GPU_MAX_HW_QUEUES=24 timeout 200 python bench.py $Q > $O/b_q24.json 2> $O/b_q24.err; show $O/b_q24.json queues24
GPU_MAX_HW_QUEUES=32 timeout 200 python bench.py $Q > $O/b_q32.json 2> $O/b_q32.err; show $O/b_q32.json queues32
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $Q > $O/b_q8.json 2> $O/b_q8.err; show $O/b_q8.json queues8
