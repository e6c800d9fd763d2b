#!/bin/bash
# Round 5, call A: (1) every off-by-default knob through the GPU parity test (BM2_CHAIN_COOP_FLT, BM2_EXT_WAVE_BUDGET, BM2_CHAIN_CLOCK had never met
# a GPU; BM2_BWD_EXPORT_AGE / BM2_BWD_HEAVY_AFTER / BM2_P3_BPC are new), (2) the same knobs timed on the 3100 Mbp bench chunk in one process,
# (3) config 5 with and without the cooperative chain filter.
#   gpurun --timeout 1100 -- 'bash tools/gpu/run_r05_a.sh r05a'
TAG=${1:-r05a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "off_by_default or repeat_rich" > $O/pytest_knobs.log 2>&1; at pytest $?
tail -5 $O/pytest_knobs.log
timeout 420 python tools/gpu/sweep.py $O --steps 4 --budget-s 300 --only "seeding:,chain clock,kept-chain walk,seed budget" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -40
Q="--workload ont2d --no-cpu-baseline --steps 2 --warmup 1"
timeout 200 python bench.py $Q > $O/ont_default.json 2> $O/ont_default.err; at ont_default $?
BM2_CHAIN_COOP_FLT=1 timeout 200 python bench.py $Q > $O/ont_coop.json 2> $O/ont_coop.err; at ont_coop $?
python - <<P
import json
for n in ("ont_default", "ont_coop"):
    try:
        d = json.load(open("$O/%s.json" % n))
        print(n, "value %.0f reads/s, %.1f ms/step" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, json.dumps(d.get("parity"))[:200])
    except Exception as e:
        print(n, "no line:", e)
P
