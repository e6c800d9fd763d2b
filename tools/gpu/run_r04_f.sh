#!/bin/bash
# Round 4, call F: queue priority for the long classes' extension launches (A/B over processes), PMC pass of the extension stage.
TAG=${1:-r04f}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
HOT="python bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --steps 12 --warmup 4"
for M in 0 112 113 240 0 112; do
  if [ $(left) -gt 100 ]; then
    BM2_SIDE_PRIO_MASK=$M timeout 120 $HOT > $O/hot_m$M.json 2> $O/hot_m$M.err
    python - <<P
import json
try:
    d = json.load(open("$O/hot_m$M.json"))
    print("prio mask $M: %.2f M reads/s, %.2f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
except Exception as ex:
    print("no line:", ex)
P
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
