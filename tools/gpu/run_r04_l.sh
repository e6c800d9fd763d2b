#!/bin/bash
# Round 4, call L: the end-to-end leg's worker split (tail workers x threads) after the hot path got shorter.
TAG=${1:-r04l}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
E2E="python bench.py --no-cpu-baseline --no-parity --no-binding --no-side-workloads --steps 6 --warmup 4"
for V in "3 5" "2 8" "2 7" "4 4" "3 6" "2 8"; do
  set -- $V
  if [ $(left) -gt 90 ]; then
    BM2_E2E_TAILS=$1 BM2_E2E_TAIL_THREADS=$2 timeout 200 $E2E > $O/e2e_$1x$2.json 2> $O/e2e_$1x$2.err
    python - <<P
import json
try:
    d = json.load(open("$O/e2e_$1x$2.json"))
    e = d["end_to_end"]
    print("tails $1 x $2 threads: end to end %.2f M reads/s (%.2f of %.2f M), stages %s" % (e["value"] / 1e6, e["frac_of_hot_path"], d["value"] / 1e6, {k: round(v, 1) for k, v in e["stage_ms_per_chunk"].items()}))
except Exception as ex:
    print("no line:", ex)
P
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
