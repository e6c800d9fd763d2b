#!/bin/bash
# Round 4, call J: where the island kernel's time goes on config 5 (serial reads), seed-rich 150 bp reads on the island kernel (sweep).
TAG=${1:-r04j}; LIMIT=${2:-500}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 200 python bench.py --workload ont2d --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err
python - <<P
import json
d = json.load(open("$O/bench_ont2d.json"))
print("ont2d: %.0f reads/s, stages %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
print(json.dumps(d.get("chain_kernel")))
P


