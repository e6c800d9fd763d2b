#!/bin/bash
# Round 4, call H: the island kernel's own phase clock on config 5, its parity gate, the long-read tests.
TAG=${1:-r04h}; LIMIT=${2:-400}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python bench.py --workload ont2d --no-cpu-baseline --parity-reads 200 --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err
python - <<P
import json
d = json.load(open("$O/bench_ont2d.json"))
print("ont2d: %.0f reads/s, stages %s, parity %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}))
print(json.dumps(d.get("chain_kernel"), indent=1))
P
timeout 200 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -s -k "long_reads" 2>&1 | grep -E "island path|passed|failed"
