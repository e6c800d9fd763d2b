#!/bin/bash
# Round 4, call K: config 5 with the island kernel's reads longest first and look-up-free stray seeds in the serial reads; the whole GPU suite.
TAG=${1:-r04k}; LIMIT=${2:-700}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python bench.py --workload ont2d --no-cpu-baseline --parity-reads 200 --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err
python - <<P
import json
d = json.load(open("$O/bench_ont2d.json"))
print("ont2d: %.0f reads/s, stages %s, parity %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}))
print(json.dumps(d.get("chain_kernel")))
P
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
