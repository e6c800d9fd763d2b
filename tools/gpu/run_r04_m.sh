#!/bin/bash
# Round 4, call M: target bases four rows at a time + query staging by dwords in the lane kernel: S1 A/B, hot path, parity gate on a prefix.
TAG=${1:-r04m}; LIMIT=${2:-500}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 100 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw.json 2> $O/bench_bsw.err
python -c "
import json; d=json.load(open('$O/bench_bsw.json')); print('bsw', d['extend_kernel'], d['parity']['pairs_equal'])"
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --parity-reads 51200 > $O/bench.json 2> $O/bench.err
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
print("parity", {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")})
P
timeout 200 python -m pytest tests/test_pipeline_gpu.py tests/test_bsw_gpu.py -m gpu -x -q 2>&1 | tail -2
