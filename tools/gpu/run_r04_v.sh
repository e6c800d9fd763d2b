#!/bin/bash
# Round 4, call V: the last device-code change of the round (idle wavefronts of the extension kernels leave without their counter atomics): pipeline tests + gate.
TAG=${1:-r04v}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 200 python -m pytest tests/test_pipeline_gpu.py tests/test_bsw_gpu.py -m gpu -x -q > $O/pytest_some.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_some.log
timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding --parity-reads 51200 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
print("parity", {k: (d.get("parity") or {}).get(k) for k in ("regs_equal", "fin_equal", "sam_equal")})
P
