#!/bin/bash
# Round 4, call I: B-tree nodes visited through registers: the pipeline's GPU tests, config 5, the 150 bp hot path with and without.
TAG=${1:-r04i}; LIMIT=${2:-700}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python bench.py --workload ont2d --no-cpu-baseline --parity-reads 200 --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err
python - <<P
import json
d = json.load(open("$O/bench_ont2d.json"))
print("ont2d: %.0f reads/s, stages %s, parity %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}))
print(json.dumps(d.get("chain_kernel")))
P
HOT="python bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --steps 12 --warmup 4"
for RN in 1 0 1 0; do
  if [ $(left) -gt 100 ]; then
    BM2_CHAIN_REGNODES=$RN timeout 120 $HOT > $O/hot_rn$RN.json 2> $O/hot_rn$RN.err
    python - <<P
import json
try:
    d = json.load(open("$O/hot_rn$RN.json"))
    print("reg nodes $RN: %.2f M reads/s, %.2f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
except Exception as ex:
    print("no line:", ex)
P
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
