#!/bin/bash
# Round 4, call D: the grouped column loop of the lane kernel and the sliding register window of the long-query kernel on the hardware:
# GPU tests, sweep (group4 on / off, pass 3), hot path under 8 / 16 hardware queues (A/B/A/B), config 5, config 2.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r04_d.sh r04d 880'
TAG=${1:-r04d}; LIMIT=${2:-880}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 70 --only "pass 3,groups of four" > $O/sweep.out 2> $O/sweep.err; at sweep $?
grep "\[sweep\]" $O/sweep.err | tail -8
timeout 400 python -m pytest tests/test_bsw_gpu.py tests/test_bsw_reference.py tests/test_pipeline_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
HOT="python bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --steps 12 --warmup 4"
for Q in 8 16 8 16; do
  if [ $(left) -gt 200 ]; then
    GPU_MAX_HW_QUEUES=$Q timeout 120 $HOT > $O/hot_q$Q.json 2> $O/hot_q$Q.err
    python - <<P
import json
try:
    d = json.load(open("$O/hot_q$Q.json"))
    print("queues $Q: %.2f M reads/s, %.2f ms/step, stages %s, lane use %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, d["extend_kernel"].get("lane_use_of_the_column_loop")))
except Exception as ex:
    print("no line:", ex)
P
  fi
done
if [ $(left) -gt 150 ]; then
  timeout 140 python bench.py --workload ont2d --no-cpu-baseline --parity-reads 200 --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont2d $?
  python - <<P
import json
try:
    d = json.load(open("$O/bench_ont2d.json"))
    print("ont2d: %.0f reads/s, stages %s, parity %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}))
except Exception as ex:
    print("no line:", ex)
P
fi
if [ $(left) -gt 60 ]; then
  timeout 50 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw.json 2> $O/bench_bsw.err; at bsw $?; python -c "
import json; d=json.load(open('$O/bench_bsw.json')); print('bsw', d['extend_kernel'], d['parity']['pairs_equal'])"
fi
echo "finished at $(( $(date +%s) - T0 ))s"
