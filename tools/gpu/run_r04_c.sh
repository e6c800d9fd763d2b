#!/bin/bash
# Round 4, call C: where pass 3 of the seeding runs (sweep), the end-to-end leg with page-locked parser output and lazily created side
# streams under 8 / 16 / 24 hardware queues, a counter pass of the lane-per-seed kernel alone (every class on it).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r04_c.sh r04c 880'
TAG=${1:-r04c}; LIMIT=${2:-880}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 60 --only "pass 3" > $O/sweep.out 2> $O/sweep.err; at sweep $?
grep "\[sweep\]" $O/sweep.err | tail -8
E2E="python bench.py --no-cpu-baseline --no-parity --no-binding --no-side-workloads --steps 6 --warmup 4"
for Q in 8 16 24; do
  if [ $(left) -gt 120 ]; then
    GPU_MAX_HW_QUEUES=$Q timeout 200 $E2E > $O/bench_q$Q.json 2> $O/bench_q$Q.err; at "e2e queues=$Q" $?
    python - <<P
import json
try:
    d = json.load(open("$O/bench_q$Q.json"))
    e = d["end_to_end"]
    print("queues $Q: hot path %.2f M reads/s (%.1f ms), end to end %.2f M reads/s (%.2f), stages %s, lane use %s" % (d["value"] / 1e6, d["ms_per_step"], e["value"] / 1e6, e["frac_of_hot_path"],
          {k: round(v, 1) for k, v in e["stage_ms_per_chunk"].items()}, d["extend_kernel"].get("lane_use_of_the_column_loop")))
except Exception as ex:
    print("no line:", ex)
P
  fi
done
cd /tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $(left) -gt 90 ]; then
  BM2_EXT_WAVE_QMIN=161 timeout 150 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --steps 1 --warmup 1 > $O/bench_pmc_lanes_only.json 2> $O/pmc_sq1.err; at pmc $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1_lanes_only.md > /dev/null 2>> $O/pmc_sq1.err
  grep "k_ext" $O/pmc_sq1_lanes_only.md | head -20
fi
echo "finished at $(( $(date +%s) - T0 ))s"
