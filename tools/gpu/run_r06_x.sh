#!/bin/bash
# Round 6: two parts per chunk (BM2_N_SUB=2) once more on this round's tree -- round 4 measured 63.7 instead of 69.7 ms on a lone context and 74 instead of 69 ms
# in the bench's rotation of resident chunks: the lone context by tools/gpu/sweep.py, the rotation by bench.py, alternating processes.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_x.sh r06x 850'
TAG=${1:-r06x}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
timeout 400 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "sub-batches" > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s"
grep "\[sweep\]" $O/sweep.err | tail -6 | cut -c1-420
for cfg in ${CFGS:-1 2 1 2 3}; do
  if [ $(left) -gt 120 ]; then
    env BM2_N_SUB=$cfg timeout 200 python bench.py --steps 12 --warmup 4 --no-parity --no-cpu-baseline --no-side-workloads --no-binding --no-e2e --full-json $O/bench_sub$cfg.json > /dev/null 2> $O/bench_sub$cfg.err
    echo "== N_SUB=$cfg rc=$? at $(( $(date +%s) - T0 ))s"
    python3 -c "
import json; d=json.load(open('$O/bench_sub$cfg.json')); print('  ', d.get('ms_per_step'), d.get('value'), d.get('stage_ms_per_step'))"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
