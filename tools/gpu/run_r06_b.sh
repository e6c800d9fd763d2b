#!/bin/bash
# Round 6, second call: A/B of the extension stage's new launch-policy knobs on the resident 3100 Mbp chunk (tools/gpu/sweep.py: one process, regs checksummed) --
# wave priority of the long classes (BM2_EXT_PRIO_QMIN / BM2_EXT_PRIO / BM2_EXT_WAVE_PRIO), rows in registers (BM2_EXT_REG_QMIN) -- then the same register
# rows on seam S1 (config 2's workload: one phase, every pair against the reference's kernels).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_b.sh r06b 850'
TAG=${1:-r06b}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 500 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "extension: rows in registers" > $O/sweep.out 2> $O/sweep.err; at sweep $?
grep "\[sweep\]" $O/sweep.err | tail -40 | cut -c1-330
for q in 0 96 48; do
  if [ $(left) -gt 100 ]; then
    BM2_EXT_REG_QMIN=$q timeout 150 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 --full-json $O/bench_bsw_reg$q.json > $O/bsw_reg$q.line 2> $O/bsw_reg$q.err; at bsw_reg$q $?
    python -c "import json; d=json.load(open('$O/bench_bsw_reg$q.json')); print('REG_QMIN=$q', d['extend_kernel'], d['parity']['pairs_equal'])"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
