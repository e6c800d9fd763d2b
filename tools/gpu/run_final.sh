#!/bin/bash
# The round's second (last) short gpurun call: sweep of the knobs added after the first call (k_bwd LDS depth / occupancy, staged
# heavy chaining, ...), then the config-5 bench line (`--workload ont2d`), then a kernel trace with the sweep's winners exported.
#   gpurun --timeout 350 -- 'bash tools/gpu/run_final.sh r02b 335'
TAG=${1:-r02b}; LIMIT=${2:-335}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R
export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 260 python tools/gpu/sweep.py $O --steps 4 --budget-s ${SWEEP_S:-45} > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s" >> $O/sweep.err
grep "\[sweep\]" $O/sweep.err | tail -40
[ -f $O/best_env.sh ] && . $O/best_env.sh
env | grep "^BM2_" > $O/env_used.txt
if [ $(left) -gt 70 ]; then
  T=$(( $(left) - 45 )); [ $T -gt 150 ] && T=150
  timeout $T python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 48 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$? at $(( $(date +%s) - T0 ))s" | tee -a $O/bench_ont2d.err
  tail -5 $O/bench_ont2d.err; head -c 700 $O/bench_ont2d.json; echo
fi
cd /tmp
if [ $(left) -gt 35 ]; then
  timeout $(( $(left) - 5 )) rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --steps 4 --warmup 1 > $O/bench.json 2> $O/kt.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  echo "kt done at $(( $(date +%s) - T0 ))s"; head -c 400 $O/bench.json; echo; head -12 $O/kernel_trace.md
fi
echo "finished at $(( $(date +%s) - T0 ))s"
