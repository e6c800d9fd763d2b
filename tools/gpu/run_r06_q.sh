#!/bin/bash
# Round 6: config 5 (20 000 ONT-like reads per step) with l_rep by the whole wavefront in the long-read chaining kernels (notes/patches/lrep_coop.patch of
# round 5, applied): the workload's own bench line with its gate, twice; then its long-read GPU tests.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_q.sh r06q 1150'
TAG=${1:-r06q}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
for rep in 1 2; do
  timeout 500 python bench.py --workload ont2d --reads 20000 --steps 3 --warmup 1 --no-cpu-baseline --parity-reads 1024 --parity-regs-reads 100 --full-json $O/bench_ont2d_$rep.json > $O/ont2d_$rep.line 2> $O/ont2d_$rep.err; at ont2d_$rep $?
  grep "^\[bench\] hot path\|parity gate" $O/ont2d_$rep.err | cut -c1-260
done
timeout 400 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "long_reads" > $O/pytest_long.log 2>&1; echo "pytest rc=$?" >> $O/pytest_long.log; tail -3 $O/pytest_long.log
echo "finished at $(( $(date +%s) - T0 ))s"
