#!/bin/bash
# Round 6, tenth call: the FASTQ -> SAM leg over 100 chunks (as the bench runs it) with 3 x 5 (default), 3 x 6, 4 x 5 and 4 x 4 tail workers x threads.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_j.sh r06j 1150'
TAG=${1:-r06j}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
for cfg in ${CFGS:-3x5 3x6 4x5 4x4 3x5}; do       # CFGS="3x6 3x7 ...": tail workers x threads per worker
  set -- ${cfg%x*} ${cfg#*x}
  if [ $(left) -gt 150 ]; then
    BM2_E2E_TAILS=$1 BM2_E2E_TAIL_THREADS=$2 BM2_E2E_DEVS=${DEVS:-2} BM2_E2E_PARSE_THREADS=${PARSE:-6} timeout 300 python bench.py --steps 8 --warmup 4 --no-parity --no-cpu-baseline --no-side-workloads --no-binding --full-json $O/bench_$1x$2.json > /dev/null 2> $O/bench_$1x$2.err; at "tails $1 x $2" $?
    grep "end_to_end (FASTQ" $O/bench_$1x$2.err | cut -c1-260
    python3 -c "
import json; e=json.load(open('$O/bench_$1x$2.json'))['end_to_end']
print('  ', {k: e.get(k) for k in ('stage_ms_per_chunk','host_cpu_s_per_chunk')})"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
