#!/bin/bash
# Round 3, call K: a chunk as staggered parts (seeding of part i + 1 under the extension of part i): parity test, then the hot path at 1 .. 4 parts.
TAG=${1:-r03k}; LIMIT=${2:-500}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 200 python -m pytest tests/test_pipeline_gpu.py tests/test_bsw_reference.py -x -q -m gpu 2>&1 | tail -3
for v in "1 1" "2 1" "3 1" "4 1" "2 0" "6 1"; do
  set -- $v
  [ $(left) -lt 60 ] && break
  BM2_N_SUB=$1 BM2_SUB_STAGGER=$2 timeout 200 python bench.py --steps 8 --warmup 2 --no-parity --no-cpu-baseline --no-e2e > $O/bench_n$1_s$2.json 2> $O/bench_n$1_s$2.err
  python -c "import json; d=json.load(open('$O/bench_n$1_s$2.json')); print('N_SUB=$1 stagger=$2: value %.2f M, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'bwd', round(d['roofline']['avg_launch_ms'],2), 'frac', round(d['roofline']['frac'],3))"
done
echo "finished at $(( $(date +%s) - T0 ))s"
