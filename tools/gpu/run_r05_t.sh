#!/bin/bash
# Round 5, call T: k_chain_serial's helper lanes touch the leaves of the next 64 seeds (BM2_CHAIN_SERIAL_PREFETCH)
TAG=${1:-r05t}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -x -m gpu -k "long_reads" > $O/tests.log 2>&1; at tests $?; tail -3 $O/tests.log
B="python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
for kv in "BM2_X=0" "BM2_CHAIN_SERIAL_PREFETCH=0" "BM2_X=1"; do env $kv timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', round(d['value']), {k: round(v) for k, v in d['stage_ms_per_step'].items()}, {k: (round(v) if isinstance(v, float) else None) for k, v in d['chain_kernel']['serial_reads'].items()})"; done 2>&1 | tee $O/variants.txt; at variants $?
