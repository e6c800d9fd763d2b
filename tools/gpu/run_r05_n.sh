#!/bin/bash
# Round 5, call N: k_chain_serial BESIDE the island kernel (it takes a read as soon as the island kernel lists it)
TAG=${1:-r05n}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -x -m gpu -k "long_reads or (off_by_default and SERIAL)" > $O/tests.log 2>&1; at tests $?; tail -5 $O/tests.log
B="python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
timeout 300 $B > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont $?
grep "hot path" $O/bench_ont2d.err | tail -2 | cut -c1-300
python -c "
import json
d=json.loads(open('$O/bench_ont2d.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stage_ms_per_step']); print(d['chain_kernel']['serial_reads'])"
for kv in BM2_CHAIN_SERIAL_BESIDE=0 BM2_CHAIN_TIER_MAX=512 BM2_CHAIN_TIER_MAX=256; do env $kv timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['stage_ms_per_step'])"; done; at variants $?
