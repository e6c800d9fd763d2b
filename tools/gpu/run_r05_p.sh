#!/bin/bash
# Round 5, call P: the seed filter's SW row in registers (k_seed_sw_reg), island chaining given up once equal keys are seen; variants on config 5
TAG=${1:-r05p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -x -m gpu -k "long_reads" > $O/tests.log 2>&1; at tests $?; tail -3 $O/tests.log
B="python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
timeout 300 $B > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont $?
python -c "
import json
d=json.loads(open('$O/bench_ont2d.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['stage_ms_per_step']); print(d['chain_kernel']['serial_reads'])"
for kv in BM2_SEEDSW_REG=3 BM2_SEEDSW_REG=0 BM2_CHAIN_ISL_WAVES_PER_CU=8 BM2_CHAIN_ISL_WAVES_PER_CU=12; do env $kv timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['stage_ms_per_step'], d['chain_kernel']['serial_reads']['slowest_ms'])"; done; at variants $?
