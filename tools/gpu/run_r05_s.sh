#!/bin/bash
# Round 5, call S: what the wavefront-per-read tiers of k_chain_heavy do with a long-read chunk's reads of 100..1000 seeds (295 ms of the chain stage): their own clock
TAG=${1:-r05s}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
B="python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
for kv in "BM2_X=0" "BM2_CHAIN_CLOCK=1"; do env $kv timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', round(d['value']), {k: round(v) for k, v in d['stage_ms_per_step'].items()}, d['chain_kernel'].get('k_chain_heavy_clock'))"; done 2>&1 | tee $O/variants.txt; at variants $?
