#!/bin/bash
# Round 5, call R: k_chain_serial's LDS against the heavy tiers' (both want a CU's whole LDS), island wavefronts per CU
TAG=${1:-r05r}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
B="python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
for kv in "BM2_CHAIN_ISL_WAVES_PER_CU=6" "BM2_CHAIN_ISL_WAVES_PER_CU=4" "BM2_CHAIN_ISL_WAVES_PER_CU=6 BM2_CHAIN_SERIAL_LNODES=250" "BM2_CHAIN_ISL_WAVES_PER_CU=6 BM2_CHAIN_SERIAL_LNODES=120" "BM2_CHAIN_ISL_WAVES_PER_CU=6 BM2_CHAIN_TIER_MAX=64" "BM2_CHAIN_ISL_WAVES_PER_CU=6 BM2_CHAIN_TIER_MAX=256" "BM2_CHAIN_ISL_WAVES_PER_CU=8 BM2_CHAIN_SERIAL_LNODES=250" "BM2_CHAIN_ISL_WAVES_PER_CU=6 BM2_CHAIN_TIER_MAX=256 BM2_CHAIN_SERIAL_LNODES=250"; do env $kv timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', round(d['value']), {k: round(v) for k, v in d['stage_ms_per_step'].items()}, {k: (round(v) if isinstance(v, float) else None) for k, v in d['chain_kernel']['serial_reads'].items()})"; done 2>&1 | tee $O/variants.txt; at variants $?
