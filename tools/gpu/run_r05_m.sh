#!/bin/bash
# Round 5, call M: config 5 at 20 000 reads per step with k_chain_serial: per-kernel times (kernel trace), then the tiers of k_chain_heavy cut at 256 / 64 seeds
TAG=${1:-r05m}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
B="python $R/bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_ont -o s -- $B > $O/bench_ont2d_kt.json 2> $O/kt.err; at kt $?
python $R/tools/rocpd_summary.py $(find /tmp/p_ont -name "*.db" | head -1) $O/kernel_trace_ont2d.md > /dev/null 2>> $O/kt.err
head -16 $O/kernel_trace_ont2d.md | cut -c1-150
for tm in 256 64; do BM2_CHAIN_TIER_MAX=$tm timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tier_max $tm', d['value'], d['stage_ms_per_step'], d['chain_kernel']['serial_reads'])"; done; at tiers $?
