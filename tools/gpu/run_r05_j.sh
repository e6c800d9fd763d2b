#!/bin/bash
# Round 5, call J: config 5's SQ counter pass at the chunk size of the default line (20 000 reads) -> profiles/r05_ont2d_pmc_sq.md, r05_ont2d_ext_pmc_sq.json
TAG=${1:-r05j}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
timeout 400 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_ont -o s -- python $R/bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 2 > $O/bench_ont2d_pmc.json 2> $O/pmc_ont.err; at ont_sq $?
python $R/tools/rocpd_summary.py $(find /tmp/p_ont -name "*.db" | head -1) $O/pmc_sq1_ont2d.md > /dev/null 2>> $O/pmc_ont.err
grep -n "k_walk<1>\|k_chain_islands\|k_seed_sw\|k_ext_wave" $O/pmc_sq1_ont2d.md | head -12
