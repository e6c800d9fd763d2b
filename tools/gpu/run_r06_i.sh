#!/bin/bash
# Round 6, ninth call: the FASTQ -> SAM leg on the bench's own 3100 Mbp genome, ten 1 M-read chunks: how many tail workers x threads per worker keep the
# 16-CPU host busiest now that the tail's device batches are shorter (the leg is bound by the tail workers: 3 x 262 ms per chunk).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_i.sh r06i 850'
TAG=${1:-r06i}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=200 PROBE_SPLITS="3x5 4x4 4x5 5x3 5x4 6x3 4x3 3x6" timeout 800 python tools/gpu/tail_probe.py $O 3100 10 500000 > $O/tail_probe.out 2> $O/tail_probe.err; at tail_probe $?
grep "\[probe\]" $O/tail_probe.err | cut -c1-260
echo "finished at $(( $(date +%s) - T0 ))s"
