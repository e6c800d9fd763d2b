#!/bin/bash
# Round 3, call AD (what was left of the budget: ~90 s): the rebuilt index builder on the GPU box -- its time at 3100 Mbp -- and three steps on its index
TAG=${1:-r03ad}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp BM2_VERBOSE=1
timeout 76 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-e2e --no-binding --resident-chunks 1 > $O/bench.json 2> $O/bench.err
echo "rc=$?"; grep -E "bm2_index_build|\[bench\]" $O/bench.err | tail -16; head -c 300 $O/bench.json
