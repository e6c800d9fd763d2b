#!/bin/bash
# Round 5, call U: bench.py with configs 5 and 2 run AHEAD of the main line (before the process opens the device); a reduced line (no e2e / binding / parity legs)
TAG=${1:-r05u}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 320 python bench.py --steps 5 --warmup 2 --no-e2e --no-binding --no-parity > $O/bench.json 2> $O/bench.err; at bench $?
grep "^\[bench\]" $O/bench.err | tail -8 | cut -c1-420
