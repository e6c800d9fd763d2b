#!/bin/bash
# Round 5, call I: pass 3 started with the stage on a thin grid (one or two workgroups per CU: it then runs beside k_walk<1> AND k_bwd of pass 1).
TAG=${1:-r05i}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "pass 3 workgroups" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -10 | cut -c1-400
