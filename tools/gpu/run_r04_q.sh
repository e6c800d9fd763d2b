#!/bin/bash
# Round 4, call Q: launch-policy sweep after the join-order fix (which classes go to the wavefront kernel, rounds, dispatch order, chain tiers).
TAG=${1:-r04q}; LIMIT=${2:-400}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 360 python tools/gpu/sweep.py $O --steps 4 --budget-s 240 --only "extension wave,extension rounds,extension dispatch,extension launches,chain waves,sub-batches,chain heavy" > $O/sweep.log 2>&1
echo "sweep rc=$?"; tail -40 $O/sweep.log
