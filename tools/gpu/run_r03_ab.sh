#!/bin/bash
# Round 3, call AB: the launch-policy knobs of chaining / SA lookup / rounds once more on the round's last code
TAG=${1:-r03ab}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 330 python tools/gpu/sweep.py $O --steps 5 --only "chain heavy threshold,chain waves,SA lookup,purge threshold,extension rounds" --budget-s 170 2>&1 | grep "\[sweep\]" | tail -20
echo "finished at $(( $(date +%s) - T0 ))s"
