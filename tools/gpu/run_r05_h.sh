#!/bin/bash
# Round 5, call H: pass 3 of the seeding with its first bases from the k-mer table (BM2_KTAB_K=12): parity of the knob settings, then timed against
# the walks without the table in one process (BM2_KTAB_USE=0), with pass 3 beside k_walk<1> / k_bwd of pass 1 / pass 2.
TAG=${1:-r05h}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 400 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "off_by_default" > $O/pytest_knobs.log 2>&1; at pytest $?
tail -3 $O/pytest_knobs.log
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "k-mer table" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -10 | cut -c1-520
