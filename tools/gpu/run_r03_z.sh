#!/bin/bash
# Round 3, call Z: issue priority (s_setprio) for the wavefronts of the long query classes; the shortest class behind the second shortest
TAG=${1:-r03z}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 100 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "golden_all_stages" 2>&1 | tail -2
timeout 400 python tools/gpu/sweep.py $O --steps 5 --only "distinct hardware queues,issue priority" --budget-s 200 2>&1 | grep "\[sweep\]" | tail -14
echo "finished at $(( $(date +%s) - T0 ))s"
