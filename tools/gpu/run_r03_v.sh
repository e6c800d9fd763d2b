#!/bin/bash
# Round 3, call V: launch-policy sweep of the lane-kernel variants (rows in registers up to a class bound, scores by byte permute) and of k_bwd's occupancy
TAG=${1:-r03v}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 200 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "lane_kernel_variants or golden_all_stages" 2>&1 | tail -2
echo "pytest done at $(( $(date +%s) - T0 ))s"
timeout 400 python tools/gpu/sweep.py $O --steps 4 --only "rows in registers,byte permute,k_bwd,walk blocks" --budget-s 200 2>&1 | grep "\[sweep\]" | tail -24
for q in 1 2; do timeout 60 tools/ubench/randline 6200 8 1500 2 $q; timeout 60 tools/ubench/randline 6200 4 1500 2 $q; done
echo "finished at $(( $(date +%s) - T0 ))s"
