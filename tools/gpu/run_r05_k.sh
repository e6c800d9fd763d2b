#!/bin/bash
# Round 5, call K: k_chain_heavy allocated for 2 / 3 / 4 wavefronts per SIMD (BM2_CHAIN_HEAVY_WPE); the knob parity test first.
TAG=${1:-r05k}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 400 python -m pytest tests/test_pipeline_gpu.py -q -x -m gpu -k "off_by_default_knobs" > $O/knobs.log 2>&1; at knobs $?; tail -3 $O/knobs.log
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "wavefronts per SIMD the heavy" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -10 | cut -c1-400
