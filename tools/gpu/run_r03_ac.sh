#!/bin/bash
# Round 3, call AC (the last): the tree after the measured-negative variants were taken out again: pipeline tests and the gate on the GPU
TAG=${1:-r03ac}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 120 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "pytest done at $(( $(date +%s) - T0 ))s"
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-binding --parity-reads 20480 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? at $(( $(date +%s) - T0 ))s"
grep "parity" $O/bench.err | tail -3
python -c "import json; d=json.load(open('$O/bench.json')); print('pe150: %.2f M reads/s, %.1f ms' % (d['value']/1e6, d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
