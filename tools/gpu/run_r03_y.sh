#!/bin/bash
# Round 3, call Y: the extension side's eight launches on eight hardware queues (BM2_EXT_QUEUE_MAP) against the old stream assignment; timeline of a step
TAG=${1:-r03y}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 100 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "golden_all_stages or fresh_inputs" 2>&1 | tail -2
timeout 400 python tools/gpu/sweep.py $O --steps 5 --only "distinct hardware queues,extension wave classes" --budget-s 200 2>&1 | grep "\[sweep\]" | tail -12
echo "sweep done at $(( $(date +%s) - T0 ))s"
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/p_tl -o tl -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-binding --steps 2 --warmup 1 > $O/bench_tl.json 2> $O/tl.err; echo "trace rc=$? at $(( $(date +%s) - T0 ))s"
python $R/tools/rocpd_timeline.py $(find /tmp/p_tl -name "*.db" | head -1) $O/timeline.tsv 2>> $O/tl.err
echo "finished at $(( $(date +%s) - T0 ))s"
