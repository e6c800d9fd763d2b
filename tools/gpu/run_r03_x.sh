#!/bin/bash
# Round 3, call X: the wave-kernel class bound again now that a lane cell costs 22 instructions; per-dispatch timeline of two steps (what the extension phases look like)
TAG=${1:-r03x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 400 python tools/gpu/sweep.py $O --steps 4 --only "extension wave classes,extension dispatch,extension prefetch,purge threshold,byte permute" --budget-s 200 2>&1 | grep "\[sweep\]" | tail -16
echo "sweep done at $(( $(date +%s) - T0 ))s"
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/p_tl -o tl -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-binding --steps 2 --warmup 1 > $O/bench_tl.json 2> $O/tl.err; echo "trace rc=$? at $(( $(date +%s) - T0 ))s"
python $R/tools/rocpd_timeline.py $(find /tmp/p_tl -name "*.db" | head -1) $O/timeline.tsv 2>> $O/tl.err
wc -l $O/timeline.tsv
echo "finished at $(( $(date +%s) - T0 ))s"
