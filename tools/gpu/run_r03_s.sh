#!/bin/bash
# Round 3, call S: kernel trace of the config-5 workload (10 000 ONT-like reads per step)
TAG=${1:-r03s}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --workload ont2d --steps 2 --warmup 1 --no-parity --no-cpu-baseline > $O/bench_ont2d.json 2> $O/kt.err
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace_ont2d.md > /dev/null 2>> $O/kt.err
head -40 $O/kernel_trace_ont2d.md
