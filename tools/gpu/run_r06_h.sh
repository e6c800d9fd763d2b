#!/bin/bash
# Round 6, eighth call: the bench's main line with its parity gates and the FASTQ -> SAM leg (100 chunks) on the current code; no side workloads, no CLI leg.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_h.sh r06h 850'
TAG=${1:-r06h}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 --no-side-workloads --no-binding --full-json $O/bench_full.json > $O/bench_stdout.txt 2> $O/bench.err; at bench $?
grep "^\[bench\]" $O/bench.err | tail -8 | cut -c1-330
python3 -c "
import json; d=json.load(open('$O/bench_full.json')); e=d['end_to_end']
print({k: e.get(k) for k in ('value','chunks','steady_state','stage_ms_per_chunk','host_cpu_s_per_chunk','host_cpu_bound_ms_per_chunk','ms_per_chunk','frac_of_hot_path')})
"
echo "finished at $(( $(date +%s) - T0 ))s"
