#!/bin/bash
# Round 5, call L: the reads with equal chain keys in a launch of their own (k_chain_serial), the island kernel without the serial code (128 registers instead of 264)
TAG=${1:-r05l}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 500 python -m pytest tests/test_pipeline_gpu.py -q -x -m gpu -k "long_reads or off_by_default_knobs" > $O/tests.log 2>&1; at tests $?; tail -5 $O/tests.log
timeout 300 python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont $?
grep "\[bench\]" $O/bench_ont2d.err | tail -5 | cut -c1-300
python - <<PY
import json
d=json.loads(open("$O/bench_ont2d.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms_per_step"])
ck=d.get("chain_kernel",{}); print({k:ck.get(k) for k in ("reads_by_islands","reads_chained_serially_equal_keys","serial_reads","phase_ms_per_read")})
PY
for wpe in 3; do BM2_CHAIN_ISL_WPE=$wpe timeout 300 python bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpe $wpe', d['value'], d['stage_ms_per_step'])"; done; at wpe $?
