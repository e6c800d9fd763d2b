#!/bin/bash
# Round 3, call N: chaining with the light reads in work classes (BM2_PERM_MODE=5) against the plain light-first partition (4): parity test, then timing
TAG=${1:-r03n}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
BM2_PERM_MODE=5 timeout 200 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
for m in 4 5 5 4; do
  BM2_PERM_MODE=$m timeout 200 python bench.py --steps 8 --warmup 4 --no-parity --no-cpu-baseline --no-e2e > $O/bench_perm$m.json 2> $O/bench_perm$m.err
  python -c "import json; d=json.load(open('$O/bench_perm$m.json')); print('PERM_MODE=$m: value %.2f M, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})"
done
cd /tmp
BM2_PERM_MODE=5 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --steps 4 --warmup 4 > $O/bench_kt.json 2> $O/kt.err
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace_perm5.md > /dev/null 2>> $O/kt.err; grep -E "k_chain|k_class|k_scan" $O/kernel_trace_perm5.md
