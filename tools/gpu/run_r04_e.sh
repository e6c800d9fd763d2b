#!/bin/bash
# Round 4, call E: S1 through the lane kernel (config 2), the tail's batches over several contexts, config 5 again.
TAG=${1:-r04e}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 120 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw.json 2> $O/bench_bsw.err; at bsw $?
python -c "
import json; d=json.load(open('$O/bench_bsw.json')); print('bsw', d['extend_kernel'], d['parity'], d['cpu_baseline']['value'])"
BM2_BSW_LANES=0 timeout 120 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw_wave_only.json 2> $O/bench_bsw_wave_only.err; at bsw_wave $?
python -c "
import json; d=json.load(open('$O/bench_bsw_wave_only.json')); print('bsw (pair per wavefront)', d['extend_kernel'])"
timeout 400 python -m pytest tests/test_bsw_gpu.py tests/test_sharded.py tests/test_zz_tail_kernels_gpu.py tests/test_end_to_end_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
echo "finished at $(( $(date +%s) - T0 ))s"
