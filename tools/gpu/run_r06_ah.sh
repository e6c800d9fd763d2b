#!/bin/bash
# Round 6: (1) sweep of the SA lookup with refilled lanes (k_sal_refill) and its workgroups per CU, (2) the whole GPU suite on the new kernels (k_sal_refill,
# k_chain_finish_wave, the a19 kernels in classes of hit count, the CIGAR ring order), (3) FIN_PERM A/B on the end-to-end leg's a19 stage.
#   gpurun --timeout 2400 -- 'bash tools/gpu/run_r06_ah.sh r06ah'
TAG=${1:-r06ah}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R; export TMPDIR=/tmp
timeout 500 python tools/gpu/sweep.py $O --steps 4 --budget-s 300 --only "SA lookup: lanes" > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s"
grep "\[sweep\]" $O/sweep.err | tail -9 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"\[sweep\] (.*?): ([\d.]+) ms/step.*?'sal': ([\d.]+)\",l)
    print(m.group(1)[-60:], m.group(2), 'sal', m.group(3)) if m else print(l[:200].rstrip())"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$? at $(( $(date +%s) - T0 ))s"; tail -3 $O/pytest_gpu.log
for p in 1 0; do
  BM2_FIN_PERM=$p timeout 300 python bench.py --steps 4 --warmup 2 --no-parity --no-cpu-baseline --no-side-workloads --no-binding --e2e-rounds 3 --full-json $O/bench_finperm$p.json > /dev/null 2> $O/bench_finperm$p.err
  python3 -c "
import json; d=json.load(open('$O/bench_finperm$p.json')); e=d['end_to_end']
print('FIN_PERM=$p', {k: e.get(k) for k in ('value','stage_ms_per_chunk','host_cpu_s_per_chunk')})"
done
echo "finished at $(( $(date +%s) - T0 ))s"
