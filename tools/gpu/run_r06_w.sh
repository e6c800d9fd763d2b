#!/bin/bash
# Round 6: the FASTQ -> SAM leg with the host's waits for the device asleep instead of spinning (BM2_BLOCKING_SYNC=1: the device workers and the tail
# workers' batches each hold a CPU of 16 while they wait), A/B in alternating processes.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_w.sh r06w 1150'
TAG=${1:-r06w}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
i=0
for cfg in ${CFGS:-0 1 0 1 1:POOL0 1:TAIL8}; do       # BM2_BLOCKING_SYNC value [: extra]
  i=$((i+1)); bs=${cfg%%:*}; extra=""; case $cfg in *:POOL0) extra="BM2_POOL_SPIN_US=0";; *:TAIL8) extra="BM2_E2E_TAIL_THREADS=8";; esac
  if [ $(left) -gt 150 ]; then
    env BM2_BLOCKING_SYNC=$bs $extra timeout 300 python bench.py --steps 8 --warmup 4 --no-parity --no-cpu-baseline --no-side-workloads --no-binding --full-json $O/bench_${i}_$bs.json > /dev/null 2> $O/bench_${i}_$bs.err
    echo "== $cfg rc=$? at $(( $(date +%s) - T0 ))s"
    grep "end_to_end (FASTQ" $O/bench_${i}_$bs.err | cut -c1-260
    python3 -c "
import json; d=json.load(open('$O/bench_${i}_$bs.json')); e=d['end_to_end']
print('  ', d.get('ms_per_step'), {k: e.get(k) for k in ('value','steady_state','stage_ms_per_chunk','host_cpu_s_per_chunk')})"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
