#!/bin/bash
# Round 6: which of the legs before it costs the FASTQ -> SAM leg its 6 % inside the driver's whole run?  The bench with the legs switched off one at a time.
#   gpurun --timeout 1500 -- 'bash tools/gpu/run_r06_am.sh r06am 1450'
TAG=${1:-r06am}; LIMIT=${2:-1450}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
i=0
for cfg in "--no-side-workloads --no-binding" "--no-side-workloads --no-binding --no-cpu-baseline" "--no-side-workloads --no-binding --no-parity" "--no-side-workloads --no-binding --no-parity --no-cpu-baseline"; do
  i=$((i+1))
  if [ $(left) -gt 330 ]; then
    timeout 320 python bench.py --steps 8 --warmup 4 $cfg --full-json $O/bench_$i.json > /dev/null 2> $O/bench_$i.err
    echo "== [$cfg] rc=$? at $(( $(date +%s) - T0 ))s"
    python3 -c "
import json; d=json.load(open('$O/bench_$i.json')); e=d['end_to_end']
print('  hot %.2f ms | e2e %.2f M (steady %.2f) | %s | cpu %.3f' % (d['ms_per_step'], e['value']/1e6, e['steady_state']['reads_per_s']/1e6, {k: round(v,1) for k,v in e['stage_ms_per_chunk'].items()}, e['host_cpu_s_per_chunk']))"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
