#!/bin/bash
# Round 6 (a TRIAL build with bm2_device_local_cpus and --no-numa-bind, not kept: profiles/r06al_*): the FASTQ -> SAM leg with the process bound to the hardware threads next to its GPU (bench.py's default now) and not (--no-numa-bind), alternating processes.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_al.sh r06al 1150'
TAG=${1:-r06al}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
i=0
for cfg in bind nobind bind nobind bind nobind; do
  i=$((i+1)); extra=""; [ $cfg = nobind ] && extra="--no-numa-bind"
  if [ $(left) -gt 150 ]; then
    timeout 300 python bench.py --steps 8 --warmup 4 --no-parity --no-cpu-baseline --no-side-workloads --no-binding $extra --full-json $O/bench_${i}_$cfg.json > /dev/null 2> $O/bench_${i}_$cfg.err
    echo "== $cfg rc=$? at $(( $(date +%s) - T0 ))s: $(grep 'bound to' $O/bench_${i}_$cfg.err | cut -c9-120)"
    python3 -c "
import json; d=json.load(open('$O/bench_${i}_$cfg.json')); e=d['end_to_end']
print('  hot %.2f ms | e2e %.2f M (steady %.2f) | %s | cpu %.3f' % (d['ms_per_step'], e['value']/1e6, e['steady_state']['reads_per_s']/1e6, {k: round(v,1) for k,v in e['stage_ms_per_chunk'].items()}, e['host_cpu_s_per_chunk']))"
  fi
done
echo "finished at $(( $(date +%s) - T0 ))s"
