#!/bin/bash
# Round 6: the FASTQ -> SAM leg's host CPU by kind of thread (stage workers, the parser's and the tail workers' pools, the HIP runtime's threads).
#   gpurun --timeout 700 -- 'bash tools/gpu/run_r06_ai.sh r06ai'
TAG=${1:-r06ai}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 400 python bench.py --steps 8 --warmup 4 --no-parity --no-cpu-baseline --no-side-workloads --no-binding --full-json $O/bench.json > /dev/null 2> $O/bench.err; echo "rc=$?"
grep "end_to_end (FASTQ" $O/bench.err | cut -c1-260
python3 -c "
import json; d=json.load(open('$O/bench.json')); e=d['end_to_end']
print(d.get('ms_per_step'), {k: e.get(k) for k in ('value','steady_state','stage_ms_per_chunk','host_cpu_s_per_chunk')})
for k,v in e['host_cpu_s_per_chunk_by_thread_kind'].items(): print('  %-28s %s' % (k, v))"
