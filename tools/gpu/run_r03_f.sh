#!/bin/bash
# Round 3, call F: which split of the 16 CPUs over the end-to-end stages (tail workers x threads, parser threads, device workers)?
TAG=${1:-r03f}; LIMIT=${2:-420}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
timeout 200 python bench.py --steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-e2e > $O/bench_short.json 2> $O/bench_short.err; at bench $?
python -c "import json; d=json.load(open('$O/bench_short.json')); print('value %.2f M, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
V=""
for spec in "3 5 6 2" "3 6 6 2" "3 4 8 2" "4 4 6 2" "2 8 6 2" "3 5 6 3" "3 5 4 2" "3 5 6 1"; do
  set -- $spec; V="$V BM2_E2E_TAILS=$1,BM2_E2E_TAIL_THREADS=$2,BM2_E2E_PARSE_THREADS=$3,BM2_E2E_DEVS=$4"
done
PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=60 PROBE_ENVS="$V" timeout $(( $(left) - 10 )) python tools/gpu/tail_probe.py $O 3100 10 500000 > $O/probe.out 2> $O/probe.err
at probe $?; grep "\[probe\]" $O/probe.err | tail -14
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_periods|nr_throttled|throttled_usec" | tr '\n' ' '; echo
