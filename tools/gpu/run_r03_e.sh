#!/bin/bash
# Round 3, call E: the allocation-free host tail in the bench; end-to-end variants (workers x threads) on 10 chunks each.
TAG=${1:-r03e}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
BM2_TAIL_PROF=1 timeout 420 python bench.py --steps 10 --warmup 2 --parity-reads 20480 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "parity|end-to-end|cpu baseline|index built" $O/bench_full.err | tail -10
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]))
    e = d.get("end_to_end") or {}
    print("end_to_end %.2f M reads/s" % (e.get("value", 0) / 1e6), {k: v for k, v in e.items() if k not in ("scope",)})
except Exception as e:
    print("no bench line:", e)
P
grep "\[tail\]" $O/bench_full.err | tail -150 > $O/tail_phases.txt
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_periods|nr_throttled|throttled_usec" | tr '\n' ' '; echo
if [ $(left) -gt 120 ]; then
  PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=100 \
  PROBE_ENVS="BM2_E2E_DEVS=1 BM2_E2E_TAILS=3,BM2_E2E_TAIL_THREADS=4 BM2_E2E_TAILS=2,BM2_E2E_TAIL_THREADS=8,BM2_E2E_PARSE_THREADS=3 BM2_E2E_TAILS=1,BM2_E2E_TAIL_THREADS=12" \
  timeout $(( $(left) - 10 )) python tools/gpu/tail_probe.py $O 3100 10 500000 > $O/probe.out 2> $O/probe.err
  at probe $?; grep "\[probe\]" $O/probe.err | tail -12
fi
echo "finished at $(( $(date +%s) - T0 ))s"
