#!/bin/bash
# Round 3, call A: the bench at HEAD with the tail's phase clock, then the end-to-end leg under a kernel trace on the SAME 3100 Mbp index
# (it stays on the box's disk between the processes of one call), config 5 and config 2 lines, the launch-policy sweep.
#   gpurun --timeout 1000 -- 'bash tools/gpu/run_r03_a.sh r03a 980'
TAG=${1:-r03a}; LIMIT=${2:-980}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt; free -g >> $O/host.txt
# 1. the bench as the driver runs it, phase clock of the tail on
BM2_TAIL_PROF=1 timeout 520 python bench.py --steps 10 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "parity|end-to-end|cpu baseline|index built|genome" $O/bench_full.err | tail -8
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
    print("parity", d.get("parity")); print("end_to_end", json.dumps(d.get("end_to_end"))[:900])
except Exception as e:
    print("no bench line:", e)
P
grep "\[tail\]" $O/bench_full.err | tail -60 > $O/tail_phases_bench.txt
# 2. the end-to-end leg alone under a kernel trace, same index; then ONE device worker / ONE tail worker (clean phase clock), then variants
cd /tmp
if [ $(left) -gt 200 ]; then
  PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=120 BM2_TAIL_PROF=1 \
  PROBE_ENVS="BM2_E2E_DEVS=1,BM2_E2E_TAILS=1,BM2_TAIL_PROF=1 BM2_E2E_DEVS=1 BM2_E2E_DEVS=1,BM2_E2E_TAILS=2 BM2_TAIL_PIN=0" \
  timeout 190 rocprofv3 --kernel-trace --stats -d /tmp/p_e2e -o kt -- python $R/tools/gpu/tail_probe.py $O 3100 4 500000 > $O/probe.out 2> $O/probe.err
  at probe $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_e2e -name "*.db" | head -1) $O/e2e_kernel_trace.md > /dev/null 2>> $O/probe.err
  grep "\[probe\]" $O/probe.err | tail -12; head -30 $O/e2e_kernel_trace.md
  grep "\[tail\]" $O/probe.err > $O/tail_phases_probe.txt
fi
cd $R
# 3. config 5 (10 kb reads, -x ont2d) with its parity gate
if [ $(left) -gt 120 ]; then
  T=$(( $(left) - 90 )); [ $T -gt 170 ] && T=170
  timeout $T python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 200 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont2d $?
  tail -5 $O/bench_ont2d.err; head -c 900 $O/bench_ont2d.json; echo
fi
# 4. config 2 (S1 alone, resident batch)
if [ $(left) -gt 60 ]; then
  timeout 50 python bench.py --workload bsw --steps 5 --warmup 2 > $O/bench_bsw.json 2> $O/bench_bsw.err; at bsw $?; head -c 700 $O/bench_bsw.json; echo
fi
# 5. launch-policy sweep on the resident chunk
if [ $(left) -gt 80 ]; then
  timeout $(( $(left) - 10 )) python tools/gpu/sweep.py $O --steps 4 --budget-s $(( $(left) - 50 )) > $O/sweep.out 2> $O/sweep.err; at sweep $?
  grep "\[sweep\]" $O/sweep.err | tail -45
fi
echo "finished at $(( $(date +%s) - T0 ))s"
