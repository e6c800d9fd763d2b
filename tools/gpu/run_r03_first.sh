#!/bin/bash
# FIRST gpurun call of round 3: everything the last session of round 2 changed without a GPU, in the order of what is least known.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r03_first.sh r03a 880'
# Afterwards: copy gpurun_out/r03a/* summaries into profiles/ and COMMIT in the same turn (round 2 lost 80 GPU-minutes of results).
TAG=${1:-r03a}; LIMIT=${2:-880}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
# 1. the bench as the driver runs it (index build ~130 s, parity gate, CPU baseline, end-to-end leg with 2 device / 3 tail workers)
timeout 560 python bench.py --steps 5 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? at $(( $(date +%s) - T0 ))s" | tee -a $O/bench_full.err
grep -E "parity|end-to-end|cpu baseline" $O/bench_full.err | tail -8
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
    print("parity", d.get("parity")); print("end_to_end", json.dumps(d.get("end_to_end"))[:900])
except Exception as e:
    print("no bench line:", e)
P
# 2. the end-to-end leg on a 128 Mbp genome under the alternatives (one device worker, no pinning, two tail workers)
if [ $(left) -gt 150 ]; then
  PROBE_ENVS="BM2_E2E_DEVS=1 BM2_TAIL_PIN=0 BM2_E2E_TAILS=2 BM2_E2E_TAILS=4" BM2_TAIL_PROF= timeout 140 python tools/gpu/tail_probe.py $O 128 4 > $O/probe.out 2> $O/probe.err
  echo "probe rc=$? at $(( $(date +%s) - T0 ))s"; grep "\[probe\]" $O/probe.err | tail -10
fi
# 3. launch-policy sweep (BM2_CHAIN_STAGE, BM2_SAL_QUAD, BM2_BWD_* ... are in its grid) on the resident chunk
if [ $(left) -gt 120 ]; then
  timeout $(( $(left) - 60 )) python tools/gpu/sweep.py $O --steps 4 --budget-s 60 > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s"
  grep "\[sweep\]" $O/sweep.err | tail -30
fi
# 3b. config 2 (the S1 kernel alone, batch resident)
cd $R
if [ $(left) -gt 60 ]; then
  timeout 50 python bench.py --workload bsw --steps 5 --warmup 2 > $O/bench_bsw.json 2> $O/bench_bsw.err; echo "bsw rc=$? at $(( $(date +%s) - T0 ))s"; head -c 600 $O/bench_bsw.json; echo
fi
# 4. kernel trace of the defaults (k_ext_lanes: did the shorter DP cell pay?)
cd /tmp
if [ $(left) -gt 45 ]; then
  timeout $(( $(left) - 5 )) rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --steps 4 --warmup 1 > $O/bench_kt.json 2> $O/kt.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null 2>> $O/kt.err; head -14 $O/kernel_trace.md
fi
echo "finished at $(( $(date +%s) - T0 ))s"
