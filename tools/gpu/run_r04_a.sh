#!/bin/bash
# Round 4, call A: the extension stage with one lane per SEED (left + right in one launch, seeds sorted on the device, no host round trip):
# bench + parity gate on a prefix, per-dispatch timeline, the pipeline's GPU tests, sweep of the extension knobs.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r04_a.sh r04a 880'
TAG=${1:-r04a}; LIMIT=${2:-880}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 400 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-e2e --parity-reads 51200 > $O/bench.json 2> $O/bench.err; at bench $?
grep -E "parity|index built|genome" $O/bench.err | tail -6
python - <<P
import json
try:
    d = json.load(open("$O/bench.json"))
    print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
    print("parity", d.get("parity"))
except Exception as e:
    print("no bench line:", e)
P
cd /tmp
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --steps 3 --warmup 4 > $O/bench_kt.json 2> $O/kt.err; at kt $?
  DB=$(find /tmp/p_kt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1
  tail -260 $O/timeline_all.tsv > $O/timeline.tsv; rm -f $O/timeline_all.tsv
  head -24 $O/kernel_trace.md
fi
cd $R
if [ $(left) -gt 150 ]; then
  timeout $(( $(left) - 100 )) python -m pytest tests/test_pipeline_gpu.py tests/test_sharded.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
fi
if [ $(left) -gt 80 ]; then
  timeout $(( $(left) - 10 )) python tools/gpu/sweep.py $O --steps 4 --budget-s $(( $(left) - 50 )) --only extension > $O/sweep.out 2> $O/sweep.err; at sweep $?
  grep "\[sweep\]" $O/sweep.err | tail -30
fi
echo "finished at $(( $(date +%s) - T0 ))s"
