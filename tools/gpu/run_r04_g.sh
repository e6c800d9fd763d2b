#!/bin/bash
# Round 4, call G: chaining of long reads by islands: GPU tests, config 5 with its parity gate and a kernel trace.
TAG=${1:-r04g}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -s -k "long_reads" > $O/pytest_long.log 2>&1; echo "pytest rc=$?" >> $O/pytest_long.log; grep -E "island path|passed|failed|rc=" $O/pytest_long.log | tail -6
timeout 280 python bench.py --workload ont2d --no-cpu-baseline --parity-reads 200 --steps 2 --warmup 1 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; at ont2d $?
python - <<P
import json
try:
    d = json.load(open("$O/bench_ont2d.json"))
    print("ont2d: %.0f reads/s, stages %s, parity %s" % (d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}))
except Exception as ex:
    print("no line:", ex)
P
tail -3 $O/bench_ont2d.err
cd /tmp
if [ $(left) -gt 100 ]; then
  timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/p_ont -o kt -- python $R/bench.py --workload ont2d --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d_kt.json 2> $O/ont_kt.err; at ont_kt $?
  DB=$(find /tmp/p_ont -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace_ont2d.md > /dev/null 2>> $O/ont_kt.err
  head -16 $O/kernel_trace_ont2d.md
fi
echo "finished at $(( $(date +%s) - T0 ))s"
