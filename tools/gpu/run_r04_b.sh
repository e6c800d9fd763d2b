#!/bin/bash
# Round 4, call B: the bench as the driver runs it (with the config-5 / config-2 legs inside), then a kernel trace of the ont2d workload.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r04_b.sh r04b 880'
TAG=${1:-r04b}; LIMIT=${2:-880}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 700 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "parity|end-to-end|cpu baseline|index built|genome|binding|ont2d|bsw|S1" $O/bench_full.err | tail -30
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
    print("parity", json.dumps(d.get("parity"))[:600]); print("end_to_end", json.dumps(d.get("end_to_end"))[:700])
    print("config5", json.dumps(d.get("config5"))[:1500]); print("config2", json.dumps(d.get("config2"))[:1500]); print("binding", json.dumps(d.get("binding"))[:500])
except Exception as e:
    print("no bench line:", e)
P
cd /tmp
if [ $(left) -gt 120 ]; then
  timeout 110 rocprofv3 --kernel-trace --stats -d /tmp/p_ont -o kt -- python $R/bench.py --workload ont2d --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d_kt.json 2> $O/ont_kt.err; at ont_kt $?
  DB=$(find /tmp/p_ont -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace_ont2d.md > /dev/null 2>> $O/ont_kt.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_ont_all.tsv >> $O/ont_kt.err 2>&1
  tail -200 $O/timeline_ont_all.tsv > $O/timeline_ont2d.tsv; rm -f $O/timeline_ont_all.tsv
  head -24 $O/kernel_trace_ont2d.md
fi
echo "finished at $(( $(date +%s) - T0 ))s"
