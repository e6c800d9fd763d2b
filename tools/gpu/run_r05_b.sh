#!/bin/bash
# Round 5, calls B and C: the hand-over with its own continuation kernel (k_bwd_cont), the rare steps of the seeding lanes in every (mask + 1)-th round,
# chain filter with wavefront-scope fences: parity test of every setting, then timed on the 3100 Mbp bench chunk in one process, then a kernel trace
# of the bench with the sweep's best settings.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r05_b.sh r05b'
TAG=${1:-r05b}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 500 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "off_by_default" > $O/pytest_knobs.log 2>&1; at pytest $?
tail -5 $O/pytest_knobs.log
timeout 420 python tools/gpu/sweep.py $O --steps 4 --budget-s 300 --only "seeding: k_bwd hands,kept-chain walk,chain clock" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -40
cd /tmp
set -a; [ -f $O/best_env.sh ] && . $O/best_env.sh; set +a
cat $O/best_env.sh
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --no-binding"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 4 > $O/bench_kt.json 2> $O/kt.err; at kt $?
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1; tail -160 $O/timeline_all.tsv > $O/timeline.tsv; rm -f $O/timeline_all.tsv
head -30 $O/kernel_trace.md
python - <<P
import json
d = json.load(open("$O/bench_kt.json"))
print("bench (traced): %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "roofline frac", d["roofline"]["frac"])
P
