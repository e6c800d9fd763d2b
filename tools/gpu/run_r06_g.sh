#!/bin/bash
# Round 6, seventh call: the side streams reordered by measured hardware-queue class (bm2_side_streams: BM2_QUEUE_PROBE) -- the hot path as the driver runs it
# (20 steps over four resident contexts) with the probe off and on, two processes each way, then a per-dispatch timeline of the probed run.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_g.sh r06g 850'
TAG=${1:-r06g}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
for rep in 1 2; do
  for p in 0 1; do
    BM2_QUEUE_PROBE=$p BM2_QUEUE_PROBE_LOG=1 timeout 200 $B --steps 20 --warmup 5 --full-json $O/bench_probe${p}_$rep.json > /dev/null 2> $O/probe${p}_$rep.err; at probe${p}_$rep $?
    grep "hardware-queue classes" $O/probe${p}_$rep.err | head -4
    grep "^\[bench\] hot path" $O/probe${p}_$rep.err | tail -1 | cut -c1-170
  done
done
if [ $(left) -gt 150 ]; then
  BM2_QUEUE_PROBE_LOG=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 8 --warmup 4 --full-json $O/bench_kt.json > /dev/null 2> $O/kt.err; at kt $?
  DB=$(find /tmp/p_kt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1
  python3 - <<PY
rows = open("$O/timeline_all.tsv").read().split("\n")
hdr, rows = rows[0], [r for r in rows[1:] if r]
idx = [i for i, r in enumerate(rows) if "k_walk<1>" in r]
open("$O/timeline.tsv", "w").write("\n".join([hdr] + rows[idx[-1] - 3:]) + "\n")
PY
  rm -f $O/timeline_all.tsv
  grep "hardware-queue classes" $O/kt.err | head -4
  grep "^\[bench\] hot path" $O/kt.err | tail -1 | cut -c1-170
  grep "k_ext\|k_advance\|k_reg_init" $O/timeline.tsv | cut -c1-110
fi
echo "finished at $(( $(date +%s) - T0 ))s"
