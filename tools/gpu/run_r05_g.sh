#!/bin/bash
# Round 5, call G: config 2 once more (the S1 binding with a sleeping leader, -t quota - 2), twice in a row to see the spread.
TAG=${1:-r05g}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-e2e --no-side-workloads > $O/bench_min.json 2> $O/bench_min.err; at genome_and_index $?
for k in 1 2; do
  timeout 300 python bench.py --workload bsw --steps 3 --warmup 1 > $O/bench_bsw_$k.json 2> $O/bench_bsw_$k.err; at bsw$k $?
  python - <<P
import json
try:
    d = json.load(open("$O/bench_bsw_$k.json")); s = d["s1_binding"]
    print("run $k: config2 %.0f G cells/s; reference %.3f s (-t %s), bm2s1 %.3f s (-t %s): ratio %.3f; floor %.3f; own clocks %s | %s; batches %s of %s calls; SAM equal %s"
          % (d["extend_kernel"]["gcups"], s["reference"]["chunk_real_s"], s["threads"], s["bm2s1"]["chunk_real_s"], s.get("threads_bm2s1"), s["bm2s1"]["chunk_real_s"] / s["reference"]["chunk_real_s"],
             s["seam_share"]["floor_of_bm2s1_over_reference"], s["reference"]["own_clocks"], s["bm2s1"]["own_clocks"], s["bm2s1"].get("device_batches"), s["bm2s1"].get("calls"), s["sam_equal"]))
except Exception as e:
    print("run $k: no line:", e)
P
done
