#!/bin/bash
# Round 5, call O: config 5's per-dispatch timeline with k_chain_serial beside the island kernel
TAG=${1:-r05o}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
B="python $R/bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_ont -o s -- $B > $O/bench_ont2d_kt.json 2> $O/kt.err; at kt $?
DB=$(find /tmp/p_ont -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/kernel_trace_ont2d.md > /dev/null 2>> $O/kt.err
python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1
tail -150 $O/timeline_all.tsv > $O/timeline_ont2d.tsv; rm -f $O/timeline_all.tsv
awk -F'\t' '$3 > 3.0' $O/timeline_ont2d.tsv | cut -c1-160
