#!/bin/bash
# Round 5, the closing call on the round's last code: the bench as the driver runs it, the whole GPU suite, smoke(), config 5's kernel trace.
#   gpurun --timeout 1800 -- 'bash tools/gpu/run_r05_final.sh r05z 1750'
TAG=${1:-r05z}; LIMIT=${2:-1750}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 1300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; at bench $?
grep "^\[bench\]" $O/bench.err | tail -12 | cut -c1-400
if [ $(left) -gt 400 ]; then
  timeout 380 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
if [ $(left) -gt 120 ]; then
  timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; at smoke $?; tail -2 $O/smoke.log
fi
cd /tmp
if [ $(left) -gt 150 ]; then       # config 5's per-kernel times on the closing code (k_walk<1>'s FETCH pass of profiles/r05_ont2d_k_walk_pmc.json stands: that kernel has not changed)
  timeout 200 rocprofv3 --kernel-trace -d /tmp/p_ok -o s -- python $R/bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d_kt.json 2> $O/kt_ont.err; at ont_kt $?
  DB=$(find /tmp/p_ok -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace_ont2d.md > /dev/null 2>> $O/kt_ont.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt_ont.err 2>&1; tail -120 $O/timeline_all.tsv > $O/timeline_ont2d.tsv; rm -f $O/timeline_all.tsv
  head -14 $O/kernel_trace_ont2d.md | cut -c1-140
fi
echo "finished at $(( $(date +%s) - T0 ))s"
