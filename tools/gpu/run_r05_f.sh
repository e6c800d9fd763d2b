#!/bin/bash
# Round 5, call F: the bench as the driver runs it (configs 5 and 2 and the CLI legs inside), the whole GPU suite, then the rocprofv3 passes whose
# summaries go to profiles/ (kernel trace + timeline, FETCH / WRITE, SQ of the 150 bp workload; kernel trace of config 5 at 20 000 reads).
#   gpurun --timeout 2400 -- 'bash tools/gpu/run_r05_f.sh r05f 2300'
TAG=${1:-r05f}; LIMIT=${2:-2300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 1300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; at bench $?
grep "^\[bench\]" $O/bench.err | tail -14 | cut -c1-420
if [ $(left) -gt 520 ]; then
  timeout 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $(left) -gt 140 ]; then
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 4 > $O/bench_kt.json 2> $O/kt.err; at kt $?
  DB=$(find /tmp/p_kt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1; tail -240 $O/timeline_all.tsv > $O/timeline.tsv; rm -f $O/timeline_all.tsv
fi
if [ $(left) -gt 110 ]; then
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_f.err; at fetch $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
fi
if [ $(left) -gt 110 ]; then
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_w.err; at write $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null 2>> $O/pmc_w.err
fi
if [ $(left) -gt 110 ]; then
  timeout 120 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- $B --steps 4 --warmup 2 > $O/bench_sq1.json 2> $O/pmc_sq1.err; at sq1 $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_sq1.err
fi
if [ $(left) -gt 170 ]; then
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_ont -o s -- python $R/bench.py --workload ont2d --reads 20000 --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_ont2d_kt.json 2> $O/kt_ont.err; at ont_kt $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_ont -name "*.db" | head -1) $O/kernel_trace_ont2d.md > /dev/null 2>> $O/kt_ont.err
  head -24 $O/kernel_trace_ont2d.md
fi
echo "finished at $(( $(date +%s) - T0 ))s"
