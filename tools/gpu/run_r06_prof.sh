#!/bin/bash
# Round 6: the rocprofv3 passes whose summaries go to profiles/ (tools/pmc_to_profiles.py <out> r06) on the round's last device code: an un-profiled bench run
# (its full record: bench.json), kernel trace + timeline, FETCH_SIZE, WRITE_SIZE + SQ counters (separate passes; --pmc never with the trace domains gpurun refuses).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_prof.sh r06p 850'
TAG=${1:-r06p}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
timeout 200 $B --steps 8 --warmup 4 --full-json $O/bench.json > $O/bench.line 2> $O/bench.err; at bench $?
grep "^\[bench\] hot path" $O/bench.err | tail -1 | cut -c1-220
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 8 --warmup 4 --full-json $O/bench_kt.json > /dev/null 2> $O/kt.err; at kt $?
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
head -16 $O/kernel_trace.md | cut -c1-110
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 1 --warmup 1 --full-json $O/bench_pmc_f.json > /dev/null 2> $O/pmc_f.err; at fetch $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
fi
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 1 --warmup 1 --full-json $O/bench_pmc_w.json > /dev/null 2> $O/pmc_w.err; at write $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null 2>> $O/pmc_w.err
fi
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- $B --steps 1 --warmup 1 --full-json $O/bench_pmc_sq1.json > /dev/null 2> $O/pmc_sq1.err; at sq1 $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_sq1.err
fi
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/p_sq2 -o s -- $B --steps 1 --warmup 1 --full-json $O/bench_pmc_sq2.json > /dev/null 2> $O/pmc_sq2.err; at sq2 $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) $O/pmc_sq2.md > /dev/null 2>> $O/pmc_sq2.err
fi
ls -la $O | head -30
echo "finished at $(( $(date +%s) - T0 ))s"
