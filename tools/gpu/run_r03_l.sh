#!/bin/bash
# Round 3, call L: the profiling passes at HEAD -> profiles/r03_* (tools/pmc_to_profiles.py gpurun_out/r03l r03): kernel trace, FETCH_SIZE, WRITE_SIZE + SQ
# counters, second SQ set; the bench line of the trace run.
TAG=${1:-r03l}; LIMIT=${2:-520}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e"
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 1 > $O/bench.json 2> $O/kt.err; at trace $?
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null 2>> $O/kt.err
head -c 500 $O/bench.json; echo; head -12 $O/kernel_trace.md
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_f.err; at fetch $?
python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
timeout 120 rocprofv3 --pmc WRITE_SIZE $SQ1 --kernel-trace -d /tmp/p_ws -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_ws.err; at write_sq1 $?
DB=$(find /tmp/p_ws -name "*.db" | head -1)
if [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/pmc_sq1.md > /dev/null 2>> $O/pmc_ws.err && grep -q SQ_INSTS_VALU $O/pmc_sq1.md && grep -q WRITE_SIZE $O/pmc_sq1.md; then
  cp $O/pmc_sq1.md $O/pmc_write.md
else
  timeout 120 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_sq1.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_sq1.err
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_w.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null 2>> $O/pmc_w.err
  at sq1_write_separately 0
fi
if [ $(left) -gt 60 ]; then
  timeout 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/p_sq2 -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_sq2.err; at sq2 $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) $O/pmc_sq2.md > /dev/null 2>> $O/pmc_sq2.err
fi
grep -E "k_bwd|k_ext_lanes" $O/pmc_sq1.md | head -30
echo "finished at $(( $(date +%s) - T0 ))s"
