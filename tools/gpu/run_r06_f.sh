#!/bin/bash
# Round 6, sixth call: kernel trace + per-dispatch timeline of the HEADLINE workload on the current code; SQ counters of the tail kernels (FASTQ -> SAM leg on a
# 128 Mbp genome): what the CIGAR kernels and the register-row rescue kernel wait for.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_f.sh r06f 850'
TAG=${1:-r06f}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 2 --full-json $O/bench_kt.json > $O/bench_kt.line 2> $O/kt.err; at kt $?
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1
python3 - <<PY
rows = open("$O/timeline_all.tsv").read().split("\n")
hdr, rows = rows[0], [r for r in rows[1:] if r]
# the last step of the run: from the last k_walk<1> on
idx = [i for i, r in enumerate(rows) if "k_walk<1>" in r]
open("$O/timeline.tsv", "w").write("\n".join([hdr] + rows[idx[-1] - 3:]) + "\n")
print("timeline rows of the last step:", len(rows) - idx[-1] + 3)
PY
rm -f $O/timeline_all.tsv
grep "^\[bench\] hot path" $O/kt.err | tail -1 | cut -c1-300
head -24 $O/kernel_trace.md | cut -c1-120
if [ $(left) -gt 250 ]; then
  PROBE_LIMIT_S=100 timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/p_tsq -o t -- python $R/tools/gpu/tail_probe.py $O 128 2 500000 > $O/tail_probe_pmc.out 2> $O/tail_probe_pmc.err; at tail_pmc $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_tsq -name "*.db" | head -1) $O/tail_pmc_sq.md > /dev/null 2>> $O/tail_probe_pmc.err
  grep -i "ksw\|cigar\|kernel \|---" $O/tail_pmc_sq.md | cut -c1-260 | head -20
fi
echo "finished at $(( $(date +%s) - T0 ))s"
