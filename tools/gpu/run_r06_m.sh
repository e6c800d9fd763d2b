#!/bin/bash
# Round 6: the CIGAR kernels after the round's changes (loads requested eight at a time in the flat kernel and the MD pass, kputw by constant divisors, the
# DP loop's LDS words requested a column ahead): GPU tests, per-kernel times on the probe's chunks (compare profiles/r06d_tail_kernel_trace_ksw_reg_ab.md), then
# the FASTQ -> SAM leg over 100 chunks three times.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_m.sh r06m 1150'
TAG=${1:-r06m}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python -m pytest tests/test_zz_tail_kernels_gpu.py tests/test_end_to_end_gpu.py -m gpu -x -q > $O/pytest_tail.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tail.log; tail -3 $O/pytest_tail.log
cd /tmp
PROBE_LIMIT_S=100 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tail -o t -- python $R/tools/gpu/tail_probe.py $O 128 3 1000000 > $O/tail_probe.out 2> $O/tail_probe.err; at tail_probe $?
grep "\[probe\]" $O/tail_probe.err | cut -c1-200
python $R/tools/rocpd_summary.py $(find /tmp/p_tail -name "*.db" | head -1) $O/tail_kernel_trace.md > /dev/null 2>> $O/tail_probe.err
grep -i "ksw\|cigar" $O/tail_kernel_trace.md | cut -c1-110
cd $R
CFGS="3x7 3x7 3x7" bash tools/gpu/run_r06_j.sh ${TAG}_e2e $(( $(left) - 20 )) 2>&1 | grep -v "^   {"
echo "finished at $(( $(date +%s) - T0 ))s"
