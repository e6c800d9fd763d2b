#!/bin/bash
# A two-minute gpurun call: the end-to-end leg on a 128 Mbp genome under rocprofv3 --kernel-trace with the tail's phase clock on.
#   gpurun --timeout 115 -- 'bash tools/gpu/run_tail.sh r02c'
TAG=${1:-r02c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BM2_TAIL_PROF=1 timeout 105 rocprofv3 --kernel-trace --stats -d /tmp/p_tail -o kt -- python $R/tools/gpu/tail_probe.py $O 128 3 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?" >> $O/probe.err
python $R/tools/rocpd_summary.py $(find /tmp/p_tail -name "*.db" | head -1) $O/kernel_trace_tail.md > /dev/null 2>> $O/probe.err
grep -v "^\[tail\]" $O/probe.err | tail -8; head -c 900 $O/probe.out; echo; head -16 $O/kernel_trace_tail.md
