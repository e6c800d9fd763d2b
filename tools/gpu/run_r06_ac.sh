#!/bin/bash
# Round 6: the CIGAR kernels with their sequences fetched eight bases per load (RefPtr::nib8) -- kernel trace of the tail probe (128 Mbp genome, 2 chunks of
# 1 M reads; compare profiles/r06n_tail_kernel_trace_cigar_stores_on.md: k_gen_cigar<2> 8.71, k_cigar_flat 3.22, k_gen_cigar<1> 1.92 ms per launch), then the
# CIGAR / tail GPU tests.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_ac.sh r06ac'
TAG=${1:-r06ac}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd /tmp; export TMPDIR=/tmp
PROBE_LIMIT_S=100 timeout 250 rocprofv3 --kernel-trace -d /tmp/p_tail -o t -- python $R/tools/gpu/tail_probe.py $O 128 2 500000 > $O/tail_probe.out 2> $O/tail_probe.err
echo "probe rc=$? at $(( $(date +%s) - T0 ))s"
python $R/tools/rocpd_summary.py $(find /tmp/p_tail -name "*.db" | head -1) $O/tail_kernel_trace.md > /dev/null 2>> $O/tail_probe.err
grep -i "cigar\|ksw" $O/tail_kernel_trace.md | cut -c1-110
grep "\[probe\]" $O/tail_probe.err | cut -c1-200
cd $R
timeout 500 python -m pytest tests/test_zz_tail_kernels_gpu.py -x -q -m gpu > $O/pytest_tail.log 2>&1; echo "pytest tail rc=$? at $(( $(date +%s) - T0 ))s"; tail -3 $O/pytest_tail.log
