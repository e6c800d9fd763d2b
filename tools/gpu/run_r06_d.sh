#!/bin/bash
# Round 6, fourth call: (1) the GPU tests of the kernels that are new this round (register rows of the extension lanes and of the mate-rescue SW), (2) the
# extension sweep once more with the register kernel allocated for three wavefronts per SIMD, (3) the FASTQ -> SAM leg on a 128 Mbp genome under
# rocprofv3 --kernel-trace, with and without BM2_KSW_REG: what the rescue kernel costs per chunk in either form.
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_d.sh r06d 1150'
TAG=${1:-r06d}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 400 python -m pytest tests/test_zz_tail_kernels_gpu.py tests/test_bsw_gpu.py tests/test_bsw_reference.py tests/test_end_to_end_gpu.py -m gpu -x -q > $O/pytest_new_kernels.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new_kernels.log; tail -3 $O/pytest_new_kernels.log
timeout 300 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "knob" > $O/pytest_knobs.log 2>&1; echo "pytest rc=$?" >> $O/pytest_knobs.log; tail -3 $O/pytest_knobs.log
at tests 0
timeout 400 python tools/gpu/sweep.py $O --steps 4 --budget-s 150 --only "extension: rows in registers" > $O/sweep.out 2> $O/sweep.err; at sweep $?
grep "\[sweep\]" $O/sweep.err | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"\[sweep\] (.*?): ([\d.]+) ms/step.*?'extend': ([\d.]+)\",l)
    print((m.group(1)[-80:]+' '+m.group(2)+' extend '+m.group(3)) if m else l[:160].rstrip())
"
cd /tmp
if [ $(left) -gt 200 ]; then
  PROBE_LIMIT_S=100 PROBE_ENVS="BM2_KSW_REG=0 BM2_KSW_REG=1" timeout 330 rocprofv3 --kernel-trace -d /tmp/p_tail -o t -- python $R/tools/gpu/tail_probe.py $O 128 3 1000000 > $O/tail_probe.out 2> $O/tail_probe.err; at tail_probe $?
  grep "\[probe\]" $O/tail_probe.err | cut -c1-300
  DB=$(find /tmp/p_tail -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/tail_kernel_trace.md > /dev/null 2>> $O/tail_probe.err
  grep -i "ksw\|cigar\|fin_" $O/tail_kernel_trace.md | cut -c1-150
fi
echo "finished at $(( $(date +%s) - T0 ))s"
