#!/bin/bash
# Round 6: (1) sweep of the chain-finish variants (k_chain_finish_wave, fused finish), (2) the CIGAR kernels with eight bases per load: tail probe under the
# kernel trace + the tail kernels' GPU tests, (3) the pipeline knob tests.
#   gpurun --timeout 1500 -- 'bash tools/gpu/run_r06_ad.sh r06ad'
TAG=${1:-r06ad}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R; export TMPDIR=/tmp
timeout 400 python tools/gpu/sweep.py $O --steps 4 --budget-s 250 --only "k_chain_finish" > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s"
grep "\[sweep\]" $O/sweep.err | tail -8 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"\[sweep\] (.*?): ([\d.]+) ms/step.*?'chain': ([\d.]+)\",l)
    print(m.group(1)[-110:], m.group(2), 'chain', m.group(3)) if m else print(l[:200].rstrip())"
bash tools/gpu/run_r06_ac.sh $TAG
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "knobs or heavy or long" > $O/pytest_knobs.log 2>&1; echo "pytest knobs rc=$? at $(( $(date +%s) - T0 ))s"; tail -3 $O/pytest_knobs.log
