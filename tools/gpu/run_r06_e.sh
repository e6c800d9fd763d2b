#!/bin/bash
# Round 6, fifth call: the wavefront-per-task seeding kernel on a hardware queue of its own (it sat on the main stream's: k_bwd waited for it), its grid; the
# extension's register rows with the new default once more.
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_e.sh r06e 850'
TAG=${1:-r06e}; LIMIT=${2:-850}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 500 python tools/gpu/sweep.py $O --steps 4 --budget-s 220 --only "seeding: stream,extension: rows in registers" > $O/sweep.out 2> $O/sweep.err; at sweep $?
grep "\[sweep\]" $O/sweep.err | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"\[sweep\] (.*?): ([\d.]+) ms/step \{'smem': ([\d.]+), 'smem.walk1': ([\d.]+), 'smem.bwd1': ([\d.]+), 'smem.cont1': ([\d.]+), 'smem.walk2': ([\d.]+), 'smem.bwd2': ([\d.]+).*?'extend': ([\d.]+)\",l)
    print((m.group(1)[-70:]+' step '+m.group(2)+' smem '+m.group(3)+' bwd1 '+m.group(5)+' bwd2 '+m.group(8)+' extend '+m.group(9)) if m else l[:200].rstrip())
"
echo "finished at $(( $(date +%s) - T0 ))s"
