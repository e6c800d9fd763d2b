#!/bin/bash
# Round 6: the whole GPU suite + smoke() on the tree as it stands (after the closing call: the S1 binding's eight slots, comments, docs).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_check.sh r06zz 850'
TAG=${1:-r06zz}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
