#!/bin/bash
# short, individually time-limited GPU steps for debugging (every step under its own small timeout)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-dbg}; mkdir -p $O; cd $R; export TMPDIR=/tmp
shift
i=0
for CMD in "$@"; do
  i=$((i+1))
  echo "=== step $i: $CMD" | tee -a $O/steps.log
  ( eval "$CMD" ) > $O/step$i.out 2> $O/step$i.err; echo "rc=$?" | tee -a $O/steps.log
  tail -15 $O/step$i.out | cut -c1-400; tail -25 $O/step$i.err | cut -c1-400
done
