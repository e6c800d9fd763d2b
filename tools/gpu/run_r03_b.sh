#!/bin/bash
# Round 3, call B: does the host tail scale with its threads on the box?  (128 Mbp genome: the tail does not care)
TAG=${1:-r03b}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
for tune in 0 1; do
  SCALING_TAG=tune$tune BM2_MALLOC_TUNE=$tune BM2_TAIL_PROF=1 timeout 200 python tools/gpu/tail_scaling.py $O 128 500000 > $O/scaling_tune$tune.out 2> $O/scaling_tune$tune.err
  echo "tune=$tune rc=$?"; grep "\[scaling\] [0-9t]" $O/scaling_tune$tune.err
done
SCALING_TAG=nopin BM2_TAIL_PIN=0 BM2_MALLOC_TUNE=1 BM2_TAIL_PROF=1 SCALING_THREADS="72 216" timeout 120 python tools/gpu/tail_scaling.py $O 128 500000 > $O/scaling_nopin.out 2> $O/scaling_nopin.err
echo "nopin rc=$?"; grep "\[scaling\] [0-9t]" $O/scaling_nopin.err
cat /sys/kernel/mm/transparent_hugepage/enabled /proc/sys/kernel/numa_balancing 2>/dev/null; uname -r; ldd --version | head -1
