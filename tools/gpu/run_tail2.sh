#!/bin/bash
# the end-to-end leg on a 128 Mbp genome for several splits of the host threads over tail workers (no profiler)
TAG=${1:-r02d}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
PROBE_SPLITS="${PROBE_SPLITS:-4x64 6x40 8x32 4x32 3x85 12x20}" timeout ${PROBE_T:-70} python $R/tools/gpu/tail_probe.py $O 128 ${PROBE_CHUNKS:-3} > $O/probe.out 2> $O/probe.err
echo "probe rc=$?" >> $O/probe.err
grep "\[probe\]" $O/probe.err | tail -12
