#!/bin/bash
# Round 3, call G: the reshaped CIGAR kernels (flat / ring / row / global) on the GPU: tests, then kernel times on a 1 M-read chunk (128 Mbp).
TAG=${1:-r03g}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_zz_tail_kernels_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp
for ring in 0 1; do
  SCALING_TAG=cg_noring$ring BM2_CIGAR_NO_RING=$ring BM2_TAIL_PROF=1 SCALING_THREADS="12" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_cg$ring -o kt -- python $R/tools/gpu/tail_scaling.py $O 128 500000 > $O/cg$ring.out 2> $O/cg$ring.err
  echo "no_ring=$ring rc=$?"
  python $R/tools/rocpd_summary.py $(find /tmp/p_cg$ring -name "*.db" | head -1) $O/kernel_trace_noring$ring.md > /dev/null 2>> $O/cg$ring.err
  grep -E "gen_cigar|ksw_align|cigar_flat|cigar_compact|cigar_sizes" $O/kernel_trace_noring$ring.md
  grep -E "by shape|H2D \+ kernel" $O/cg$ring.err | tail -4
done
