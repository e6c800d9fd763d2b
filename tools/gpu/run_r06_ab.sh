#!/bin/bash
# Round 6: kernel trace of the chaining stage with k_chain_finish's part fused into k_chain (default) and not (where does k_chain_finish's 1.35 ms go?).
#   gpurun --timeout 700 -- 'bash tools/gpu/run_r06_ab.sh r06ab'
TAG=${1:-r06ab}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --no-binding"
timeout 200 $B --steps 2 --warmup 1 > /dev/null 2> $O/prep.err; echo "prep rc=$? at $(( $(date +%s) - T0 ))s"
for f in 1 0; do
  BM2_CHAIN_FUSE_FINISH=$f timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt$f -o kt -- $B --steps 4 --warmup 2 > /dev/null 2> $O/kt$f.err; echo "kt fuse=$f rc=$? at $(( $(date +%s) - T0 ))s"
  DB=$(find /tmp/p_kt$f -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace_fuse$f.md > /dev/null 2>> $O/kt$f.err
  grep "k_chain\|k_slot_base\|k_advance\|k_read_base\|k_class" $O/kernel_trace_fuse$f.md | head -12 | cut -c1-120
done
