#!/bin/bash
# Round 6: how much the hot path varies from process to process (the hardware-queue lottery), probe on and off, six processes each way; then the CIGAR ring
# kernel without its direction stores (a timing experiment: BM2_CIGAR_DBG_NOZ=1).
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_n.sh r06n 1150'
TAG=${1:-r06n}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
for rep in 1 2 3 4 5 6; do
  for p in 1 0; do
    if [ $(left) -gt 200 ]; then
      BM2_QUEUE_PROBE=$p BM2_QUEUE_PROBE_LOG=1 timeout 200 $B --steps 20 --warmup 5 --full-json $O/bench_probe${p}_$rep.json > /dev/null 2> $O/probe${p}_$rep.err
      echo "probe=$p rep=$rep: $(grep '^\[bench\] hot path' $O/probe${p}_$rep.err | tail -1 | cut -c9-120) | $(grep 'hardware-queue classes' $O/probe${p}_$rep.err | sed 's/.*(main stream: //' | tr '\n' ';' | cut -c1-200)"
    fi
  done
done
if [ $(left) -gt 150 ]; then
  for z in 0 1; do
    BM2_CIGAR_DBG_NOZ=$z PROBE_LIMIT_S=100 timeout 200 rocprofv3 --kernel-trace -d /tmp/p_tail$z -o t -- python $R/tools/gpu/tail_probe.py $O 128 2 500000 > $O/tail_probe_noz$z.out 2> $O/tail_probe_noz$z.err
    python $R/tools/rocpd_summary.py $(find /tmp/p_tail$z -name "*.db" | head -1) $O/tail_kernel_trace_noz$z.md > /dev/null 2>> $O/tail_probe_noz$z.err
    echo "BM2_CIGAR_DBG_NOZ=$z:"; grep -i "cigar" $O/tail_kernel_trace_noz$z.md | cut -c1-100
  done
fi
echo "finished at $(( $(date +%s) - T0 ))s"
