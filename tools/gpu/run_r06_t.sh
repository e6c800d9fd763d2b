#!/bin/bash
# Round 6: L1 / L2 request counters of the hot path's kernels (what does k_chain move through the caches?).
#   gpurun --timeout 900 -- 'bash tools/gpu/run_r06_t.sh r06t 850'
TAG=${1:-r06t}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\(TCC\|TCP\)_[A-Z_]*\(sum\)\?" | sort -u | tr '\n' ' ' | cut -c1-3000 > $O/avail_tcc_tcp.txt; wc -c $O/avail_tcc_tcp.txt
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
timeout 300 $B --steps 1 --warmup 1 > /dev/null 2> $O/prep.err; at prep $?
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  tag=$(echo $set | cut -c1-3 | tr 'A-Z' 'a-z')
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/p_$tag -o c -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_$tag.err; at "pmc $tag" $?
  DB=$(find /tmp/p_$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/pmc_$tag.md > /dev/null 2>> $O/pmc_$tag.err
  tail -3 $O/pmc_$tag.err | cut -c1-200
done
grep -h "k_chain \|k_chain_heavy\|k_bwd<\|k_ext_seeds<" $O/pmc_tcc.md $O/pmc_tcp.md 2>/dev/null | tail -40 | cut -c1-140
echo "finished at $(( $(date +%s) - T0 ))s"
