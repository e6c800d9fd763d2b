#!/bin/bash
# Round 6: seam S1 inside the reference's program (bwa-mem2.bm2s1) under 2 (default) / 4 / 8 device slots: its own clocks say a call waits 2.1 ms for a batch
# that occupies its slot 0.9 ms (profiles/r06z_bench_full.json: config2.s1_binding.bm2s1.seam_clock).
#   gpurun --timeout 1200 -- 'bash tools/gpu/run_r06_r.sh r06r 1150'
TAG=${1:-r06r}; LIMIT=${2:-1150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
timeout 400 python bench.py --steps 1 --warmup 1 --no-parity --no-cpu-baseline --no-e2e --no-side-workloads > /dev/null 2> $O/prep.err; at "genome + index" $?
for n in ${SLOTS:-2 4 8 3}; do
  BM2_S1_CONTEXTS=$n timeout 400 python bench.py --workload bsw --steps 3 --warmup 1 --full-json $O/bench_bsw_slots$n.json > /dev/null 2> $O/bsw_slots$n.err; at "slots $n" $?
  python3 -c "
import json; s=json.load(open('$O/bench_bsw_slots$n.json'))['s1_binding']
print('  slots $n: bm2s1 chunk %.2f s (reference %.2f s), bsw_s %.2f vs %.2f, batches %s, seam clock %s, SAM equal %s' % (s['bm2s1']['chunk_real_s'], s['reference']['chunk_real_s'], s['bm2s1']['own_clocks'].get('bsw_s', 0), s['reference']['own_clocks'].get('bsw_s', 0), s['bm2s1'].get('device_batches'), s['bm2s1'].get('seam_clock'), s['sam_equal']))"
done
echo "finished at $(( $(date +%s) - T0 ))s"
