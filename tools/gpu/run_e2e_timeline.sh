#!/bin/bash
# Per-dispatch timeline of the end-to-end leg (FASTQ -> SAM pipeline of bench.py): which hardware queue every launch of the device workers
# and of the tail workers sat on, and what waited behind what (notes/NEXT.md, "First experiment for the end-to-end leg").
#   gpurun --timeout 500 -- 'bash tools/gpu/run_e2e_timeline.sh <tag> [GPU_MAX_HW_QUEUES]'
TAG=${1:-e2e_tl}; Q=${2:-8}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=$Q
T0=$(date +%s)
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_e2e -o e2e -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-binding > $O/bench.json 2> $O/bench.err; echo "rc=$? at $(( $(date +%s) - T0 ))s"
python $R/tools/rocpd_timeline.py $(find /tmp/p_e2e -name "*.db" | head -1) $O/timeline.tsv 2>> $O/bench.err
python - <<P
import json
d = json.load(open("$O/bench.json")); e = d.get("end_to_end") or {}
print("GPU_MAX_HW_QUEUES=$Q: hot path %.2f M reads/s, end_to_end %.2f M reads/s (%.2f)" % (d["value"] / 1e6, e.get("value", 0) / 1e6, e.get("frac_of_hot_path", 0)), {k: round(v, 1) for k, v in (e.get("stage_ms_per_chunk") or {}).items()})
P
wc -l $O/timeline.tsv; gzip -f $O/timeline.tsv
echo "finished at $(( $(date +%s) - T0 ))s"
