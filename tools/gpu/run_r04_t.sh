#!/bin/bash
# Round 4, call T: the end-to-end leg three times round its ten distinct chunks (30 M reads), with the steady-state rate beside the whole-leg rate.
TAG=${1:-r04t}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 330 python bench.py --no-side-workloads --no-binding --no-cpu-baseline --parity-reads 51200 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -5 $O/bench.err
python - <<P
import json
d = json.load(open("$O/bench.json"))
e = d.get("end_to_end") or {}
print("hot %.1f ms %.2f M | e2e %s" % (d["ms_per_step"], d["value"] / 1e6, json.dumps({k: e.get(k) for k in ("value", "frac_of_hot_path", "chunks", "distinct_chunks", "steady_state", "wall_s", "ms_per_chunk", "host_cpu_s_per_chunk", "stage_ms_per_chunk", "chunk_check", "error")})))
print("parity", {k: (d.get("parity") or {}).get(k) for k in ("regs_equal", "fin_equal", "sam_equal")})
P
