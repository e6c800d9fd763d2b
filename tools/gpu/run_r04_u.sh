#!/bin/bash
# Round 4, call U: __graft_entry__.smoke() on the final code; the thirty-chunk end-to-end leg with two tail workers instead of three.
TAG=${1:-r04u}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
BM2_E2E_TAILS=2 timeout 200 python bench.py --steps 6 --warmup 3 --no-side-workloads --no-binding --no-cpu-baseline --no-parity > $O/bench_tails2.json 2> $O/bench_tails2.err; echo "bench rc=$?"
python - <<P
import json
d = json.load(open("$O/bench_tails2.json"))
e = d.get("end_to_end") or {}
print("tails2: hot %.1f ms | e2e %s" % (d["ms_per_step"], json.dumps({k: e.get(k) for k in ("value", "frac_of_hot_path", "chunks", "steady_state", "ms_per_chunk", "host_cpu_s_per_chunk", "stage_ms_per_chunk", "error")})))
P
