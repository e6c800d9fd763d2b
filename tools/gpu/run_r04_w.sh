#!/bin/bash
# Round 4, call W (the budget's last 100 seconds): eight k_chain_heavy tiers (BM2_CHAIN_FINE_TIERS=1, with the gate) against five.
TAG=${1:-r04w}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
show() { python - <<P
import json
d = json.load(open("$1"))
print("$2: value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
print("parity", {k: (d.get("parity") or {}).get(k) for k in ("regs_equal", "fin_equal", "sam_equal")})
P
}
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding"
BM2_CHAIN_FINE_TIERS=1 timeout 110 python bench.py $Q --parity-reads 51200 > $O/bench_fine.json 2> $O/bench_fine.err; show $O/bench_fine.json fine_tiers
timeout 40 python bench.py $Q --no-parity > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json five_tiers
