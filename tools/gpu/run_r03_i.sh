#!/bin/bash
# Round 3, call I: end-to-end leg with dynamic tail threads + high-priority tail streams (A/B by environment), k_bwd one-request variant (A/B by library)
TAG=${1:-r03i}; LIMIT=${2:-560}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 400 python bench.py --steps 10 --warmup 2 --parity-reads 20480 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; at bench $?
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, {k: round(v, 2) for k, v in d["roofline"]["seeding_stage"]["kernel_ms"].items()})
    e = d.get("end_to_end") or {}
    print("end_to_end %.2f M reads/s" % (e.get("value", 0) / 1e6), {k: v for k, v in e.items() if k not in ("scope",)})
except Exception as e:
    print("no bench line:", e)
P
BM2_LIB=$R/bwa-mem2_amd/libbm2_sb.so timeout 100 python bench.py --steps 10 --warmup 2 --no-parity --no-cpu-baseline --no-e2e > $O/bench_sb.json 2> $O/bench_sb.err; at bench_sb $?
python -c "import json; d=json.load(open('$O/bench_sb.json')); print('same-block variant: value %.2f M, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: round(v,2) for k,v in d['roofline']['seeding_stage']['kernel_ms'].items()})"
if [ $(left) -gt 100 ]; then
  PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=60 \
  PROBE_ENVS="BM2_E2E_TAIL_PRIO=0 BM2_E2E_DYN_THREADS=0 BM2_E2E_TAIL_PRIO=0,BM2_E2E_DYN_THREADS=0 BM2_E2E_TAILS=2,BM2_E2E_TAIL_THREADS=7 BM2_E2E_TAILS=4,BM2_E2E_TAIL_THREADS=4" \
  timeout $(( $(left) - 10 )) python tools/gpu/tail_probe.py $O 3100 10 500000 > $O/probe.out 2> $O/probe.err
  at probe $?; grep "\[probe\]" $O/probe.err | tail -8
fi
echo "finished at $(( $(date +%s) - T0 ))s"
