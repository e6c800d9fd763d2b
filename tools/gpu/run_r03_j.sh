#!/bin/bash
# Round 3, call J: whole GPU suite at HEAD (saturating subtractions in the SW kernels), the integer-VALU micro-benchmark, the bench with the
# drop-in timing leg, end-to-end variants (sub-batches of the hot path).
TAG=${1:-r03j}; LIMIT=${2:-700}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 60 tools/ubench/valu_int 20000 > $O/valu_int.txt 2>&1; at valu_int $?; cat $O/valu_int.txt
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; at pytest $?; tail -4 $O/pytest.log
timeout 500 python bench.py --steps 10 --warmup 2 --parity-reads 20480 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "binding|parity|index built" $O/bench_full.err | tail -8
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()})
    e = d.get("end_to_end") or {}
    print("end_to_end %.2f M reads/s (%.2f of the hot path)" % (e.get("value", 0) / 1e6, e.get("frac_of_hot_path", 0)), e.get("stage_ms_per_chunk"), e.get("chunk_check"))
    print("binding", d.get("binding"))
except Exception as e:
    print("no bench line:", e)
P
if [ $(left) -gt 80 ]; then
  PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=60 PROBE_ENVS="BM2_N_SUB=2 BM2_E2E_DEVS=3" \
  timeout $(( $(left) - 10 )) python tools/gpu/tail_probe.py $O 3100 10 500000 > $O/probe.out 2> $O/probe.err
  at probe $?; grep "\[probe\]" $O/probe.err | tail -5
fi
echo "finished at $(( $(date +%s) - T0 ))s"
