#!/bin/bash
# Round 6, the closing call on the round's last code: the bench exactly as the driver runs it (is the last stdout line compact, strict JSON with roofline +
# cpu_baseline + parity?), the whole GPU suite, smoke().
#   gpurun --timeout 2400 -- 'bash tools/gpu/run_r06_final.sh r06z 2300'
TAG=${1:-r06z}; LIMIT=${2:-2300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench.err; at bench $?
cp /tmp/bm2_bench/bench_full_pe150.json $O/bench_full.json 2>/dev/null
cp /tmp/bm2_bench/bench_full_ont2d.json $O/bench_full_ont2d.json 2>/dev/null
cp /tmp/bm2_bench/bench_full_bsw.json $O/bench_full_bsw.json 2>/dev/null
python - <<PY
import json
lines = [l for l in open("$O/bench_stdout.txt").read().split("\n") if l.strip()]
print("stdout lines:", len(lines), "; last line bytes:", len(lines[-1].encode()) if lines else None)
d = json.loads(lines[-1])
print("parsed: value %.0f %s, ms_per_step %.2f, roofline.frac %.3f, cpu_baseline %.0f (%s cores), parity %s, value_end_to_end %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d.get("parity"), d.get("value_end_to_end")))
PY
grep "^\[bench\]" $O/bench.err | tail -9 | cut -c1-420
grep "bm2s1\] per call" $O/bench.err | tail -2
if [ $(left) -gt 420 ]; then
  timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
if [ $(left) -gt 120 ]; then
  timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; at smoke $?; tail -2 $O/smoke.log
fi
echo "finished at $(( $(date +%s) - T0 ))s"
