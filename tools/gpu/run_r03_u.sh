#!/bin/bash
# Round 3, call U: B-tree nodes as register copies (chaining), 32-bit quad sums in backwardExt (seeding): GPU tests, pe150 with the gate, chain knobs, config 5
TAG=${1:-r03u}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 400 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "pytest done at $(( $(date +%s) - T0 ))s"
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-binding --parity-reads 20480 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? at $(( $(date +%s) - T0 ))s"
grep "parity" $O/bench.err | tail -4
python -c "import json; d=json.load(open('$O/bench.json')); print('pe150: %.2f M reads/s' % (d['value']/1e6), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, d.get('roofline'))"
timeout 260 python tools/gpu/sweep.py $O --steps 4 --only "chain heavy,chain waves,rows in registers" --budget-s 150 2>&1 | grep "\[sweep\]" | tail -20
echo "sweep done at $(( $(date +%s) - T0 ))s"
timeout 400 python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 200 --no-cpu-baseline > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$? at $(( $(date +%s) - T0 ))s"
grep "parity" $O/bench_ont2d.err | tail -3
python -c "import json; d=json.load(open('$O/bench_ont2d.json')); print('ont2d: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
for w in 5 6; do
  BM2_CHAIN_OVF_WAVES_PER_EU=$w timeout 300 python bench.py --workload ont2d --steps 2 --warmup 1 --no-parity --no-cpu-baseline > $O/bench_ont2d_wpe$w.json 2> $O/bench_ont2d_wpe$w.err
  python -c "import json; d=json.load(open('$O/bench_ont2d_wpe$w.json')); print('ont2d, overflow chaining at $w waves per SIMD: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
done
echo "finished at $(( $(date +%s) - T0 ))s"
