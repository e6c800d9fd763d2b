#!/bin/bash
# Round 3, call Q (re-run as R with the workgroup-per-read SMEM sort): long-read GPU tests, config 5 at 2000 / 10 000 reads per step, pe150 check
TAG=${1:-r03q}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 300 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 400 python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 200 --no-cpu-baseline > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$? at $(( $(date +%s) - T0 ))s"
grep "parity" $O/bench_ont2d.err | tail -3
python -c "import json; d=json.load(open('$O/bench_ont2d.json')); print('ont2d 2000: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
for n in 10000; do
  for w in 32; do
    BM2_CHAIN_OVF_WAVES_PER_CU=$w timeout 300 python bench.py --workload ont2d --reads $n --steps 2 --warmup 1 --no-parity --no-cpu-baseline > $O/bench_ont2d_${n}_w$w.json 2> $O/bench_ont2d_${n}_w$w.err
    python -c "import json; d=json.load(open('$O/bench_ont2d_${n}_w$w.json')); print('ont2d $n reads/step, $w waves per CU: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
  done
done
timeout 150 python bench.py --steps 6 --warmup 2 --no-parity --no-cpu-baseline --no-e2e 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pe150: %.2f M reads/s' % (d['value']/1e6), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
