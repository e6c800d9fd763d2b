#!/bin/bash
# Round 3, call AA: configs 5 and 2 at the round's last code (the wavefront kernel's `rev` argument now carries a priority field)
TAG=${1:-r03aa}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
timeout 400 python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 200 --no-cpu-baseline > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$? at $(( $(date +%s) - T0 ))s"
grep "parity" $O/bench_ont2d.err | tail -3
python -c "import json; d=json.load(open('$O/bench_ont2d.json')); print('ont2d: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
timeout 200 python bench.py --workload bsw --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bsw.json 2> $O/bench_bsw.err; echo "bsw rc=$? at $(( $(date +%s) - T0 ))s"
python -c "import json; d=json.load(open('$O/bench_bsw.json')); print('bsw:', d['value'], d['unit'], d.get('ms_per_step'))"
echo "finished at $(( $(date +%s) - T0 ))s"
