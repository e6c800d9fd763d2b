#!/bin/bash
# Round 3, call P: config 5 with a chunk of 10 000 reads (100 Mbases, the chunk of `bwa-mem2 mem -K 100000000`) instead of 2000
TAG=${1:-r03p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
for n in 10000 5000; do
  timeout 400 python bench.py --workload ont2d --reads $n --steps 2 --warmup 1 --no-parity --no-cpu-baseline > $O/bench_ont2d_$n.json 2> $O/bench_ont2d_$n.err; echo "ont2d $n rc=$? at $(( $(date +%s) - T0 ))s"
  python -c "import json; d=json.load(open('$O/bench_ont2d_$n.json')); print('ont2d $n reads/step: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})" || tail -5 $O/bench_ont2d_$n.err
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5
