#!/bin/bash
# Round 3, call O: configs 5 and 2 at HEAD (staged heavy chaining is the default now), GPU suite at HEAD
TAG=${1:-r03o}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s)
(python -c "import torch" > /dev/null 2>&1 &)
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? at $(( $(date +%s) - T0 ))s"; tail -3 $O/pytest.log
timeout 400 python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 200 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$? at $(( $(date +%s) - T0 ))s"
grep "\[bench\]" $O/bench_ont2d.err | tail -5
python -c "
import json; d=json.load(open('$O/bench_ont2d.json')); print('ont2d: %.0f reads/s, %.0f ms/step' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, d['parity'].get('regs_equal'), d['parity'].get('fin_equal'), d['parity'].get('sam_equal'), (d.get('cpu_baseline') or {}).get('value'))"
for st in 0; do
  BM2_CHAIN_STAGE=$st timeout 200 python bench.py --workload ont2d --steps 3 --warmup 1 --no-parity --no-cpu-baseline > $O/bench_ont2d_stage$st.json 2> $O/bench_ont2d_stage$st.err
  python -c "import json; d=json.load(open('$O/bench_ont2d_stage$st.json')); print('ont2d CHAIN_STAGE=$st: %.0f reads/s' % d['value'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
done
timeout 100 python bench.py --workload bsw --steps 5 --warmup 2 > $O/bench_bsw.json 2> $O/bench_bsw.err; echo "bsw rc=$? at $(( $(date +%s) - T0 ))s"
python -c "import json; d=json.load(open('$O/bench_bsw.json')); print('bsw: %.2f M reads/s-equivalent, %.1f G cells/s' % (d['value']/1e6, d['extend_kernel']['gcups']), d['parity'])"
