#!/bin/bash
# Round 5, call E: knob parity tests; the seeding knobs once more (hand-over ids drawn exactly, k_bwd_heavy's grid); config 5 against chunk size
# (tools/gpu/ont_scaling.py), with and without the cooperative chain filter; config 2 with the combining S1 binding (bwa-mem2.bm2s1 on 1 M SE reads).
TAG=${1:-r05e}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "off_by_default" > $O/pytest_knobs.log 2>&1; at pytest $?
tail -3 $O/pytest_knobs.log
timeout 300 python tools/gpu/sweep.py $O --steps 4 --budget-s 200 --only "seeding: k_bwd hands,seeding: workgroups per CU of k_bwd_heavy,kept-chain walk" > $O/sweep.log 2>&1; at sweep $?
grep "\[sweep\]" $O/sweep.log | tail -14 | cut -c1-420
timeout 300 python tools/gpu/ont_scaling.py $O/ont_scaling.json --sizes 10000,20000,40000 > $O/ont_scaling.log 2>&1; at ont_scaling $?
grep "ont_scaling" $O/ont_scaling.log | cut -c1-300
BM2_CHAIN_COOP_FLT=1 timeout 200 python tools/gpu/ont_scaling.py $O/ont_scaling_coop.json --sizes 10000 > $O/ont_scaling_coop.log 2>&1; at ont_coop $?
grep "ont_scaling" $O/ont_scaling_coop.log | cut -c1-300
timeout 400 python bench.py --workload bsw --steps 3 --warmup 1 > $O/bench_bsw.json 2> $O/bench_bsw.err; at bsw $?
grep -E "S1 binding|bm2s1|cpu baseline|parity" $O/bench_bsw.err | tail -8 | cut -c1-300
python - <<P
import json
try:
    d = json.load(open("$O/bench_bsw.json"))
    print("config2: %.0f G cells/s" % d["extend_kernel"]["gcups"], json.dumps(d.get("s1_binding"))[:1500])
except Exception as e:
    print("no bsw line:", e)
P
