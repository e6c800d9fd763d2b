#!/bin/bash
# Round 3, call W (last): the whole GPU suite, the bench exactly as the driver runs it, the kernel trace and one SQ counter pass of the same code -> profiles/r03w_*
TAG=${1:-r03w}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
T0=$(date +%s); at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
timeout 500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; at pytest $?; tail -2 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "\[bench\]" $O/bench_full.err | tail -12
python - <<P
import json
d = json.load(open("$O/bench_full.json"))
print("value %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()})
print("roofline frac %.3f; k_bwd %.2f ms" % (d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]))
e = d.get("end_to_end") or {}
print("end_to_end %.2f M reads/s (%.2f)" % (e.get("value", 0) / 1e6, e.get("frac_of_hot_path", 0)), e.get("chunk_check"))
print("binding", {k: v for k, v in (d.get("binding") or {}).items() if k != "scope"})
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
P
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-binding"
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 1 > $O/bench.json 2> $O/kt.err; at trace $?
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null 2>> $O/kt.err
head -14 $O/kernel_trace.md
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_f.err; at fetch $?
python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
timeout 150 rocprofv3 --pmc WRITE_SIZE $SQ1 --kernel-trace -d /tmp/p_ws -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_ws.err; at write_sq1 $?
python $R/tools/rocpd_summary.py $(find /tmp/p_ws -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_ws.err; cp $O/pmc_sq1.md $O/pmc_write.md
grep -E "k_bwd|k_ext_lanes" $O/pmc_sq1.md | head -24
echo "finished at $(( $(date +%s) - T0 ))s"
