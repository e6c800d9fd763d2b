#!/bin/bash
# Round 4, the round's last code on the GPU: the whole GPU suite, the bench as the driver runs it (configs 5 and 2 and the 10 M-read CLI leg inside),
# then the rocprofv3 passes whose summaries go to profiles/ (kernel trace, FETCH / WRITE, SQ of the 150 bp workload; SQ of config 5).
#   gpurun --timeout 1500 -- 'bash tools/gpu/run_r04_final.sh r04z 1450'
TAG=${1:-r04z}; LIMIT=${2:-1450}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; at bench $?
grep -E "parity|end-to-end|cpu baseline|index built|binding|S1" $O/bench.err | tail -20
python - <<P
import json
try:
    d = json.load(open("$O/bench.json"))
    print("value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}))
    print("roofline frac %.3f" % d["roofline"]["frac"], "e2e %.2f M (%.2f)" % (d["end_to_end"]["value"] / 1e6, d["end_to_end"]["frac_of_hot_path"]))
    print("parity", json.dumps(d.get("parity"))[:300])
    for k in ("config5", "config2"):
        c = d.get(k) or {}
        print(k, c.get("value"), c.get("stage_ms_per_step"), json.dumps(c.get("parity"))[:200], json.dumps(c.get("cpu_baseline"))[:200])
    print("binding", json.dumps(d.get("binding"))[:900])
except Exception as e:
    print("no bench line:", e)
P
if [ $(left) -gt 500 ]; then
  timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $(left) -gt 120 ]; then
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 4 > $O/bench_kt.json 2> $O/kt.err; at kt $?
  DB=$(find /tmp/p_kt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1; tail -240 $O/timeline_all.tsv > $O/timeline.tsv; rm -f $O/timeline_all.tsv
fi
if [ $(left) -gt 100 ]; then
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_f.err; at fetch $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
fi
if [ $(left) -gt 100 ]; then
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_w.err; at write $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null 2>> $O/pmc_w.err
fi
if [ $(left) -gt 100 ]; then
  timeout 120 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_sq1.err; at sq1 $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_sq1.err
fi
if [ $(left) -gt 100 ]; then
  timeout 150 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_ont -o s -- python $R/bench.py --workload ont2d --no-cpu-baseline --no-parity --steps 2 --warmup 2 > $O/bench_ont2d_pmc.json 2> $O/pmc_ont.err; at ont_sq $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_ont -name "*.db" | head -1) $O/pmc_sq1_ont2d.md > /dev/null 2>> $O/pmc_ont.err
fi
echo "finished at $(( $(date +%s) - T0 ))s"
