#!/bin/bash
# Round 3, call H: GPU tests of the tail kernels (DPP row maximum, CIGAR shapes), the bench with the new end-to-end defaults, the end-to-end leg
# under a kernel trace, SQ counters of the tail kernels.
TAG=${1:-r03h}; LIMIT=${2:-560}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s); left() { echo $(( LIMIT - ($(date +%s) - T0) )); }; at() { echo "$1 rc=$2 at $(( $(date +%s) - T0 ))s"; }
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 200 python -m pytest tests/test_zz_tail_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python bench.py --steps 10 --warmup 2 --parity-reads 20480 > $O/bench_full.json 2> $O/bench_full.err; at bench $?
grep -E "parity|index built" $O/bench_full.err | tail -6
python - <<P
import json
try:
    d = json.load(open("$O/bench_full.json"))
    print("value %.2f M reads/s, %.1f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()})
    e = d.get("end_to_end") or {}
    print("end_to_end %.2f M reads/s" % (e.get("value", 0) / 1e6), {k: v for k, v in e.items() if k not in ("scope",)})
except Exception as e:
    print("no bench line:", e)
P
cd /tmp
if [ $(left) -gt 100 ]; then
  PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924 PROBE_LIMIT_S=80 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/p_e2e -o kt -- python $R/tools/gpu/tail_probe.py $O 3100 10 500000 > $O/probe.out 2> $O/probe.err
  at probe $?; grep "\[probe\]" $O/probe.err | tail -3
  python $R/tools/rocpd_summary.py $(find /tmp/p_e2e -name "*.db" | head -1) $O/e2e_kernel_trace.md > /dev/null 2>> $O/probe.err; head -22 $O/e2e_kernel_trace.md
fi
if [ $(left) -gt 60 ]; then
  SCALING_TAG=pmc BM2_TAIL_PROF=1 SCALING_THREADS="12" timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d /tmp/p_pmc -o s -- python $R/tools/gpu/tail_scaling.py $O 128 500000 > $O/pmc.out 2> $O/pmc.err
  at pmc $?
  python $R/tools/rocpd_summary.py $(find /tmp/p_pmc -name "*.db" | head -1) $O/tail_pmc_sq.md > /dev/null 2>> $O/pmc.err
  grep -E "k_ksw_align2|k_gen_cigar|k_cigar_flat" $O/tail_pmc_sq.md | head -40
fi
echo "finished at $(( $(date +%s) - T0 ))s"
