#!/bin/bash
# Round 4, call P: the main stream's waits queued after the last launch of a phase (extension, S1, chaining; k_chain on a side stream),
# against the build before (BM2_LIB); a kernel trace with the per-dispatch timeline of the new build; SQ counters of steady-state steps.
TAG=${1:-r04p}; LIMIT=${2:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
show() { python - <<P
import json
d = json.load(open("$1"))
print("$2: value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
print("parity", {k: (d.get("parity") or {}).get(k) for k in ("regs_equal", "fin_equal", "sam_equal")})
P
}
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding"
timeout 300 python bench.py $Q --parity-reads 51200 > $O/bench_new.json 2> $O/bench_new.err; show $O/bench_new.json new
BM2_CHAIN_MAIN_SIDE=0 timeout 300 python bench.py $Q --no-parity > $O/bench_chain_main.json 2> $O/bench_chain_main.err; show $O/bench_chain_main.json chain_on_main
if [ -f $R/bwa-mem2_amd/libbm2_base.so ]; then
  BM2_LIB=$R/bwa-mem2_amd/libbm2_base.so timeout 300 python bench.py $Q --no-parity > $O/bench_base.json 2> $O/bench_base.err; show $O/bench_base.json base
fi
timeout 100 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw.json 2> $O/bench_bsw.err
python -c "
import json; d=json.load(open('$O/bench_bsw.json')); print('bsw', d['extend_kernel'], d['parity']['pairs_equal'])"
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --no-binding"
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 4 > $O/bench_kt.json 2> $O/kt.err; echo "kt rc=$?"
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/kernel_trace.md > /dev/null 2>> $O/kt.err
python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1; tail -240 $O/timeline_all.tsv > $O/timeline.tsv; rm -f $O/timeline_all.tsv
SQA="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"
timeout 150 rocprofv3 --pmc $SQA --kernel-trace -d /tmp/p_a -o s -- $B --steps 2 --warmup 2 > /dev/null 2> $O/pmc_a.err; echo "sqa rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/p_a -name "*.db" | head -1) $O/pmc_sq_steady.md > /dev/null 2>> $O/pmc_a.err
grep -n "second half" -A 400 $O/pmc_sq_steady.md | grep "k_ext_seeds\|k_ext_wave\|k_bwd<\|k_chain" | head -60
