#!/bin/bash
# Round 4, call O: the lane kernel's target bases in 28-row windows of the packed reference + query staging with four loads in flight,
# against the build before (BM2_LIB), and two SQ counter passes that say what the extension kernels wait for.
TAG=${1:-r04o}; LIMIT=${2:-700}
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
if [ -f $R/bwa-mem2_amd/libbm2_base.so ]; then
  BM2_LIB=$R/bwa-mem2_amd/libbm2_base.so timeout 300 python bench.py $Q --no-parity > $O/bench_base.json 2> $O/bench_base.err; show $O/bench_base.json base
fi
timeout 100 python bench.py --workload bsw --steps 5 --warmup 2 --no-binding-s1 > $O/bench_bsw.json 2> $O/bench_bsw.err
python -c "
import json; d=json.load(open('$O/bench_bsw.json')); print('bsw', d['extend_kernel'], d['parity']['pairs_equal'])"
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --no-binding"
SQA="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY"
SQB="SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES"
timeout 120 rocprofv3 --pmc $SQA --kernel-trace -d /tmp/p_a -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_a.err; echo "sqa rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/p_a -name "*.db" | head -1) $O/pmc_sq_a.md > /dev/null 2>> $O/pmc_a.err
timeout 120 rocprofv3 --pmc $SQB --kernel-trace -d /tmp/p_b -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_b.err; echo "sqb rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/p_b -name "*.db" | head -1) $O/pmc_sq_b.md > /dev/null 2>> $O/pmc_b.err
grep -n "k_ext_seeds\|k_ext_wave" $O/pmc_sq_a.md $O/pmc_sq_b.md | grep "SQ_" | head -40
