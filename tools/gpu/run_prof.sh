#!/bin/bash
# One SHORT gpurun call (made for a round with ~10 GPU-minutes left): launch-policy sweep on the GRCh38-sized bench workload, then
# the rocprofv3 passes whose summaries go to profiles/ -- with the sweep's best settings exported, so the profiles describe the
# policy that becomes the default -- then the bench with its parity gate.  Steps are ordered by value; each checks the clock.
#   gpurun --timeout 600 -- 'bash tools/gpu/run_prof.sh <tag> <deadline_s>'
TAG=${1:-r02}; LIMIT=${2:-450}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
left() { echo $(( LIMIT - ($(date +%s) - T0) )); }
cd $R
export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)          # the first import on a fresh box pages the image in: do it beside the index build
timeout 420 python tools/gpu/sweep.py $O --steps 4 --budget-s ${SWEEP_S:-60} > $O/sweep.out 2> $O/sweep.err; echo "sweep rc=$? at $(( $(date +%s) - T0 ))s" >> $O/sweep.err
grep "\[sweep\]" $O/sweep.err | tail -60
[ -f $O/best_env.sh ] && . $O/best_env.sh
env | grep "^BM2_" > $O/env_used.txt
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e"
if [ $(left) -gt 60 ]; then
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 1 > $O/bench.json 2> $O/kt.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null 2>> $O/kt.err
  echo "kt done at $(( $(date +%s) - T0 ))s"; head -c 900 $O/bench.json; echo; head -14 $O/kernel_trace.md
fi
if [ $(left) -gt 50 ]; then
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_f.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null 2>> $O/pmc_f.err
  echo "fetch done at $(( $(date +%s) - T0 ))s"
fi
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $(left) -gt 50 ]; then      # SQ and TCC counters sit in different blocks: one pass for both; split if rocprofv3 refuses
  timeout 120 rocprofv3 --pmc WRITE_SIZE $SQ1 --kernel-trace -d /tmp/p_ws -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_ws.err
  DB=$(find /tmp/p_ws -name "*.db" | head -1)
  if [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/pmc_sq1.md > /dev/null 2>> $O/pmc_ws.err && grep -q SQ_INSTS_VALU $O/pmc_sq1.md && grep -q WRITE_SIZE $O/pmc_sq1.md; then
    cp $O/pmc_sq1.md $O/pmc_write.md; echo "write+sq1 (one pass) done at $(( $(date +%s) - T0 ))s"
  else
    rm -f $O/pmc_sq1.md
    timeout 120 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p_sq1 -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_sq1.err
    python $R/tools/rocpd_summary.py $(find /tmp/p_sq1 -name "*.db" | head -1) $O/pmc_sq1.md > /dev/null 2>> $O/pmc_sq1.err
    if [ $(left) -gt 50 ]; then
      timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_w.err
      python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null 2>> $O/pmc_w.err
    fi
    echo "sq1, write (separate passes) done at $(( $(date +%s) - T0 ))s"
  fi
fi
cd $R
if [ $(left) -gt 120 ]; then     # the bench as the driver runs it, minus the CPU baseline and the end-to-end leg: the parity gate on the new defaults
  timeout $(( $(left) - 10 )) python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_parity.json 2> $O/bench_parity.err; echo "bench+parity rc=$? at $(( $(date +%s) - T0 ))s" | tee -a $O/bench_parity.err
  grep "parity" $O/bench_parity.err | tail -4
fi
cd /tmp
if [ $(left) -gt 50 ]; then
  timeout 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/p_sq2 -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_sq2.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) $O/pmc_sq2.md > /dev/null 2>> $O/pmc_sq2.err
  echo "sq2 done at $(( $(date +%s) - T0 ))s"
fi
cd $R
if [ $(left) -gt 100 ]; then
  timeout $(( $(left) - 5 )) python -m pytest tests/test_pipeline_gpu.py tests/test_end_to_end_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
fi
echo "finished at $(( $(date +%s) - T0 ))s"
