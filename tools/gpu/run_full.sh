#!/bin/bash
# One gpurun call of a round: GPU test suite, the default bench (parity gate, CPU baseline, end-to-end leg) on the GRCh38-sized
# workload, then -- on the index that run left in /tmp -- the rocprofv3 kernel trace and the PMC passes whose summaries go to profiles/.
#   gpurun --timeout 1500 -- 'bash tools/gpu/run_full.sh <tag> [steps]'
TAG=${1:-r02}; STEPS=${2:-10}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
fi
BM2_TAIL_PROF=1 timeout 700 python bench.py --steps $STEPS --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err; tail -12 $O/bench.err
head -c 1500 $O/bench.json; echo
if [ -z "$SKIP_ONT" ]; then     # config 5 shape on the same index: its own bench line (parity on a sample, CPU baseline on a sample)
  timeout 500 python bench.py --workload ont2d --steps 3 --warmup 1 --parity-reads 48 > $O/bench_ont2d.json 2> $O/bench_ont2d.err; echo "ont2d rc=$?" >> $O/bench_ont2d.err; tail -6 $O/bench_ont2d.err
  head -c 600 $O/bench_ont2d.json; echo
fi
if [ -n "$TRY_NSUB" ]; then     # experiment: the chunk as two sub-batches on two sets of streams
  for NS in 1 2; do BM2_N_SUB=$NS timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-e2e 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N_SUB=$NS', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"; done | tee $O/nsub.txt
fi
if [ -z "$SKIP_PROF" ]; then
  cd /tmp
  B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e"
  timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $B --steps 4 --warmup 1 > $O/bench_kt.json 2> $O/kt.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) $O/kernel_trace.md > /dev/null
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_f.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) $O/pmc_fetch.md > /dev/null
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_w.err
  python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_write.md > /dev/null
  rocprofv3 -L > $O/counters_avail.txt 2>&1
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/p_sq$i -o s -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_sq$i.err
    python $R/tools/rocpd_summary.py $(find /tmp/p_sq$i -name "*.db" | head -1) $O/pmc_sq$i.md > /dev/null
  done
  head -30 $O/kernel_trace.md
fi
