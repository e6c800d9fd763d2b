#!/bin/bash
# Round 4, call S: the end-to-end leg under other worker counts / queue counts / without the tail contexts' stream priority; hot path at 12 and 20 queues.
TAG=${1:-r04s}; LIMIT=${2:-420}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
show() { python - <<P
import json
try:
    d = json.load(open("$1"))
    e = d.get("end_to_end") or {}
    print("$2: hot %.1f ms | e2e %.2f M reads/s, %.1f ms/chunk, cpu %.2f s/chunk, stages %s" % (d["ms_per_step"], (e.get("value") or 0) / 1e6, e.get("ms_per_chunk") or 0, e.get("host_cpu_s_per_chunk") or 0,
          {k: round(v, 1) for k, v in (e.get("stage_ms_per_chunk") or {}).items()}))
except Exception as ex:
    print("$2: no line", ex)
P
}
Q="--steps 6 --warmup 3 --no-cpu-baseline --no-side-workloads --no-binding --no-parity"
timeout 200 python bench.py $Q > $O/e_default.json 2> $O/e_default.err; show $O/e_default.json default
BM2_E2E_DEVS=1 timeout 120 python bench.py $Q > $O/e_dev1.json 2> $O/e_dev1.err; show $O/e_dev1.json devs1
BM2_E2E_TAILS=2 timeout 120 python bench.py $Q > $O/e_tail2.json 2> $O/e_tail2.err; show $O/e_tail2.json tails2
BM2_E2E_DEVS=1 BM2_E2E_TAILS=2 timeout 120 python bench.py $Q > $O/e_dev1_tail2.json 2> $O/e_dev1_tail2.err; show $O/e_dev1_tail2.json devs1_tails2
BM2_E2E_TAIL_PRIO=0 timeout 120 python bench.py $Q > $O/e_noprio.json 2> $O/e_noprio.err; show $O/e_noprio.json no_tail_prio
GPU_MAX_HW_QUEUES=8 timeout 120 python bench.py $Q > $O/e_q8.json 2> $O/e_q8.err; show $O/e_q8.json queues8
GPU_MAX_HW_QUEUES=12 timeout 120 python bench.py $Q > $O/e_q12.json 2> $O/e_q12.err; show $O/e_q12.json queues12
GPU_MAX_HW_QUEUES=20 timeout 120 python bench.py $Q > $O/e_q20.json 2> $O/e_q20.err; show $O/e_q20.json queues20
