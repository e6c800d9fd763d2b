#!/bin/bash
# Round 4, call N: the reference string four bases per byte on the device (refseq.h) + k_advance over the list of kept regs:
# all GPU tests, the hot path packed vs BM2_REF_BYTES=1 (same build, same chunks), config 5 for the kept list.
TAG=${1:-r04n}; LIMIT=${2:-900}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
(python -c "import torch" > /dev/null 2>&1 &)
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
show() { python - <<P
import json
d = json.load(open("$1"))
print("$2: value %.2f M reads/s, %.1f ms/step, stages %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}))
print("parity", {k: d["parity"].get(k) for k in ("regs_equal", "fin_equal", "sam_equal")}, "replica GB", d.get("index_replica_gb"))
P
}
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding --parity-reads 51200 > $O/bench_packed.json 2> $O/bench_packed.err
show $O/bench_packed.json packed
BM2_REF_BYTES=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-side-workloads --no-binding --no-parity > $O/bench_bytes.json 2> $O/bench_bytes.err
show $O/bench_bytes.json bytes
timeout 300 python bench.py --workload ont2d --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_ont2d.json 2> $O/bench_ont2d.err
show $O/bench_ont2d.json ont2d
