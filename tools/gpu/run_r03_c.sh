#!/bin/bash
# Round 3, call C: is the box's CPU time capped (cgroup quota)?  Tail scaling at small thread counts, with and without the pools' spin.
TAG=${1:-r03c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; export TMPDIR=/tmp
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>&1)"; cat /sys/fs/cgroup/cpu.stat 2>&1; grep -i "cpus_allowed_list" /proc/self/status; cat /proc/pressure/cpu 2>&1; } | tee $O/cgroup_before.txt
for spin in 1000 20; do
  SCALING_TAG=spin$spin BM2_POOL_SPIN_US=$spin BM2_TAIL_PROF=1 SCALING_THREADS="16 24 32 48 64 96" timeout 200 python tools/gpu/tail_scaling.py $O 128 500000 > $O/scaling_spin$spin.out 2> $O/scaling_spin$spin.err
  echo "spin=$spin rc=$?"; grep "\[scaling\] [0-9t]" $O/scaling_spin$spin.err
  cat /sys/fs/cgroup/cpu.stat 2>&1 | tee $O/cgroup_after_spin$spin.txt
done
# a plain CPU burner: 256 busy threads for 2 s -- how much CPU time does the box really give?
python - <<'P' 2>&1 | tee $O/burn.txt
import os, time, threading, ctypes
import numpy as np
def burn(sec, out, i):
    t0 = time.perf_counter(); n = 0
    a = np.ones(1 << 16)
    while time.perf_counter() - t0 < sec:
        a.sum(); n += 1
    out[i] = n
for nt in (16, 64, 128, 256):
    out = [0] * nt
    c0 = time.process_time(); w0 = time.perf_counter()
    th = [threading.Thread(target=burn, args=(1.5, out, i)) for i in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    print("threads %3d: cpu %.1f s in wall %.2f s -> %.1f CPUs busy; work %d" % (nt, time.process_time() - c0, time.perf_counter() - w0, (time.process_time() - c0) / (time.perf_counter() - w0), sum(out)))
P
cat /sys/fs/cgroup/cpu.stat 2>&1 | tail -4
