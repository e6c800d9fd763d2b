#!/bin/bash
# Round 6: is the chaining stage bimodal step by step?  80 steps of the hot path under the kernel trace; per step: k_chain's and the tiers' start / end.
#   gpurun --timeout 600 -- 'bash tools/gpu/run_r06_aw.sh r06aw'
TAG=${1:-r06aw}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-parity --no-e2e --no-side-workloads --no-binding"
timeout 200 $B --steps 2 --warmup 1 > /dev/null 2> $O/prep.err
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_kt -o kt -- $B --steps 80 --warmup 4 --full-json $O/bench_kt.json > /dev/null 2> $O/kt.err; echo "kt rc=$?"
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB $O/timeline_all.tsv >> $O/kt.err 2>&1
python3 - <<PY
rows = [l.rstrip("\n").split("\t") for l in open("$O/timeline_all.tsv")][1:]
ev = [(float(r[0]), float(r[1]), r[3], r[10]) for r in rows]
steps, cur = [], None
for a, b, q, n in ev:
    if "k_walk<1>" in n:
        cur = {"t0": a, "kc": None, "tiers": [], "isl": None, "fin": None}; steps.append(cur)
    elif cur is not None:
        if n.startswith("k_chain(") or n == "k_chain": cur["kc"] = (a - cur["t0"], b - cur["t0"], q)
        elif "k_chain_heavy" in n: cur["tiers"].append((a - cur["t0"], b - cur["t0"], q))
        elif "k_chain_islands" in n: cur["isl"] = (a - cur["t0"], b - cur["t0"], q)
        elif "k_chain_finish_wave" in n: cur["fin"] = b - cur["t0"]
out = open("$O/chain_stage_per_step.tsv", "w")
out.write("step\tk_chain_start\tk_chain_end\tk_chain_q\ttiers_first_start\ttiers_last_end\tislands_start\tislands_end\tstage_end\n")
for i, s in enumerate(steps):
    if not s["kc"] or not s["tiers"]: continue
    out.write("%d\t%.2f\t%.2f\t%s\t%.2f\t%.2f\t%s\t%s\t%s\n" % (i, s["kc"][0], s["kc"][1], s["kc"][2], min(t[0] for t in s["tiers"]), max(t[1] for t in s["tiers"]),
              "%.2f" % s["isl"][0] if s["isl"] else "", "%.2f" % s["isl"][1] if s["isl"] else "", "%.2f" % s["fin"] if s["fin"] else ""))
out.close()
import statistics
d = [s["kc"][1] - s["kc"][0] for s in steps if s["kc"]]
print("steps", len(d), "k_chain ms: min %.2f median %.2f max %.2f" % (min(d), statistics.median(d), max(d)))
e = [s["fin"] - min(t[0] for t in s["tiers"]) for s in steps if s["kc"] and s["tiers"] and s["fin"]]
print("stage (first tier start -> finish_wave end) ms: min %.2f median %.2f max %.2f; >8.8: %d of %d" % (min(e), statistics.median(e), max(e), sum(1 for x in e if x > 8.8), len(e)))
PY
rm -f $O/timeline_all.tsv
head -12 $O/chain_stage_per_step.tsv
