#!/usr/bin/env python3
"""End-to-end leg (FASTQ text -> SAM text) of bench.py on a SMALL genome, without torch: where the time of the host tail and of its two
device batches (mate-rescue SW, CIGAR) goes.  Run it under `rocprofv3 --kernel-trace` with BM2_TAIL_PROF=1 for the per-kernel and
per-phase clocks.   python tools/gpu/tail_probe.py <out_dir> [genome_mbp] [chunks]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))


def main():
    out = sys.argv[1]
    mbp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    os.makedirs(out, exist_ok=True)
    import bench
    import bm2
    wd = os.environ.get("PROBE_WORKDIR", "/tmp/bm2_tail_probe")          # PROBE_WORKDIR=/tmp/bm2_bench PROBE_SEED=20260924: bench.py's own index
    os.makedirs(wd, exist_ok=True)
    t = time.time()
    prefix, contigs = bench.prepare_genome(wd, mbp, int(os.environ.get("PROBE_SEED", 777)))
    procs = []
    for i in range(n_chunks + 1):
        fa, fb = os.path.join(wd, "c%d_1.fq" % i), os.path.join(wd, "c%d_2.fq" % i)
        procs.append((fa, fb, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gen_chunk.py"), prefix + ".contigs.npz", str(900 + i),
                                                str(int(sys.argv[4]) if len(sys.argv) > 4 else 500000), "150", fa, fb, "c%d_" % i])))
    texts = []
    for fa, fb, pr in procs:
        assert pr.wait() == 0
        texts.append((open(fa, "rb").read(), open(fb, "rb").read()))
    print("[probe] inputs ready in %.1fs" % (time.time() - t), file=sys.stderr, flush=True)
    ctx = bm2.Context(0, prefix)
    opt = bm2.default_opt()
    r = bench.end_to_end(ctx, bm2, texts[1:], opt, True, 0, limit_s=float(os.environ.get("PROBE_LIMIT_S", 150)))     # (its warm-up chunks pass through the same threads before the clock starts)
    print("[probe] defaults: %.2f M reads/s %s" % (r["value"] / 1e6, {k: round(v) for k, v in r["stage_ms_per_chunk"].items()}), file=sys.stderr, flush=True)
    r["genome_mbp"] = mbp
    r["splits"] = []                                        # tail workers x host threads per worker: which split feeds the device best
    for tails, threads in [(s_.split("x")) for s_ in os.environ.get("PROBE_SPLITS", "").split()]:
        os.environ["BM2_E2E_TAILS"] = tails; os.environ["BM2_E2E_TAIL_THREADS"] = threads
        os.environ.pop("BM2_TAIL_PROF", None)
        q = bench.end_to_end(ctx, bm2, texts[1:], opt, True, 0)
        r["splits"].append({"tail_workers": int(tails), "threads_per_worker": int(threads), "value": q["value"], "stage_ms_per_chunk": q["stage_ms_per_chunk"]})
        print("[probe] %s workers x %s threads: %.2f M reads/s %s" % (tails, threads, q["value"] / 1e6, {k: round(v) for k, v in q["stage_ms_per_chunk"].items()}), file=sys.stderr, flush=True)
    r["variants"] = []                                      # PROBE_ENVS="BM2_E2E_DEVS=1 BM2_TAIL_PIN=0,BM2_E2E_DEVS=1 ...": the leg again under each setting
    for spec in os.environ.get("PROBE_ENVS", "").split():
        kv = dict(x.split("=", 1) for x in spec.split(","))
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        os.environ.pop("BM2_TAIL_PROF", None); os.environ.pop("BM2_E2E_TAILS", None); os.environ.pop("BM2_E2E_TAIL_THREADS", None)
        os.environ.update(kv)
        try:
            q = bench.end_to_end(ctx, bm2, texts[1:], opt, True, 0, limit_s=120)
            r["variants"].append({"env": kv, "value": q["value"], "stage_ms_per_chunk": q["stage_ms_per_chunk"]})
            print("[probe] %s: %.2f M reads/s %s" % (spec, q["value"] / 1e6, {k: round(v) for k, v in q["stage_ms_per_chunk"].items()}), file=sys.stderr, flush=True)
        except Exception as e:                                # noqa
            print("[probe] %s: FAILED %s" % (spec, e), file=sys.stderr, flush=True)
            break
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("[probe] rescue stats (planned, used, missed):", bm2.sam_rescue_stats(), "cigar stats:", bm2.sam_cigar_stats(), file=sys.stderr, flush=True)
    json.dump(r, open(os.path.join(out, "tail_probe.json"), "w"), indent=1)
    print(json.dumps(r), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
