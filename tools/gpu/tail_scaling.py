#!/usr/bin/env python3
"""How the host tail of ONE chunk scales with its thread count on the GPU box, with nothing else running (no pipeline): the device produces
the hits of a 1 M-read chunk once, then bm2_sam_pe_dev runs on it with 8 .. 216 threads (BM2_TAIL_PROF=1 prints the phases and, for the
text pass, where the threads' time goes).  Then three callers at once (as the end-to-end leg runs them).
    python tools/gpu/tail_scaling.py <out_dir> [genome_mbp] [pairs]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))


def main():
    out = sys.argv[1]
    mbp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
    os.makedirs(out, exist_ok=True)
    import subprocess
    import bench
    import bm2
    wd = os.environ.get("PROBE_WORKDIR", "/tmp/bm2_tail_probe")
    os.makedirs(wd, exist_ok=True)
    prefix, contigs = bench.prepare_genome(wd, mbp, int(os.environ.get("PROBE_SEED", 777)))
    fa, fb = os.path.join(wd, "s_1.fq"), os.path.join(wd, "s_2.fq")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_chunk.py"), prefix + ".contigs.npz", "901", str(pairs), "150", fa, fb, "s_"])
    t1, t2 = open(fa, "rb").read(), open(fb, "rb").read()
    ctx = bm2.Context(0, prefix)
    opt = bm2.default_opt()
    ch = bm2.FastqChunk(t1, t2, 32)
    ctx.batch_upload_chunk(ch); ctx.batch_run(opt); ctx.batch_finish(opt)
    aln, aln_off = ctx.batch_download_alnregs()
    buf = np.empty(int(3 * (int(ch.f.n_bases) + 200 * ch.n_reads)), np.uint8)
    res = {"reads": ch.n_reads, "single": [], "three_callers": []}
    threads = [int(x) for x in os.environ.get("SCALING_THREADS", "8 32 72 144 216").split()]
    for th in threads + threads[-2:]:
        so = bm2.default_sam_opt(n_threads=th)
        best = None
        for rep in range(3):
            print("[scaling] ---- %d threads, repeat %d" % (th, rep), file=sys.stderr, flush=True)
            t = time.perf_counter(); txt = ctx.sam(ch, opt, so, aln, aln_off, 0, True, out=buf); dt = time.perf_counter() - t
            best = dt if best is None or dt < best else best
        res["single"].append({"threads": th, "best_ms": best * 1e3, "bytes": int(len(txt))})
        print("[scaling] %d threads: best of 3 = %.1f ms (%.2f M reads/s)" % (th, best * 1e3, ch.n_reads / best / 1e6), file=sys.stderr, flush=True)
    for th in (72, 40):
        ctxs = [bm2.Context(share=ctx) for _ in range(3)]
        bufs = [np.empty(len(buf), np.uint8) for _ in range(3)]
        dts = [[] for _ in range(3)]

        def run(k):
            so = bm2.default_sam_opt(n_threads=th)
            for rep in range(4):
                t = time.perf_counter(); ctxs[k].sam(ch, opt, so, aln, aln_off, 0, True, out=bufs[k]); dts[k].append(time.perf_counter() - t)
        print("[scaling] ---- three callers x %d threads" % th, file=sys.stderr, flush=True)
        tt = [threading.Thread(target=run, args=(k,)) for k in range(3)]
        t = time.perf_counter()
        for x in tt:
            x.start()
        for x in tt:
            x.join()
        wall = time.perf_counter() - t
        res["three_callers"].append({"threads_each": th, "wall_s": wall, "reads_per_s": 12 * ch.n_reads / wall, "calls_ms": [[round(d * 1e3, 1) for d in v] for v in dts]})
        print("[scaling] three callers x %d threads: 12 chunks in %.2f s = %.2f M reads/s; calls %s" % (th, wall, 12 * ch.n_reads / wall / 1e6, res["three_callers"][-1]["calls_ms"]), file=sys.stderr, flush=True)
        for c in ctxs:
            c.close()
    json.dump(res, open(os.path.join(out, "tail_scaling_%s.json" % os.environ.get("SCALING_TAG", "x")), "w"), indent=1)
    ch.close(); ctx.close()


if __name__ == "__main__":
    main()
