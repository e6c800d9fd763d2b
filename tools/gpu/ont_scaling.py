#!/usr/bin/env python3
"""Config 5 (10 kb `-x ont2d` reads) against BATCH SIZE, in one process: reads/s of the device hot path for chunks of 10 000 ... 40 000 reads
(the reads of the largest chunk generated once, the smaller chunks are its prefixes), stage times and the device memory the chunk's workspaces
take.  A step of 10 000 reads leaves the GPU mostly empty (157 wavefronts of k_walk<1> for 1024 SIMDs; chaining as long as its slowest reads):
this is the measurement behind the chunk size bench.py uses for config 5.

    python tools/gpu/ont_scaling.py <out.json> [--sizes 10000,20000,40000] [--genome-mbp 3100] [--steps 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--sizes", default="10000,20000,40000")
    ap.add_argument("--genome-mbp", type=int, default=3100)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--workdir", default=os.environ.get("BM2_BENCH_WORKDIR", "/tmp/bm2_bench"))
    a = ap.parse_args()
    import bench
    import bm2
    from tools import refio, synth
    try:
        import torch
        mem = lambda: torch.cuda.mem_get_info()[0] / 1e9                      # noqa: E731  (free device memory, GB)
    except Exception:                                                         # noqa
        mem = lambda: float("nan")                                            # noqa: E731
    emu = os.environ.get("BM2_EMU_LIB")
    if emu:
        bm2.LIB_PATH = emu
    sizes = sorted(int(x) for x in a.sizes.split(","))
    seed = 20260924
    os.makedirs(a.workdir, exist_ok=True)
    prefix, contigs = bench.prepare_genome(a.workdir, a.genome_mbp, seed)
    t = time.time()
    seqs = synth.make_reads_long(seed, contigs(), sizes[-1], mean_len=10000, max_len=30000)
    print("[ont_scaling] %d reads generated in %.1fs" % (len(seqs), time.time() - t), file=sys.stderr, flush=True)
    free0 = mem()
    ctx = bm2.Context(0, prefix)
    opt = bm2.default_opt(**bench.ONT2D)
    rows = []
    for n in sizes:
        enc, off, ln = refio.pack_reads(seqs[:n])
        ctx.batch_upload(enc, off, ln)
        ctx.batch_run(opt)                                   # warm-up: workspaces of this size
        kms = {}
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.batch_run(opt)
            for name, ms in ctx.batch_kernel_ms():
                kms[name] = kms.get(name, 0.0) + ms / a.steps
        dt = (time.perf_counter() - t0) / a.steps
        st = ctx.batch_stats()
        stage = {}
        for k, v in kms.items():
            stage[k.split(".")[0]] = stage.get(k.split(".")[0], 0.0) + v
        cn = np.asarray(ctx.batch_fetch("counters", np.uint64), np.float64)
        row = {"reads": n, "bases": int(ln.sum()), "ms_per_step": dt * 1e3, "reads_per_s": n / dt, "stage_ms": stage, "kernel_ms": kms,
               "device_memory_in_use_gb": free0 - mem(), "seeds": int(st["n_sa"]), "reads_chained_serially": int(cn[16]), "reads_by_islands": int(cn[17])}
        rows.append(row)
        print("[ont_scaling] %6d reads: %8.1f ms per step = %7.0f reads/s  %s  memory %.1f GB, %d serial reads"
              % (n, dt * 1e3, n / dt, {k: round(v, 1) for k, v in stage.items()}, row["device_memory_in_use_gb"], row["reads_chained_serially"]), file=sys.stderr, flush=True)
    json.dump({"workload": "ONT-like reads, mean 10 kb, cap 30 kb, -x ont2d, %d Mbp genome" % a.genome_mbp, "knobs": {k: v for k, v in os.environ.items() if k.startswith("BM2_")},
               "rows": rows}, open(a.out, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
