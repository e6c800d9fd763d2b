#!/usr/bin/env python3
"""bench.py -- throughput of libbm2 on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the device pipeline (SMEM seeding -> SA lookup -> chaining -> banded extension -> regs,
i.e. mem_kernel1_core + mem_kernel2_core up to bwamem.cpp:1152) over one chunk of synthetic reads that is already
resident in HBM: `value`.  Reads shard across GPUs (per-GPU index replica, no collective on the data path;
torch.distributed is used only for the barrier and the max-over-ranks of the timed region) => weak scaling.

Prints ONE COMPACT JSON line (< 4 KB, tools/bench_line.py) on rank 0 as the LAST line of stdout -- the driver contract's keys + `roofline`,
`cpu_baseline`, `parity`, `value_end_to_end` and one-number summaries of the other legs; the run's FULL record (everything below with its notes) goes
to --full-json (default <workdir>/bench_full_<workload>.json) and as one-liners to stderr.  The objects of the full record:
  roofline     -- the FM-index seeding kernel against the HBM peak, from ALGORITHMIC bytes (128 B per backwardExt,
                  SURVEY.md section 8(d)) over its HIP-event-timed launches inside the timed region; `achieved_counter` is the
                  same with the HBM bytes the PMC passes measured (profiles/)
  cpu_baseline -- the compiled reference (oracle/_ref/bwa-mem2.<isa> mem) on this host's cores, bounded sample
  parity       -- the gate: a 512-aligned PREFIX of the timed chunk (the block rule of bwamem.cpp:834 makes a prefix the only valid
                  sample) through the reference on the same index: regs before / after mem_sort_dedup_patch byte for byte against
                  oracle/_ref/refdump, SAM text against `bwa-mem2 mem`.  A mismatch makes the run fail (exit code 3).
  end_to_end   -- the metric as stated: FASTQ text in host memory -> SAM text in host memory over several distinct chunks through a
                  pipeline of host threads (reader | device workers | tail workers), the stages of consecutive chunks overlapping
                  (reported beside `value`, never instead of it); under a watchdog, inside the run's time budget (--budget-s)
--workload ont2d / bsw: BASELINE configs 5 and 2 as workloads of their own (same JSON shape).
"""
import argparse
import json
import os
import queue
import re
import subprocess
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # the extension stage forks 8 concurrent launches per phase, chaining 5; 8 / 16 / 24 measured (profiles/r04c, r04d)
os.environ.setdefault("BM2_MALLOC_TUNE", "1")        # this process is the library's host: it opts in to the allocator settings of sam_tail.cpp

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
from tools import bench_line                                                           # noqa: E402
from tools.bench_legs import (CONFIG5_READS, HBM_PEAK_GBS, ONT2D, RANDOM_LINE_GLPS, alnregs_records, bench_bsw, binding_leg, cpu_baseline,      # noqa: E402,F401
                              cpu_baseline_from, end_to_end, host_threads, log, make_extension_pairs, parity_gate, pe_chunk, prepare_genome,
                              ref_binary, regs_records, run_reference_mem, run_side_workloads, s1_binding_leg, sam_gate, sam_lines, side_workload)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pe150", choices=["pe150", "ont2d", "bsw"],
                    help="pe150: BASELINE config 3 shape (the metric); ont2d: config 5 shape (10 kb reads, -x ont2d); bsw: config 2 shape "
                         "(the banded-SW kernel alone on the extension tasks of 150 bp reads, batch resident: S1 = bm2_bsw_upload / _run / _download)")
    ap.add_argument("--bsw-pairs", type=int, default=int(os.environ.get("BM2_BENCH_BSW_PAIRS", 2874000)),
                    help="bsw: extension tasks per step (default = the 2.874 tasks per read the pe150 workload measures x 1 M reads)")
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("BM2_BENCH_GENOME_MBP", 3100)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BM2_BENCH_READS", 0)),
                    help="reads per GPU per step (both mates counted); default 1000000 (pe150) / 10000 (ont2d)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-pairs", type=int, default=int(os.environ.get("BM2_BENCH_CPU_PAIRS", 250000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-reads", type=int, default=int(os.environ.get("BM2_BENCH_PARITY_READS", 204800)),
                    help="reads of the timed chunk's prefix that go through refdump (REGPRG / REGFIN byte for byte) and `bwa-mem2 mem` (SAM)")
    ap.add_argument("--resident-chunks", type=int, default=int(os.environ.get("BM2_BENCH_RESIDENT", 4)),
                    help="distinct chunks the timed steps go round (all resident before the clock starts; capped by --warmup and --steps)")
    ap.add_argument("--distinct-chunks", type=int, default=1,
                    help="ont2d: the timed steps go through this many DISTINCT chunks of --reads reads (BASELINE config 5 names 100 000 reads; one chunk's workspaces fill the "
                         "HBM beside the index, so the chunks take turns: every step is timed on its own, its chunk uploaded before its clock starts)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--parity-regs-reads", type=int, default=256, help="ont2d: reads of the gate whose stage dumps (refdump, one host thread) are compared; all --parity-reads go through `bwa-mem2 mem` and the SAM comparison")
    ap.add_argument("--no-binding-s1", action="store_true", help="bsw: skip the timing of `bwa-mem2.bm2s1 mem` (S1 on the GPU inside the reference's program) beside `bwa-mem2.<isa> mem`")
    ap.add_argument("--no-side-workloads", action="store_true", help="skip configs 5 and 2 (objects `config5` / `config2` of the pe150 line: --workload ont2d / bsw as processes of their own)")
    ap.add_argument("--no-binding", action="store_true", help="skip the drop-in timing (`bwa-mem2.bm2 mem` beside `bwa-mem2.<isa> mem` on the first two end-to-end chunks' files)")
    ap.add_argument("--e2e-chunks", type=int, default=int(os.environ.get("BM2_BENCH_E2E_CHUNKS", 10)))
    ap.add_argument("--e2e-rounds", type=int, default=int(os.environ.get("BM2_BENCH_E2E_ROUNDS", 10)),
                    help="the end-to-end leg goes this many times round its distinct chunks (a run of 10 chunks is one third fill and drain of the pipeline; the default, "
                         "10 x 10 chunks of 1 M reads, is BASELINE config 4's read count -- 100 M reads -- through one GPU)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE chunk of --reads reads is cut at multiples of 512 over the ranks (SURVEY.md 8(e)) instead of one chunk per rank")
    ap.add_argument("--workdir", default=os.environ.get("BM2_BENCH_WORKDIR", "/tmp/bm2_bench"))
    ap.add_argument("--full-json", default=os.environ.get("BM2_BENCH_FULL_JSON"),
                    help="where the run's FULL record goes (every leg with its notes, per-kernel tables, the side workloads' own records); default "
                         "<workdir>/bench_full_<workload>.json.  stdout's last line is the compact object of tools/bench_line.py (< 4 KB)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("BM2_BENCH_BUDGET_S", 1500)),
                    help="wall-clock budget of the whole run: an optional leg (parity gate, CPU baseline, end-to-end) that could not finish inside it is "
                         "skipped and says so in the JSON line; the end-to-end leg also runs under a watchdog")
    a = ap.parse_args()
    if not a.full_json:
        a.full_json = os.path.join(a.workdir, "bench_full_%s.json" % a.workload)
    t_start = time.time()
    hung = False

    def time_left():
        return a.budget_s - (time.time() - t_start)

    from tools import dist_util, synth
    rank, world, local = dist_util.env_rank()
    import torch
    import bm2
    emu = os.environ.get("BM2_EMU_LIB")                  # test hook: the host emulator of the device sources (tools/emu) stands in for the
    if emu:                                              # GPU so that this script's own logic can be checked on a CPU box; never a result
        bm2.LIB_PATH = emu
        torch.cuda.synchronize = lambda *x: None
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libbm2 has no CPU fallback)")
    os.makedirs(a.workdir, exist_ok=True)
    seed = 20260924
    # Config 2 AS WORDED (seeding on the host, banded SW on the GPU, inside the reference's own program: `bwa-mem2.bm2s1 mem`) is timed FIRST, before this
    # process opens the device: the binding's small device batches take 6.45 s per million reads with the GPU to themselves and 8.3-8.7 s beside a process
    # that merely holds contexts and queues there (profiles/r05e_*, r05z_*: closing this process's bm2 context did not help -- torch's stays).  A user of
    # the binding runs it alone; the object lands in `config2.s1_binding` of the line.
    early_s1 = None
    if world == 1 and not emu and a.workload == "pe150" and not a.no_side_workloads and not a.no_binding_s1 and not a.no_cpu_baseline:
        try:
            prefix, contigs = prepare_genome(a.workdir, a.genome_mbp, seed)
            f1, f2 = os.path.join(a.workdir, "cpu_1.fq"), os.path.join(a.workdir, "cpu_2.fq")
            if not os.path.exists(f1):
                c1, c2 = synth.make_reads_pe(seed + 5, contigs(), a.cpu_pairs, L=a.read_len)
                synth.write_fastq(f1, c1, suffix="/1"); synth.write_fastq(f2, c2, suffix="/2")
                del c1, c2
            early_s1 = s1_binding_leg(a.workdir, prefix)
            if isinstance(early_s1, dict):
                early_s1["when"] = "before this process opened the device (the GPU to the binding alone, as a user runs it)"
        except Exception as e:                                                        # noqa
            early_s1 = {"error": str(e)}
    # ... and so are BASELINE configs 5 and 2 as workloads of their own (processes of their own, `side_workload`): beside this process's idle contexts the
    # same config-5 chunk took 1196 ms per step (chain stage 685 ms: its persistent launches and the consumer kernel that waits beside its producer are what a
    # second process's queues on the GPU disturb most), 965-980 ms with the GPU to itself (profiles/r05zzz_bench.json against r05t / r05s)
    early_side = None
    if world == 1 and not emu and a.workload == "pe150" and not a.no_side_workloads:
        early_side = {}
        try:
            prepare_genome(a.workdir, a.genome_mbp, seed)
            early_side = run_side_workloads(a, early_s1, time_left)
        except Exception as e:                                                        # noqa
            log("side workloads ahead of the main line: %s" % e)
            early_side = None
    if not emu:
        torch.cuda.set_device(local)
    dist_util.init("gloo" if emu else "nccl", world, None if emu else torch.device("cuda", local))     # "nccl" is RCCL on ROCm

    if a.workload == "bsw":
        rc = bench_bsw(a, bm2, torch, dist_util, rank, world, local, emu, seed)
        dist_util.finish(world)
        sys.exit(rc)
    if rank == 0:
        prefix, contigs = prepare_genome(a.workdir, a.genome_mbp, seed)
    dist_util.barrier(world)
    if rank != 0:
        prefix, contigs = prepare_genome(a.workdir, a.genome_mbp, seed)

    t = time.time()
    free0 = torch.cuda.mem_get_info(local)[0] if not emu else 0
    ctx = bm2.Context(local, prefix)
    replica_gb = (free0 - torch.cuda.mem_get_info(local)[0]) / 1e9 if not emu else None       # (what the device's free memory dropped by)
    log("rank %d: index replica in HBM after %.1fs (%s GB)" % (rank, time.time() - t, "%.2f" % replica_gb if replica_gb is not None else "-"))
    ont = a.workload == "ont2d"
    paired = not ont
    if ont:
        n_reads = a.reads or 10000                           # ~100 Mbases: the chunk of `bwa-mem2 mem -K 100000000`
        opt, opt_args = bm2.default_opt(**ONT2D), ["-x", "ont2d"]
        seqs = synth.make_reads_long(dist_util.shard_seed(seed, rank), contigs(), n_reads, mean_len=10000, max_len=30000)
        longest = int(np.argmax([len(x) for x in seqs]))      # the parity gate takes the first reads of the chunk: the chunk's longest read (the
        seqs[0], seqs[longest] = seqs[longest], seqs[0]       # 30 kb cap is reached in any chunk of thousands) is one of them
        from tools import refio
        enc, off, ln = refio.pack_reads(seqs)
    else:
        n_reads = a.reads or 1000000
        opt, opt_args = bm2.default_opt(), []
        seqs = pe_chunk(a.workdir, contigs, dist_util.shard_seed(seed, 0 if a.strong else rank), n_reads, a.read_len)
        if a.strong and world > 1:                          # this rank's part of the one chunk: [lo, hi) at multiples of 512 reads
            b = dist_util.shard_bounds(len(seqs), world)
            seqs = seqs[b[rank]:b[rank + 1]]
        n_reads = len(seqs)
        enc = seqs.reshape(-1)
        off = np.arange(n_reads, dtype=np.int64) * a.read_len
        ln = np.full(n_reads, a.read_len, np.int32)
    n_bases = int(np.asarray(ln, np.int64).sum())
    ctx.batch_upload(enc, off, ln)
    # The timed steps go round SEVERAL distinct chunks, all resident in HBM before the clock starts (each on a context of its own that shares
    # the index replica): a step does not find the L2 / MALL / TLB state and the learned workspace sizes of the same reads it just aligned.
    # As many chunks as warm-up steps allow (every chunk runs once untimed: its workspaces are sized by then), at most --resident-chunks.
    n_res = 1 if (ont or a.strong) else max(1, min(a.resident_chunks, a.warmup, a.steps))
    run_ctx, res_chunks = [ctx], []
    if n_res > 1:
        t = time.time()
        meta = prefix + ".contigs.npz"
        procs = []
        for k in range(1, n_res):
            fa, fb = os.path.join(a.workdir, "res_r%d_%d_1.fq" % (rank, k)), os.path.join(a.workdir, "res_r%d_%d_2.fq" % (rank, k))
            procs.append((fa, fb, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gen_chunk.py"), meta, str(dist_util.shard_seed(seed, rank) + 7000 + k),
                                                    str(n_reads // 2), str(a.read_len), fa, fb, "r%d_" % k])))
        for fa, fb, pr in procs:
            if pr.wait() != 0:
                raise SystemExit("resident chunk generator failed")
            ch = bm2.FastqChunk(open(fa, "rb").read(), open(fb, "rb").read(), 0)
            os.remove(fa); os.remove(fb)
            c2 = bm2.Context(share=ctx)
            c2.batch_upload_chunk(ch)
            run_ctx.append(c2); res_chunks.append(ch)
        log("rank %d: %d distinct chunks resident after %.1fs" % (rank, n_res, time.time() - t))

    for i in range(a.warmup):
        run_ctx[i % n_res].batch_run(opt)
    kms = {}
    sc_sum = None
    n_distinct = max(1, min(a.distinct_chunks, a.steps)) if ont else 1
    reads_timed = None
    if n_distinct > 1:
        # BASELINE config 5 as worded -- 100 000 reads in one run -- on a GPU whose HBM holds ONE chunk's workspaces beside the index: the steps go through
        # n_distinct different chunks that take turns on the one context.  Each step is timed on its own (barrier + synchronize on both sides) with its chunk
        # uploaded BEFORE its clock starts -- `value` keeps its meaning, reads resident in HBM -- and the step times are added up.  The last step runs the
        # chunk the parity gate samples (chunk 0), so the gate reads that chunk's regs.
        from tools import refio as _refio
        chunks = [(enc, off, ln)]
        t = time.time()
        for k in range(1, n_distinct):
            sq = synth.make_reads_long(dist_util.shard_seed(seed, rank) + 31 * k, contigs(), n_reads, mean_len=10000, max_len=30000)
            chunks.append(_refio.pack_reads(sq))
            del sq
        log("rank %d: %d distinct chunks of %d long reads generated in %.1fs" % (rank, n_distinct, n_reads, time.time() - t))
        dt_local, reads_timed = 0.0, 0
        for i in range(a.steps):
            e_k, o_k, l_k = chunks[(i + 1) % n_distinct]
            ctx.batch_upload(e_k, o_k, l_k)
            torch.cuda.synchronize()
            dist_util.barrier(world)
            t0 = time.perf_counter()
            ctx.batch_run(opt)
            torch.cuda.synchronize()
            dist_util.barrier(world)
            dt_local += time.perf_counter() - t0
            reads_timed += len(l_k)
            for name, ms in ctx.batch_kernel_ms():
                kms[name] = kms.get(name, 0.0) + ms
        if a.steps % n_distinct:                              # (the gate below compares chunk 0's regs: make it the resident one)
            ctx.batch_upload(enc, off, ln); ctx.batch_run(opt)
        dt = dist_util.max_over_ranks(dt_local, world, "cpu" if emu else "cuda")
        del chunks
    else:
        torch.cuda.synchronize()
        dist_util.barrier(world)
        t0 = time.perf_counter()
        for i in range(a.steps):
            c = run_ctx[i % n_res]
            c.batch_run(opt)                                     # returns after the library's stream has drained
            for name, ms in c.batch_kernel_ms():
                kms[name] = kms.get(name, 0.0) + ms
        torch.cuda.synchronize()
        dist_util.barrier(world)
        dt = dist_util.max_over_ranks(time.perf_counter() - t0, world, "cpu" if emu else "cuda")
    st = ctx.batch_stats()
    parts = ctx.batch_parts() if hasattr(ctx, "batch_parts") else 1      # (a chunk runs as `parts` parts beside each other: the per-kernel times are summed over them)
    if n_res > 1:                                            # work counters: the mean over the chunks that ran (they differ by a fraction of a per cent)
        sts = [c.batch_stats() for c in run_ctx[:min(n_res, a.steps)]]
        st = {k: sum(x[k] for x in sts) / len(sts) for k in st}
        sc_sum = sum(np.asarray(c.batch_fetch("seed_counters", np.uint64), np.float64) for c in run_ctx[:min(n_res, a.steps)]) / min(n_res, a.steps)
    for c in run_ctx[1:]:                                    # (their workspaces are not needed by the legs that follow)
        c.close()
    for ch in res_chunks:
        ch.close()

    rc = 0
    if rank == 0:
        steps = max(a.steps, 1)
        n_total = (a.reads or 1000000) if (a.strong and not ont) else world * n_reads
        value = n_total * a.steps / dt
        if reads_timed is not None:
            value = world * reads_timed / dt
        kern_ms = {k: v / steps for k, v in kms.items()}     # every timed interval of the library's stream (HIP events)
        stage_ms = {}
        for k, v in kern_ms.items():                         # "smem.walk1" ... -> stage "smem"
            stage_ms[k.split(".")[0]] = stage_ms.get(k.split(".")[0], 0.0) + v
        # the FM-index seeding kernels: two 64-B CP_OCC lines per backwardExt; the dominant one is k_bwd
        # (two launches per step: the backward phases of pass 1 and of pass 2)
        sc = sc_sum if sc_sum is not None else ctx.batch_fetch("seed_counters", np.uint64)
        ext_of = {"walk1": int(sc[12]), "walk2": int(sc[13]), "walk3": int(sc[14]), "bwd1": int(sc[15]), "bwd2": int(sc[16])}
        smem_ms = stage_ms.get("smem", 0.0)
        # Since round 5 k_bwd hands the tasks that have had BM2_BWD_EXPORT_AGE extensions over to k_bwd_cont, a launch right behind it (intervals
        # smem.cont1 / smem.cont2): the backward phase of a pass is the two launches together -- the pass's backwardExt count covers both (and the
        # wavefront-per-task kernel of the long lists, which runs beside k_bwd), so does the time it is divided by.
        cont_ms = kern_ms.get("smem.cont1", 0.0) + kern_ms.get("smem.cont2", 0.0)
        bwd_ms = (kern_ms.get("smem.bwd1", 0.0) + kern_ms.get("smem.bwd2", 0.0) + cont_ms) / (2.0 * parts)          # average duration of a pass's backward phase (every part launches its own two)
        bwd_bytes = 128.0 * (ext_of["bwd1"] + ext_of["bwd2"]) / (2.0 * parts)                              # algorithmic bytes per launch
        ach = bwd_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
        roof_kernel, roof_launches = ("k_bwd (+ k_bwd_cont, the tasks it hands over)" if cont_ms > 0 else "k_bwd"), 2 * parts
        handed = None
        if len(sc) >= 27 and (sc[21] > 0 or sc[22] > 0):
            handed = {"export_age": int(os.environ.get("BM2_BWD_EXPORT_AGE", "256") or 0), "tasks_pass1": int(sc[21]), "tasks_pass2": int(sc[22]),
                      "rows_walked_by_k_bwd_cont": [int(sc[25]), int(sc[26])], "k_bwd_cont_ms": [kern_ms.get("smem.cont1", 0.0) / parts, kern_ms.get("smem.cont2", 0.0) / parts]}
        fm_kernels = {}                                       # every FM-index kernel of the step against the same peak: 128 algorithmic bytes per backwardExt
        for kn, ev in (("k_walk<1>", "walk1"), ("k_bwd + k_bwd_cont (pass 1)", "bwd1"), ("k_walk<2>", "walk2"), ("k_bwd + k_bwd_cont (pass 2)", "bwd2")):
            ms_k = (kern_ms.get("smem." + ev, 0.0) + kern_ms.get("smem." + ev.replace("bwd", "cont"), 0.0) * (ev.startswith("bwd"))) / parts    # (per launch: one per part)
            if ms_k > 0:
                gbs = 128.0 * ext_of[ev] / parts / (ms_k * 1e-3) / 1e9
                fm_kernels[kn] = {"ms": ms_k, "backwardExt": ext_of[ev] / parts, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
        w1 = fm_kernels.get("k_walk<1>")
        if w1 and w1["ms"] > 2.0 * bwd_ms:                    # long reads: the forward walks, not the backward phases, are the seeding stage
            roof_kernel, roof_launches = "k_walk<1>", parts
            bwd_ms, bwd_bytes, ach = w1["ms"], 128.0 * w1["backwardExt"], w1["achieved"]
        fm_bytes = 128.0 * st["n_ext"]
        # (in parts the stage intervals of the parts overlap and `smem_ms` is their sum: per part the stage moved fm_bytes / parts in smem_ms / parts)
        stage_ach = fm_bytes / (smem_ms * 1e-3) / 1e9 if smem_ms > 0 else 0.0
        cells = st["n_sw_cells"]
        ext_ms = stage_ms.get("extend", 0.0)
        lane_use = wave_cell_share = chain_kernel = None
        try:                                                 # the lane kernel's own count of its column-pair trips (128 lane slots each) and the wavefront kernel's cells
            cn = np.asarray(ctx.batch_fetch("counters", np.uint64), np.float64)
            if cn[7] > 0:
                lane_use, wave_cell_share = float((cn[5] - cn[8]) / (128.0 * cn[7])), float(cn[8] / max(cn[5], 1.0))
            if cn[17] + cn[16] > 0:                           # the island kernel of long-read chaining (chain.hip): its own counts and phase clock (100 MHz ticks summed over its reads)
                nr = max(cn[17], 1.0)
                names = ("table + SMEM cuts", "stage seeds + file buckets", "island of every seed", "places (scan)", "seeds of an island together", "islands chained", "read finished")
                chain_kernel = {"kernel": "k_chain_islands", "reads_by_islands": int(cn[17]), "reads_chained_serially_equal_keys": int(cn[16] - (cn[39] if len(cn) > 39 else 0)), "islands_per_read": float(cn[18] / nr),
                                "chains_per_read": float(cn[27] / nr), "chains_past_the_weight_test_per_read": float(cn[26] / nr),
                                "phase_ms_per_read": {nm: float(cn[19 + i] * 1e-5 / (cn[17] + cn[16])) for i, nm in enumerate(names)},
                                "phase_ms_slowest_read": {nm: float(cn[31 + i] * 1e-5) for i, nm in enumerate(names)},
                                "serial_reads": {"ms_per_read": float(cn[28] * 1e-5 / max(cn[16], 1.0)), "seeds_per_read": float(cn[29] / max(cn[16], 1.0)), "slowest_ms": float(cn[30] * 1e-5),
                                                 "where": "k_chain_serial: a launch of its own beside the island kernel, the tree's internal nodes in LDS"}}
            if len(cn) >= 48 and cn[46] > 0:                   # BM2_CHAIN_CLOCK=1: lane 0's walk of the wavefront-per-read chain launches, clocked (100 MHz ticks)
                heavy_clock = {"reads": int(cn[46]), "seeds_per_read": float(cn[47] / cn[46]),
                               "ms_per_read": {"staging by the 64 lanes": float(cn[43] * 1e-5 / cn[46]), "mem_chain_seeds (lane 0)": float(cn[44] * 1e-5 / cn[46]),
                                               "traversal + mem_chain_flt + output (lane 0)": float(cn[45] * 1e-5 / cn[46])}}
                chain_kernel = dict(chain_kernel or {}, k_chain_heavy_clock=heavy_clock)
        except Exception:                                                             # noqa
            pass
        dominant = max(stage_ms, key=stage_ms.get) if stage_ms else None
        traffic, pmc_src = None, None                        # HBM bytes per k_bwd launch from the committed PMC passes, same workload only
        ext_pmc = None
        for fn in ("r06_k_bwd_pmc.json", "r05_k_bwd_pmc.json", "r04_k_bwd_pmc.json", "r03_k_bwd_pmc.json", "r02_k_bwd_pmc.json", "r01_k_bwd_pmc.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
                wl = pm["workload"]
                if wl["genome_mbp"] == a.genome_mbp and wl["reads_per_gpu_per_step"] == n_reads and wl["read_len"] == a.read_len and not ont:
                    traffic, pmc_src = pm["hbm_bytes_per_launch"], "profiles/" + fn
                    break
            except Exception:
                pass
        if ont and traffic is None:                             # config 5: the forward walks are the seeding stage -- HBM bytes per k_walk<1> launch from the committed FETCH pass of the same chunk size
            for fn in ("r05_ont2d_k_walk_pmc.json",):
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    if pm["workload"]["genome_mbp"] == a.genome_mbp and pm["workload"]["reads_per_gpu_per_step"] == n_reads and roof_kernel.startswith("k_walk"):
                        traffic, pmc_src = pm["hbm_bytes_per_launch"], "profiles/" + fn
                        break
                except Exception:
                    pass
        ext_src = None
        # SQ counter passes of the extension stage (tools/pmc_to_profiles.py), the newest committed one OF THIS WORKLOAD (a pass of the 150 bp
        # workload says nothing about the kernels a 10 kb chunk runs)
        for fn in (("r05_ont2d_ext_pmc_sq.json", "r04_ont2d_ext_pmc_sq.json") if ont else ("r06_ext_pmc_sq.json", "r05_ext_pmc_sq.json", "r04_ext_pmc_sq.json", "r03_ext_pmc_sq.json", "r02_ext_pmc_sq.json")):
            try:
                ext_pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                ext_src = "profiles/" + fn
                break
            except Exception:
                pass
        ach_counter = traffic / (bwd_ms * 1e-3) / 1e9 if traffic and bwd_ms > 0 else None
        lines_counter = traffic / 64.0 / (bwd_ms * 1e-3) / 1e9 if traffic and bwd_ms > 0 else None
        wl_name = ("config 5 shape: %d ONT-like reads (mean 10 kb, cap 30 kb, ~10%% error) per GPU per step, `-x ont2d`%s" % (n_reads,
                   "" if n_distinct <= 1 else "; the %d steps go through %d DISTINCT chunks = %d reads in this run, every step timed on its own with its chunk resident" % (a.steps, n_distinct, reads_timed))) if ont else \
                  ("config 3 shape: %d x %d bp PE reads per GPU per step" % (n_reads, a.read_len))
        out = {
            "metric": "aligned reads/s (150bp PE vs GRCh38) at 1/2/4/8 GPU; SAM bit-exact vs ref",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            # what `value` is and is not: the task contract of this run ("whole-job throughput with inputs already resident in HBM when the timed
            # region starts ... the PCIe-inclusive rate ... is never `value`") makes it the device hot path; the metric AS WORDED -- FASTQ text in,
            # SAM text out, bit-exact -- is `end_to_end.value` of the same line, repeated here so that nobody has to look for it
            "value_scope": "device hot path (seed -> chain -> extend -> regs at bwamem.cpp:1152), reads resident in HBM; FASTQ -> SAM of the same library: value_end_to_end",
            "value_scope_short": "device hot path, reads resident in HBM -> regs (bwamem.cpp:1152); FASTQ->SAM: value_end_to_end",
            "value_end_to_end": None,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic" if not emu else "synthetic; HOST EMULATOR RUN (not a measurement)",
            "parts_per_chunk": parts,
            "parts_note": None if parts <= 1 else ("the chunk runs as %d parts (cut at multiples of 512 reads) on streams and workspaces of their own, beside each other: "
                                                   "`stage_ms_per_step` and every per-kernel time are SUMS over the parts' intervals, which overlap -- they add up to more "
                                                   "than `ms_per_step`; the roofline's launch is one part's (BM2_N_SUB=1: one part, stages in sequence)" % parts),
            "index_replica_gb": round(replica_gb, 2) if replica_gb is not None else None,
            "index_replica_layout": "Occ checkpoints 64 B per 64 symbols, suffix array 5 B per 8 positions, reference string "
                                    + ("1 byte per base (BM2_REF_BYTES=1)" if os.environ.get("BM2_REF_BYTES", "0") not in ("", "0") else "2 bits per base (refseq.h)"),
            "config": {"workload": wl_name + " (SMEM+SAL+chain+banded-SW all on device), synthetic %d Mbp genome with planted repeats/ALT/"
                                   "N-gaps, indexed in-run by bm2_index_build (3100 Mbp = GRCh38 size); `value` = device hot path "
                                   "with the reads resident in HBM (the steps go round %d distinct chunks), output = mem_alnreg_t regs at bwamem.cpp:1152; the FASTQ -> SAM "
                                   "rate of the same library is `end_to_end.value`" % (a.genome_mbp, n_res),
                       "workload_short": (("config 5 shape: %d ONT-like reads (mean 10 kb) per GPU per step, -x ont2d%s" % (n_reads, "" if n_distinct <= 1 else " (%d distinct chunks = %d reads)" % (n_distinct, reads_timed))) if ont else
                                          ("config 3 shape: %d x %d bp PE reads per GPU per step" % (n_reads, a.read_len)))
                                         + ", synthetic %d Mbp genome; SMEM+SAL+chain+banded SW on device, reads resident in HBM" % a.genome_mbp,
                       "parallelism_short": ("1 chunk cut over %d GPU(s), " if a.strong else "1 chunk per GPU x %d GPU(s), ") % world + "index replica per GPU, no collectives",
                       "resident_chunks": n_res,
                       "distinct_chunks": n_distinct, "steps_cover_reads": (reads_timed if reads_timed is not None else None),
                       "reads_per_gpu_per_step": n_reads, "bases_per_gpu_per_step": n_bases, "read_len": a.read_len if not ont else None,
                       "genome_mbp": a.genome_mbp,
                       "parallelism": ("ONE chunk cut at multiples of 512 reads over %d GPU(s) (strong scaling), " if a.strong else "one chunk per GPU over %d GPU(s), ") % world
                                      + "index replica per GPU, no collectives"},
            "multi_gpu": "no scaling curve has been measured by this repository (every box it saw has one GPU): `--gpus N` under torch.distributed.run shards chunks over ranks "
                         "(weak scaling, one chunk per rank per step, no data-path collective), `--strong` cuts ONE chunk over the ranks; tests/test_dist_cpu.py and tests/test_sharded.py cover both",
            "knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("BM2_") and k != "BM2_EMU_LIB"},      # launch-policy / pipeline settings in force ({} = defaults)
            "stage_ms_per_step": stage_ms, "dominant_stage": dominant,
            "work_per_read": {"backwardExt": st["n_ext"] / n_reads, "lf_steps": st["n_lf"] / n_reads,
                              "sa_lookups": st["n_sa"] / n_reads, "sw_tasks": st["n_sw_tasks"] / n_reads,
                              "sw_cells": cells / n_reads, "regs": st["n_reg"] / n_reads},
            "roofline": {"kernel": roof_kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_src,
                         "achieved_counter": ach_counter,
                         "frac_counter": ach_counter / HBM_PEAK_GBS if ach_counter else None,
                         "algorithmic_bytes_per_launch": bwd_bytes, "avg_launch_ms": bwd_ms, "launches_per_step": roof_launches,
                         "fm_index_kernels": fm_kernels,
                         "handed_over": handed,
                         "random_line_ceiling_glines": RANDOM_LINE_GLPS,
                         "delivered_glines": lines_counter,
                         "frac_of_random_line_ceiling": lines_counter / RANDOM_LINE_GLPS if lines_counter else None,
                         "note": "since round 4 pass 3 of the seeding (k_walk<3>, forward-only, no LDS) runs BESIDE k_bwd of pass 1 instead of beside k_walk<1>: the stage is "
                                 "shorter (walk1 8.4 -> 5.4 ms, stage 36.3 -> ~35 ms) while k_bwd's own pass-1 interval now shares the GPU -- its launches are "
                                 "averaged all the same; `seeding_stage.frac` is the stage-wide figure, `fm_index_kernels` every kernel's own.  "
                                 "k_bwd fetches isolated 64-byte lines; this GPU delivers ~55 G such lines/s (tools/ubench/randline.hip). "
                                 "`achieved` counts 128 algorithmic bytes per backwardExt; `achieved_counter` / `delivered_glines` count "
                                 "the bytes / lines HBM actually delivered (FETCH_SIZE + WRITE_SIZE of the committed PMC passes): two "
                                 "ends of an interval in one CP_OCC block and L2-resident first steps make them smaller",
                         "seeding_stage": {"kernels": "k_walk<1> + k_bwd + k_walk<2> + k_bwd (+ k_walk<3> on a second stream)",
                                           "ms": smem_ms, "algorithmic_bytes": fm_bytes, "achieved": stage_ach,
                                           "frac": stage_ach / HBM_PEAK_GBS,
                                           "kernel_ms": {k: v for k, v in kern_ms.items() if k.startswith("smem.")},
                                           "backwardExt_per_kernel": ext_of}},
            "extend_kernel": {"kernel": "k_ext_seeds + k_ext_wave", "gcups": cells / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0,
                              "stage_ms": ext_ms, "cells_per_step": cells,
                              "lane_use_of_the_column_loop": lane_use, "cell_share_of_the_wavefront_kernel": wave_cell_share,
                              "valu_frac": ext_pmc.get("valu_frac") if ext_pmc else None,
                              "valu_peak_wave_insts_per_s": ext_pmc.get("valu_peak_wave_insts_per_s") if ext_pmc else None,
                              "valu_peak_source": ext_pmc.get("valu_peak_source") if ext_pmc else None,
                              "pmc_stage_ms": ext_pmc.get("extend_stage_ms") if ext_pmc else None,
                              "lds_conflict_frac": ext_pmc.get("lds_conflict_frac") if ext_pmc else None,
                              "pmc_source": ext_src,
                              "note": "valu_frac / lds_conflict_frac are of the committed counter pass (its own stage time: pmc_stage_ms), the rest is of this run"},
        }
        if chain_kernel:
            if "serial_reads" in chain_kernel:                   # what bounds the chaining stage of a long-read chunk, from the kernel's own clock
                sr = chain_kernel["serial_reads"]
                chain_kernel["stage_bound"] = {"kind": "latency (k_chain_islands: one wavefront per read, six per CU; the reads with equal chain keys are chained again by ONE lane of "
                                                       "k_chain_serial beside it, the kbtree's internal nodes in LDS)",
                                               "floor_ms": sr["slowest_ms"], "stage_ms": stage_ms.get("chain"),
                                               "note": "the stage = the island kernel (330 ms for 20 000 reads at six wavefronts per CU: more of them and k_chain_serial's wavefront, which needs half a "
                                                       "SIMD's registers, cannot start beside it) + what k_chain_serial still has to do when it ends (the seed-richest read with equal keys has to go "
                                                       "through both: floor_ms is its serial part; %d such reads of %.0f ms on average in this chunk) + the seed filter's local SW (k_seed_sw_reg, ~4.7 ms per "
                                                       "1000 reads, VALU-bound) + k_chain_finish; profiles/r05zzz_timeline_ont2d.tsv"
                                                       % (chain_kernel["reads_chained_serially_equal_keys"], sr["ms_per_read"])}
            out["chain_kernel"] = chain_kernel
        # (stderr: the driver keeps the tail of it; the JSON line on stdout is long enough to lose its head there)
        log("hot path %.2f ms per step = %.2f M reads/s (%s); %s %.2f ms per launch = %.0f GB/s algorithmic = %.3f of the HBM peak"
            % (dt / steps * 1e3, value / 1e6, " / ".join("%s %.1f" % (k, v) for k, v in stage_ms.items()), roof_kernel, bwd_ms, ach, ach / HBM_PEAK_GBS))
        ref_run = None
        if world == 1 and not a.no_parity and time_left() < 150:
            out["parity"] = {"skipped": "time budget (%.0f s of %.0f s left)" % (time_left(), a.budget_s)}
        elif world == 1 and not a.no_parity:
            regs, reg_off = ctx.batch_download()
            n_s = min(a.parity_reads, n_reads)
            if not ont or n_s >= 512:
                n_s -= n_s % 512
            try:
                out["parity"] = parity_gate(ctx, bm2, prefix, a.workdir, seqs, regs, reg_off, opt, opt_args, paired, max(n_s, 2), a.workload,
                                            n_regs=min(n_s, a.parity_regs_reads) if ont else None)
                ref_run = out["parity"].pop("_reference_run", None)
                if ont and "regs_equal" in out["parity"]:
                    # which of the gated reads took the rare paths: the prefix once more as a chunk of its own (its regs must not depend on what else
                    # is in the batch) -- the island kernel counts the reads it chained by islands and those it had to chain serially (equal chain keys)
                    c3 = bm2.Context(share=ctx)
                    try:
                        from tools import refio as _refio
                        e3, o3, l3 = _refio.pack_reads(seqs[:n_s])
                        c3.batch_upload(e3, o3, l3); c3.batch_run(opt)
                        r3, ro3 = c3.batch_download()
                        cn3 = np.asarray(c3.batch_fetch("counters", np.uint64), np.float64)
                        same = bool(regs_records(r3, ro3, 0, n_s).tobytes() == regs_records(regs, reg_off, 0, n_s).tobytes())
                        out["parity"]["gated_reads_chained_serially_equal_keys"] = int(cn3[16])
                        out["parity"]["gated_reads_chained_by_islands"] = int(cn3[17])
                        out["parity"]["prefix_alone_equals_prefix_in_chunk"] = same
                        if not same:
                            rc = 3
                    finally:
                        c3.close()
                if not (out["parity"].get("regs_equal") and out["parity"].get("sam_equal") and out["parity"].get("fin_equal")):
                    rc = 3
            except Exception as e:                                                    # noqa  (the line is still printed: the gate did not run to its end)
                out["parity"] = {"error": "the gate raised: %s" % e, "regs_equal": None, "fin_equal": None, "sam_equal": None}
                rc = 3
        else:
            out["parity"] = None
        if world == 1 and not a.no_cpu_baseline and time_left() < 120:
            out["cpu_baseline"] = {"skipped": "time budget (%.0f s of %.0f s left)" % (time_left(), a.budget_s)}
        elif world == 1 and not a.no_cpu_baseline:           # the reference on this host's cores: at N=1 only (the other ranks would idle)
            try:
                t = time.time()
                if ont and ref_run is not None and (out.get("parity") or {}).get("reads", 0) >= 300:
                    # the gate's `bwa-mem2 mem -x ont2d` run over its reads IS a timed run of the reference on this host's CPUs: no second one
                    cb = cpu_baseline_from(ref_run[0], ref_run[1], ref_run[2], "%d ONT-like reads (the first of the timed chunk: the parity gate's reference run)" % out["parity"]["reads"], opt_args)
                elif ont:
                    nb = min(len(seqs), 300)
                    f1 = os.path.join(a.workdir, "cpu_ont.fq")
                    synth.write_fastq(f1, seqs[:nb])
                    cb = cpu_baseline(prefix, [f1], "%d ONT-like reads (the first of the timed chunk)" % nb, opt_args)
                else:
                    c1, c2 = synth.make_reads_pe(seed + 5, contigs(), a.cpu_pairs, L=a.read_len)
                    f1, f2 = os.path.join(a.workdir, "cpu_1.fq"), os.path.join(a.workdir, "cpu_2.fq")
                    synth.write_fastq(f1, c1, suffix="/1"); synth.write_fastq(f2, c2, suffix="/2")
                    ref_sam = os.path.join(a.workdir, "cpu_ref.sam") if not a.no_parity else "/dev/null"
                    cb = cpu_baseline(prefix, [f1, f2], "%d x %d bp PE reads" % (2 * a.cpu_pairs, a.read_len), out=ref_sam)
                    log("cpu baseline took %.1fs" % (time.time() - t))
                    if cb and not a.no_parity and isinstance(out.get("parity"), dict) and "regs_equal" in out["parity"]:
                        # the wide gate: what the baseline run already paid for (all its records), then the same first mates as single-end reads
                        try:
                            g = sam_gate(ctx, bm2, [f1, f2], ref_sam, opt, True, "pe-wide")
                            out["parity"]["wide_pe"] = g
                            out["parity"]["sam_records"] = out["parity"].get("sam_records", 0) + g["sam_records"]
                            ok = g["sam_equal"]
                            if time_left() > 200:
                                se_sam = os.path.join(a.workdir, "cpu_ref_se.sam")
                                n_se = min(a.cpu_pairs, 100000)
                                f_se = os.path.join(a.workdir, "cpu_se.fq")
                                synth.write_fastq(f_se, c1[:n_se])
                                err_se, _, _ = run_reference_mem(prefix, [f_se], out=se_sam)
                                if err_se is not None:
                                    g2 = sam_gate(ctx, bm2, [f_se], se_sam, opt, False, "se-wide")
                                    out["parity"]["wide_se"] = g2
                                    out["parity"]["sam_records"] += g2["sam_records"]
                                    ok = ok and g2["sam_equal"]
                            out["parity"]["sam_equal"] = bool(out["parity"].get("sam_equal") and ok)
                            if not ok:
                                rc = 3
                        except Exception as e:                                        # noqa
                            out["parity"]["wide_error"] = str(e)
                            rc = 3
                if ont:
                    log("cpu baseline took %.1fs" % (time.time() - t))
            except Exception as e:                                                    # noqa
                cb = {"error": str(e)}
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        if world == 1 and not a.no_e2e and not ont and time_left() < 180:
            out["end_to_end"] = {"skipped": "time budget (%.0f s of %.0f s left)" % (time_left(), a.budget_s)}
        elif world == 1 and not a.no_e2e and not ont:
            t = time.time()
            texts = []                                       # distinct chunks of the same shape as the timed one, generated side by side
            meta = prefix + ".contigs.npz"
            procs = []
            for i in range(a.e2e_chunks):
                fa, fb = os.path.join(a.workdir, "e2e_%d_1.fq" % i), os.path.join(a.workdir, "e2e_%d_2.fq" % i)
                procs.append((fa, fb, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gen_chunk.py"), meta, str(seed + 100 + i),
                                                        str(n_reads // 2), str(a.read_len), fa, fb, "c%d_" % i])))
            gen_failed = 0
            for fa, fb, pr in procs:
                try:
                    if pr.wait(timeout=max(30.0, time_left() - 120)) != 0:
                        raise RuntimeError("generator exit code %d" % pr.returncode)
                    texts.append((open(fa, "rb").read(), open(fb, "rb").read()))
                    if a.no_binding:                                 # (the chunks' files stay for the drop-in timing below)
                        os.remove(fa); os.remove(fb)
                except Exception as e:                                                    # noqa  (the leg runs on the chunks that exist)
                    gen_failed += 1
                    log("end-to-end input: a chunk was not generated: %s" % e)
                    if pr.poll() is None:
                        pr.kill()
            log("end-to-end input: %d chunks generated in %.1fs" % (len(texts), time.time() - t))
            if not texts:
                out["end_to_end"] = {"error": "no input chunk could be generated"}
            else:
                # ONE attempt: a stage that fails or hangs leaves its threads behind (they may still be inside a library call on a context, which is
                # not thread-safe), so nothing else is run on these contexts afterwards -- the line is printed and the process leaves through os._exit
                try:
                    out["end_to_end"] = end_to_end(ctx, bm2, texts * max(a.e2e_rounds, 1), opt, True, 0, limit_s=max(60.0, min(420.0, time_left() - 30)))
                    out["end_to_end"]["frac_of_hot_path"] = out["end_to_end"]["value"] / value
                    out["value_end_to_end"] = out["end_to_end"]["value"]
                    if (out["end_to_end"].get("chunk_check") or {}).get("equal_to_serial_run") is False:
                        log("end-to-end leg: the text of chunk %d differs from the serial run's" % out["end_to_end"]["chunk_check"]["chunk"])
                        rc = 3
                except Exception as e:                                                # noqa  (TimeoutError: a stage is stuck)
                    out["end_to_end"] = {"error": str(e)}
                    hung = True
                    log("end-to-end leg failed: %s" % e)
        else:
            out["end_to_end"] = None
        # the literal drop-in: the reference's own binary with libbm2 linked in place of mem_process_seqs (oracle/_ref/bwa-mem2.bm2), from FASTQ
        # files to a SAM file, beside the unmodified binary on the same files and threads
        if world == 1 and not ont and not a.no_e2e and not a.no_binding and not hung and time_left() > 240:
            try:
                # (the unmodified binary on ALL the chunks when the budget allows its ~6 s per chunk: then every record of the binding's output is compared)
                out["binding"] = binding_leg(a.workdir, prefix, n_chunks=a.e2e_chunks, n_ref_chunks=a.e2e_chunks if time_left() > 700 else 2)
            except Exception as e:                                                    # noqa
                out["binding"] = {"error": str(e)}
        for fn in os.listdir(a.workdir):                             # (chunk files the drop-in timing did not get to)
            if re.match(r"(e2e_\d+_[12]\.fq|bind_.*)$", fn):
                try:
                    os.remove(os.path.join(a.workdir, fn))
                except OSError:
                    pass
        # BASELINE configs 5 and 2 as workloads of their own, in the same line: each with its parity gate, its kernels' figures and the compiled
        # reference timed beside it on this host
        if world == 1 and not ont and not a.no_side_workloads and not hung:
            # The other configs run as processes of their own on this GPU: this process gives its contexts back first (the index replica and the chunk's
            # workspaces: config 5's chunk needs the room) -- and its hardware queues: beside a process that merely HOLDS contexts the small device batches
            # of `bwa-mem2.bm2s1` took 8.3-8.7 s per million reads in rounds 5's calls, 6.45 s with the GPU to itself (profiles/r05e, r05_, r05z_)
            try:
                ctx.close()
                ctx = None
            except Exception as e:                                                    # noqa
                log("closing the main context before the side workloads: %s" % e)
            if early_side is not None:                           # (run at the start of this process, the GPU to themselves)
                for key, val in early_side.items():
                    out[key] = val
                    if isinstance(val, dict) and val.get("exit_code") not in (0, None):
                        rc = rc or 3
                    if key == "config2" and isinstance(val, dict) and (val.get("s1_binding") or {}).get("sam_equal") is False:
                        rc = rc or 3
            else:
                for key, val in run_side_workloads(a, early_s1, time_left).items():
                    out[key] = val
                    if isinstance(val, dict) and val.get("exit_code") not in (0, None):
                        rc = rc or 3
        try:                                                     # the legs once more, one line each, at the very end of stderr
            pr = out.get("parity") or {}
            if "regs_equal" in pr:
                log("parity gate: regs %s, a19 %s, SAM %s (%s records; %s regs of the %d-read prefix, %s beyond 2^32)"
                    % (pr.get("regs_equal"), pr.get("fin_equal"), pr.get("sam_equal"), pr.get("sam_records"), pr.get("regs"), pr.get("reads", 0), pr.get("regs_over_2p32")))
            e2 = out.get("end_to_end") or {}
            if e2.get("value"):
                log("end_to_end (FASTQ text -> SAM text): %.2f M reads/s over %d chunks = %.2f of the hot path (steady state %s M reads/s); last chunk equal to a serial run: %s"
                    % (e2["value"] / 1e6, e2.get("chunks", 0), e2.get("frac_of_hot_path", 0.0),
                       ("%.2f" % (e2["steady_state"]["reads_per_s"] / 1e6)) if isinstance(e2.get("steady_state"), dict) and e2["steady_state"].get("reads_per_s") else "n/a",
                       (e2.get("chunk_check") or {}).get("equal_to_serial_run")))
            cbl = out.get("cpu_baseline") or {}
            if cbl.get("value"):
                log("cpu baseline (%s, %s threads): %.3f M reads/s whole mem, %s like-for-like" % (cbl.get("kind"), cbl.get("cores"), cbl["value"] / 1e6,
                    ("%.3f M" % (cbl["hot_path_value"] / 1e6)) if cbl.get("hot_path_value") else "n/a"))
            bd = out.get("binding") or {}
            if bd.get("bm2_wall_s"):
                log("binding (bwa-mem2.bm2 mem, %d reads from files): %.1f s wall, %.2f M reads/s in steady chunks; SAM equal to the reference's: %s (%s records compared)"
                    % (bd.get("reads", 0), bd["bm2_wall_s"], (bd.get("reads_per_s_bm2_steady_chunks") or 0.0) / 1e6, bd.get("sam_equal"), bd.get("sam_records_compared")))
            c5 = out.get("config5") or {}
            if c5.get("value"):
                log("config 5 (%s reads of mean 10 kb per step, -x ont2d): %.0f reads/s, %.0f ms per step (%s); gate %s; reference on this host %s reads/s"
                    % ((c5.get("config") or {}).get("reads_per_gpu_per_step"), c5["value"], c5.get("ms_per_step", 0.0),
                       " / ".join("%s %.0f" % (k, v) for k, v in (c5.get("stage_ms_per_step") or {}).items()),
                       {k: (c5.get("parity") or {}).get(k) for k in ("regs_equal", "fin_equal", "sam_equal", "reads")}, (c5.get("cpu_baseline") or {}).get("value")))
            c2 = out.get("config2") or {}
            if c2.get("value"):
                s1 = c2.get("s1_binding") or {}
                log("config 2 (S1 alone, %s pairs resident): %s G cells/s; bwa-mem2.bm2s1 mem %s s per chunk against the unmodified binary's %s s, SAM equal: %s"
                    % ((c2.get("config") or {}).get("pairs_per_gpu_per_step"), ("%.0f" % ((c2.get("extend_kernel") or {}).get("gcups") or 0.0)),
                       (s1.get("bm2s1") or {}).get("chunk_real_s"), (s1.get("reference") or {}).get("chunk_real_s"), s1.get("sam_equal")))
        except Exception as e:                                                        # noqa  (a summary must not cost the line)
            log("summary lines: %s" % e)
        bench_line.emit(out, a.full_json)                       # the full record to a file, the compact line LAST on stdout
    if hung:                                                     # stage threads of a failed end-to-end attempt may be left: do not join them
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(rc)
    if ctx is not None:
        ctx.close()
    dist_util.finish(world)
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
