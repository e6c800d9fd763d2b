#!/usr/bin/env python3
"""Launch-policy sweep on the bench workload, in ONE process (no torch): every knob of bm2_knob() (bm2_ctx.h) is a launch-policy
setting that cannot change a result, so each candidate is timed on the resident chunk and its regs are checksummed against the
default's.  Coordinate descent: one knob at a time, the best value is kept.  Writes <out>/sweep.json (every candidate: ms per step,
stage ms, checksum) and <out>/best_env.sh (export lines for the profiling passes that follow in tools/gpu/run_prof.sh).

    python tools/gpu/sweep.py <out_dir> [--steps 3] [--genome-mbp 3100] [--reads 1000000]
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

GRID = [   # (name, candidates): a candidate is a dict of knobs set together; the first one ({} = the library's defaults) is the incumbent
    ("extension: rows in registers (lane_dp8r) from this class up", [{}, {"BM2_EXT_REG_QMIN": 0}, {"BM2_EXT_REG_QMIN": 112}, {"BM2_EXT_REG_QMIN": 96}, {"BM2_EXT_REG_QMIN": 80}]),
    ("seeding: workgroups per CU of the wavefront-per-task kernel beside k_bwd", [{}, {"BM2_BWD_HEAVY_WG": 4}, {"BM2_BWD_HEAVY_WG": 8}, {"BM2_BWD_HEAVY_WG": 16}]),
    ("seeding: k_bwd hands old tasks over (k_bwd_cont: sixteen lanes per task)", [{}, {"BM2_BWD_EXPORT_AGE": 0}, {"BM2_BWD_EXPORT_AGE": 192}, {"BM2_BWD_EXPORT_AGE": 160}]),
    ("seeding: workgroups per CU of k_bwd_cont", [{}, {"BM2_BWD_CONT_BPC": 4}, {"BM2_BWD_CONT_BPC": 8}]),
    ("chaining: the light reads in plain order (4) instead of 2x classes of seed count (5, the default)", [{}, {"BM2_PERM_MODE": 4}]),
    ("chaining: k_chain_finish's part by k_chain's lanes + one wavefront per seed-rich read (default) / by the lane-per-read kernel", [{}, {"BM2_CHAIN_FUSE_FINISH": 0}, {"BM2_CHAIN_FINISH_WAVE": 0}, {"BM2_CHAIN_FUSE_FINISH": 0, "BM2_CHAIN_FINISH_WAVE": 0}, {"BM2_CHAIN_FUSE_FINISH": 0, "BM2_CHAIN_FINISH_WAVE": 0, "BM2_CHAIN_FINISH_PERM": 0}]),
    ("chaining: k_chain on the main stream (0) instead of a side stream of its own (1, the default)", [{}, {"BM2_CHAIN_MAIN_SIDE": 0}, {"BM2_CHAIN_MAIN_SIDE": 0, "BM2_HEAVY_SA": 80}, {"BM2_HEAVY_SA": 80}, {"BM2_HEAVY_SA": 120}]),
    ("chaining AB: the tiers' knobs, the default measured between the candidates", [{}, {"BM2_CHAIN_HEAVY_WPE": 2}, {"BM2_PERM_MODE": 5}, {"BM2_CHAIN_WAVES_PER_CU": 8}, {"BM2_HEAVY_SA": 80}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_COOP_FLT": 1}, {"BM2_PERM_MODE": 4}, {"BM2_CHAIN_MAIN_SIDE": 1}, {"BM2_CHAIN_HEAVY_WPE": 2, "BM2_CHAIN_WAVES_PER_CU": 8}, {"BM2_CHAIN_STAGE": 1}, {"BM2_CHAIN_HEAVY_WPE": 2, "BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_FUSE_FINISH": 1}, {"BM2_CHAIN_HEAVY_WPE": 2, "BM2_PERM_MODE": 4}]),
    ("chaining AB2: reads beyond 512 seeds to the island kernel, the default measured between", [{}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_COOP_FLT": 1}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_STAGE": 1}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_PERM_MODE": 5}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_HEAVY_SA": 80}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_MAIN_SIDE": 1}, {"BM2_CHAIN_TIER_MAX": 256}, {"BM2_CHAIN_FUSE_FINISH": 1}, {"BM2_CHAIN_TIER_MAX": 256}]),
    ("AB3: seeding / extension knobs, the default measured between", [{}, {"BM2_BWD_HEAVY_WG": 8}, {"BM2_PERM_MODE": 5}, {"BM2_BWD_EXPORT_AGE": 160}, {"BM2_HEAVY_SA": 80}, {"BM2_BWD_HEAVY_WG": 8}, {"BM2_CHAIN_STAGE": 1}, {"BM2_BWD_EXPORT_AGE": 160}, {"BM2_CHAIN_COOP_FLT": 1}, {"BM2_EXT_PEND_DIV": 6}, {"BM2_CHAIN_MAIN_SIDE": 1}, {"BM2_EXT_PEND_DIV": 6}, {"BM2_CHAIN_FUSE_FINISH": 1}, {"BM2_BWD_HEAVY_WG": 8, "BM2_BWD_EXPORT_AGE": 160}, {"BM2_CHAIN_FINISH_WAVE": 1}, {"BM2_BWD_HEAVY_WG": 8, "BM2_BWD_EXPORT_AGE": 160}]),
    ("chain clock", [{}, {"BM2_CHAIN_CLOCK": 1}]),
    ("chain staging", [{}, {"BM2_CHAIN_STAGE": 1}]),
    ("chain heavy threshold", [{}, {"BM2_HEAVY_SA": 100}, {"BM2_HEAVY_SA": 72}, {"BM2_HEAVY_SA": 64}]),
    ("chain: seed-rich reads to the island kernel", [{}, {"BM2_CHAIN_TIER_MAX": 512}, {"BM2_CHAIN_TIER_MAX": 256}, {"BM2_CHAIN_TIER_MAX": 128}, {"BM2_CHAIN_TIER_MAX": 256, "BM2_HEAVY_SA": 160}]),
    ("chain waves per CU", [{}, {"BM2_CHAIN_WAVES_PER_CU": 32}, {"BM2_CHAIN_WAVES_PER_CU": 8}]),
    ("k_bwd LDS survivors / blocks per CU / waves per SIMD", [{}, {"BM2_BWD_LCAP": 8, "BM2_BWD_BLOCKS_PER_CU": 4}, {"BM2_BWD_LCAP": 6, "BM2_BWD_BLOCKS_PER_CU": 4},
                                                              {"BM2_BWD_LCAP": 4, "BM2_BWD_BLOCKS_PER_CU": 4},
                                                              {"BM2_BWD_LCAP": 6, "BM2_BWD_BLOCKS_PER_CU": 5, "BM2_BWD_WAVES": 5},
                                                              {"BM2_BWD_LCAP": 4, "BM2_BWD_BLOCKS_PER_CU": 5, "BM2_BWD_WAVES": 5}]),
    ("extension launches on distinct hardware queues", [{}, {"BM2_EXT_QUEUE_MAP": 0}]),
    ("extension scores by byte permute", [{}, {"BM2_EXT_PERM_SCORES": 0}]),
    ("extension wave classes", [{}, {"BM2_EXT_WAVE_QMIN": 129}, {"BM2_EXT_WAVE_QMIN": 145}, {"BM2_EXT_WAVE_QMIN": 161}, {"BM2_EXT_WAVE_QMIN": 97}]),
    ("chain: mem_chain_flt's kept-chain walk by the whole wavefront", [{}, {"BM2_CHAIN_COOP_FLT": 0}]),
    ("chain: wavefronts per SIMD the heavy reads' kernel is allocated for", [{}, {"BM2_CHAIN_HEAVY_WPE": 2}, {"BM2_CHAIN_HEAVY_WPE": 4}]),
    ("extension rounds", [{}, {"BM2_EXT_ROUNDS": 2}, {"BM2_EXT_ROUNDS": 1}, {"BM2_EXT_PEND_DIV": 6}, {"BM2_EXT_PEND_DIV": 24}]),
    ("extension dispatch order", [{}, {"BM2_EXT_REVERSE": 1}]),
    ("seeding pass 3 placement", [{}, {"BM2_P3_AT": 0}, {"BM2_P3_AT": 2}]),
    ("seeding: pass 3 workgroups per CU", [{}, {"BM2_P3_BPC": 2}, {"BM2_P3_BPC": 1}, {"BM2_P3_AT": 0, "BM2_P3_BPC": 1}, {"BM2_P3_AT": 0, "BM2_P3_BPC": 2}, {"BM2_P3_AT": 0, "BM2_P3_BPC": 3}]),
    ("SA lookup by quads", [{}, {"BM2_SAL_QUAD": 1}]),
    ("walk blocks per CU", [{}, {"BM2_WALK_BLOCKS_PER_CU": 6}, {"BM2_WALK_BLOCKS_PER_CU": 3}]),
    ("purge threshold", [{}, {"BM2_PF_HEAVY": 48}, {"BM2_PF_HEAVY": 12}]),
    ("sub-batches of the chunk on their own streams", [{}, {"BM2_N_SUB": 2}, {"BM2_N_SUB": 3}]),      # (last: the parts stay in place once made)
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--genome-mbp", type=int, default=3100)
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workdir", default=os.environ.get("BM2_BENCH_WORKDIR", "/tmp/bm2_bench"))
    ap.add_argument("--quick", action="store_true", help="two candidates per knob (a test of this script on the emulator)")
    ap.add_argument("--budget-s", type=float, default=90.0, help="stop sweeping (keep what is best so far) after this many seconds")
    ap.add_argument("--only", default="", help="comma-separated substrings: sweep only the knobs whose name contains one of them")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    import bench
    import bm2
    emu = os.environ.get("BM2_EMU_LIB")
    if emu:
        bm2.LIB_PATH = emu
    seed = 20260924
    os.makedirs(a.workdir, exist_ok=True)
    t = time.time()
    prefix, contigs = bench.prepare_genome(a.workdir, a.genome_mbp, seed)
    seqs = bench.pe_chunk(a.workdir, contigs, seed, a.reads, a.read_len)
    n = len(seqs)
    print("[sweep] workload ready in %.1fs" % (time.time() - t), file=sys.stderr, flush=True)
    t = time.time()
    ctx = bm2.Context(0, prefix)
    ctx.batch_upload(seqs.reshape(-1), np.arange(n, dtype=np.int64) * a.read_len, np.full(n, a.read_len, np.int32))
    opt = bm2.default_opt()
    print("[sweep] index + chunk resident after %.1fs" % (time.time() - t), file=sys.stderr, flush=True)

    state = {"nsub": None}

    def measure(env):
        for k in [k for k in os.environ if k.startswith("BM2_") and k not in ("BM2_EMU_LIB", "BM2_BENCH_WORKDIR")]:
            del os.environ[k]
        for k, v in env.items():
            os.environ[k] = str(v)
        if env.get("BM2_N_SUB") != state["nsub"]:                  # the chunk is cut into sub-batches when it is uploaded
            ctx.batch_upload(seqs.reshape(-1), np.arange(n, dtype=np.int64) * a.read_len, np.full(n, a.read_len, np.int32))
            state["nsub"] = env.get("BM2_N_SUB")
        ctx.batch_run(opt)                                   # warm-up (workspace sizes of this setting)
        kms = {}
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.batch_run(opt)
            for name, ms in ctx.batch_kernel_ms():
                kms[name.split(".")[0]] = kms.get(name.split(".")[0], 0.0) + ms / a.steps
                if "." in name:
                    kms[name] = kms.get(name, 0.0) + ms / a.steps
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        regs, reg_off = ctx.batch_download()
        crc = zlib.crc32(reg_off.tobytes(), zlib.crc32(regs.tobytes()))
        sc = ctx.batch_fetch("seed_counters", np.uint64)
        if len(sc) >= 27:                                        # tasks k_bwd handed over in pass 1 / 2 (exact counts) and the rows their continuations walked
            kms["handed_over"] = [int(sc[21]), int(sc[22]), int(sc[25]), int(sc[26])]
        return ms, kms, crc

    log = []
    best_env = {}
    t_start = time.time()
    base_ms, base_k, base_crc = measure(best_env)
    log.append({"env": {}, "ms": base_ms, "stages": base_k, "crc": base_crc})
    best_ms = base_ms
    print("[sweep] default: %.2f ms/step %s" % (base_ms, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in base_k.items()}), file=sys.stderr, flush=True)
    only = [x for x in a.only.split(",") if x]
    for name, cands in GRID:
        if only and not any(x in name for x in only):
            continue
        if a.quick:
            cands = cands[:2]
        if time.time() - t_start > a.budget_s:
            print("[sweep] time budget used; stopping at", name, file=sys.stderr, flush=True)
            break
        pick = {}
        for cand in cands[1:]:
            env = dict(best_env); env.update(cand)
            try:
                ms, k, crc = measure(env)
            except Exception as e:                                        # noqa
                log.append({"env": env, "error": str(e)})
                print("[sweep] %s %s: %s" % (name, cand, e), file=sys.stderr, flush=True)
                continue
            ok = crc == base_crc
            log.append({"env": env, "ms": ms, "stages": k, "crc": crc, "same_regs": ok})
            print("[sweep] %s %s: %.2f ms/step %s%s" % (name, cand, ms, {x: (round(y, 2) if isinstance(y, float) else y) for x, y in k.items()}, "" if ok else "  REGS DIFFER"),
                  file=sys.stderr, flush=True)
            if ok and ms < best_ms * 0.985:                               # keep a change only if it buys more than the noise
                best_ms, pick = ms, cand
        best_env.update(pick)
    final_ms, final_k, final_crc = measure(best_env)
    out = {"workload": {"genome_mbp": a.genome_mbp, "reads": n, "read_len": a.read_len, "steps": a.steps},
           "default": {"ms": base_ms, "stages": base_k}, "best_env": best_env, "best": {"ms": final_ms, "stages": final_k},
           "same_regs": final_crc == base_crc, "log": log}
    json.dump(out, open(os.path.join(a.out, "sweep.json"), "w"), indent=1)
    with open(os.path.join(a.out, "best_env.sh"), "w") as f:
        if final_crc == base_crc:
            for k, v in best_env.items():
                f.write("export %s=%s\n" % (k, v))
    print("[sweep] best %s: %.2f ms/step (default %.2f)" % (best_env, final_ms, base_ms), file=sys.stderr, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
