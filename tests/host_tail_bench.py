#!/usr/bin/env python3
"""Times the host tail (bm2_finish_regs, bm2_sam_se, bm2_sam_pe) on a synthetic paired-end batch whose regs come from the
CPU oracle (cached under /tmp), so the tail can be tuned without a GPU.  python tests/host_tail_bench.py [pairs] [threads...]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bm2
    if os.environ.get("BM2_BENCH_LIB"):
        bm2.LIB_PATH = os.environ["BM2_BENCH_LIB"]         # e.g. an instrumented build of the same sources
    from helpers import oracle_finish_regs, ref_binary
    from tools import oracle, synth
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    threads = [int(x) for x in sys.argv[2:]] or [1, os.cpu_count()]
    like_bench = os.environ.get("BM2_TAILBENCH_GENOME") == "bench"     # repeat content (~6 %) and read errors of bench.py's workload, 20 Mbp
    d = "/tmp/bm2_host_tail_bench_%d%s" % (n_pairs, "_b" if like_bench else "")
    os.makedirs(d, exist_ok=True)
    fa = os.path.join(d, "g.fa")
    if like_bench:
        names, ctg, alts = synth.make_genome(301, [12000000, 8000000], n_repeat_families=8, repeat_len=(300, 6000), copies=(5, 100), divergence=(0.01, 0.15))
    else:
        names, ctg, alts = synth.make_genome(301, [2000000, 1000000], n_repeat_families=20, repeat_len=(200, 3000), copies=(3, 60), divergence=(0.0, 0.08))
    if not os.path.exists(fa + ".bwt.2bit.64"):
        synth.write_fasta(fa, names, ctg)
        subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if like_bench:
        r1, r2 = synth.make_reads_pe(302, ctg, n_pairs, L=150)
    else:
        r1, r2 = synth.make_reads_pe(302, ctg, n_pairs, L=150, sub_rate=0.01, indel_frac=0.1, random_frac=0.005)
    seqs = [x for p in zip(r1, r2) for x in p]
    enc = np.concatenate(seqs)
    ln = np.array([len(s) for s in seqs], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1])]).astype(np.int64)
    opt = bm2.default_opt()
    cache = os.path.join(d, "regs.npz")
    if os.path.exists(cache):
        z = np.load(cache)
        regs, reg_off = z["regs"], z["reg_off"]
    else:
        ix = oracle.Index(fa)
        t0 = time.time()
        prg = ix.run(enc, off, ln)["REGPRG"]
        print("oracle: %.1f s" % (time.time() - t0))
        regs = np.zeros(len(prg), bm2.REG_DT)
        for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
            regs[f] = prg[f]
        ro = np.zeros(len(ln) + 1, np.int64)
        np.add.at(ro, prg["read"] + 1, 1)
        reg_off = np.cumsum(ro)
        np.savez(cache, regs=regs, reg_off=reg_off)
    rnames = [b"p%d" % (i // 2) for i in range(len(seqs))]
    quals = [b"F" * len(s) for s in seqs]
    n = len(seqs)
    for th in threads:
        so = bm2.default_sam_opt(n_threads=th)
        t0 = time.time(); aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off); t1 = time.time()
        se = bm2.sam_se(fa, enc, off, ln, opt, aln, aln_off, rnames, quals, None, so); t2 = time.time()
        pe, _ = bm2.sam_pe(fa, enc, off, ln, opt, aln, aln_off, rnames, quals, None, so); t3 = time.time()
        print("rescue alignments (planned, used, missed):", bm2.sam_rescue_stats(), " CIGAR alignments:", bm2.sam_cigar_stats())
        print("threads %3d: finish_regs %7.0f reads/s | sam_se %7.0f reads/s | sam_pe %7.0f reads/s   (%d reads, %d regs, %.1f MB SAM)"
              % (th, n / (t1 - t0), n / (t2 - t1), n / (t3 - t2), n, len(aln), len(pe) / 1e6))


if __name__ == "__main__":
    main()
