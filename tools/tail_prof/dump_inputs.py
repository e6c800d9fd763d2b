#!/usr/bin/env python3
"""Writes the inputs of the host tail (reads, names, quals, alnregs of tests/host_tail_bench.py's cached batch) as one flat binary file for
tools/tail_prof/harness.cpp, so the tail can be profiled (gprof) and timed without Python or a GPU.
    python tools/tail_prof/dump_inputs.py <pairs> <out.bin>      (run tests/host_tail_bench.py <pairs> once before: it makes the cache)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bm2
    from helpers import oracle_finish_regs
    from tools import synth
    n_pairs, out = int(sys.argv[1]), sys.argv[2]
    like_bench = os.environ.get("BM2_TAILBENCH_GENOME") == "bench"     # repeat content (~6 %) and read errors of bench.py's workload, 20 Mbp
    d = "/tmp/bm2_host_tail_bench_%d%s" % (n_pairs, "_b" if like_bench else "")
    fa = os.path.join(d, "g.fa")
    if like_bench:
        names, ctg, alts = synth.make_genome(301, [12000000, 8000000], n_repeat_families=8, repeat_len=(300, 6000), copies=(5, 100), divergence=(0.01, 0.15))
    else:
        names, ctg, alts = synth.make_genome(301, [2000000, 1000000], n_repeat_families=20, repeat_len=(200, 3000), copies=(3, 60), divergence=(0.0, 0.08))
    if like_bench:
        r1, r2 = synth.make_reads_pe(302, ctg, n_pairs, L=150)
    else:
        r1, r2 = synth.make_reads_pe(302, ctg, n_pairs, L=150, sub_rate=0.01, indel_frac=0.1, random_frac=0.005)
    seqs = [x for p in zip(r1, r2) for x in p]
    enc = np.concatenate(seqs).astype(np.uint8)
    ln = np.array([len(s) for s in seqs], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1])]).astype(np.int64)
    z = np.load(os.path.join(d, "regs.npz"))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, bm2.default_opt(), z["regs"], z["reg_off"])
    aln = np.ascontiguousarray(aln, bm2.ALNREG_DT)
    aln_off = np.ascontiguousarray(aln_off, np.int64)
    with open(out, "wb") as f:
        f.write(np.array([len(ln), len(enc), len(aln)], np.int64).tobytes())
        f.write(enc.tobytes()); f.write(off.tobytes()); f.write(ln.tobytes())
        f.write(aln.tobytes()); f.write(aln_off.tobytes())
    open(out + ".prefix", "w").write(fa)
    print("wrote %s: %d reads, %d hits" % (out, len(ln), len(aln)))


if __name__ == "__main__":
    main()
