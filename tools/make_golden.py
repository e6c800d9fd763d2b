#!/usr/bin/env python3
"""Regenerate tests/golden/*: small index + reads + stage dumps made by the COMPILED REFERENCE
(oracle/_ref/bwa-mem2.* for `index`, oracle/_ref/refdump.* for the hot-path stages).

Run only where /root/reference exists (the build container):  python tools/make_golden.py
The fixtures pin the oracle restatement (tests/test_oracle_golden.py) and, on the GPU box, the HIP path.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refio, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def arch():
    flags = open("/proc/cpuinfo").read()
    return "avx512bw" if "avx512bw" in flags else "avx2" if "avx2" in flags else "sse41"


def run(cmd, **kw):
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


def make(name, seed, contigs, n_reads, read_len, genome_kw, reads_kw, opts=(), long_reads=None):
    os.makedirs(GOLD, exist_ok=True)
    pre = os.path.join(GOLD, name)
    names, ctg, alts = synth.make_genome(seed, contigs, **genome_kw)
    synth.write_fasta(pre + ".fa", names, ctg)
    if alts:
        synth.write_alt(pre + ".fa.alt", alts)
    run([os.path.join(REF, "bwa-mem2." + arch()), "index", pre + ".fa"])
    if long_reads:
        seqs = synth.make_reads_long(seed + 1, ctg, n_reads, **long_reads)
    else:
        reads = synth.make_reads_se(seed + 1, ctg, n_reads, L=read_len, **reads_kw)
        # a few hand-made edge cases appended: all-N read, read shorter than the minimum seed, exact repeat unit
        extra = [np.full(read_len, 4, np.uint8), reads[0][:12].copy(), reads[1][:30].copy()]
        seqs = [r for r in reads] + extra
    with open(pre + ".reads.txt", "w") as f:
        for s in seqs:
            f.write("".join("ACGTN"[c] for c in s) + "\n")
    run([os.path.join(REF, "refdump." + arch())] + list(opts) + [pre + ".fa", pre + ".reads.txt", pre + ".dump"])
    d = refio.read_dump(pre + ".dump")
    np.savez_compressed(pre + ".dump.npz", **d)
    os.remove(pre + ".dump")
    os.remove(pre + ".fa")           # the index files carry everything the tests need
    for ext in (".fa.amb",):
        pass
    print(name, {k: len(v) for k, v in d.items()}, "max chains/read",
          int(np.bincount(d["CHN0"]["read"]).max()) if len(d["CHN0"]) else 0)


if __name__ == "__main__":
    make("g60k", 101, [30000, 20000, 10000], 1100, 150,
         dict(n_repeat_families=5, repeat_len=(200, 1500), copies=(4, 30), divergence=(0.0, 0.06), n_gaps=1,
              gap_len=(30, 120), alt_contigs=1, alt_len=4000),
         dict())
    # ONT-like long reads with the `-x ont2d` preset: exercises mem_flt_chained_seeds / mem_seed_sw (bwamem.cpp:401-504),
    # min_chain_weight (incl. the a_[0] quirk of mem_chain_flt), int16/int32-class extensions of kb-long queries
    make("g40k_ont", 303, [25000, 15000], 14, 0,
         dict(n_repeat_families=3, repeat_len=(200, 1500), copies=(3, 10), divergence=(0.0, 0.06), n_gaps=0, alt_contigs=0),
         dict(), opts=("-x", "ont2d"), long_reads=dict(mean_len=1600, max_len=4000))
    make("g20k_l76", 202, [12000, 8000], 300, 76,
         dict(n_repeat_families=3, repeat_len=(100, 400), copies=(3, 12), divergence=(0.0, 0.03), n_gaps=0, alt_contigs=0),
         dict(sub_rate=0.02, indel_frac=0.2))
