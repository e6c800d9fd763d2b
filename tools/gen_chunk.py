#!/usr/bin/env python3
"""One synthetic PE chunk as two FASTQ files (bench.py's end-to-end leg makes its distinct chunks with several of these processes side by
side: a chunk takes ~10 s of single-threaded numpy).   python tools/gen_chunk.py <contigs.npz> <seed> <n_pairs> <L> <out1.fq> <out2.fq> <name prefix>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth  # noqa: E402


def main():
    meta, seed, n_pairs, L, o1, o2, pre = sys.argv[1:8]
    prefix = meta[:-len(".contigs.npz")] if meta.endswith(".contigs.npz") else None
    lens_fn = prefix + ".contig_lens.npy" if prefix else None
    if lens_fn and os.path.exists(lens_fn) and os.path.exists(prefix + ".0123"):
        # the forward strand of the index's one-base-per-byte text IS the concatenated genome (N runs hold the random bases the index
        # put there): mapped, not loaded -- ten of these processes side by side would otherwise hold 6 GB each at GRCh38 size
        lens = np.load(lens_fn)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        contigs = (np.memmap(prefix + ".0123", dtype=np.uint8, mode="r")[:int(offs[-1])], offs)
    else:
        z = np.load(meta, allow_pickle=True)
        contigs = [z["c%d" % i] for i in range(int(z["n"]))]
    r1, r2 = synth.make_reads_pe(int(seed), contigs, int(n_pairs), L=int(L))
    synth.write_fastq(o1, r1, prefix=pre, suffix="/1")
    synth.write_fastq(o2, r2, prefix=pre, suffix="/2")


if __name__ == "__main__":
    main()
