#!/usr/bin/env python3
"""End-to-end harness: FASTQ in, SAM out, through the library: bm2_fastq_parse -> bm2_batch_upload / run / finish (device: seeding,
chaining, extension, mem_sort_dedup_patch; there is no CPU path, the script fails without a GPU) -> bm2_sam_se / bm2_sam_pe.
Chunks the input as `bwa-mem2 mem -K` does (whole reads / pairs until the chunk holds >= K bases, fastmap.cpp:943-949,
bwa.cpp:62-216), so insert-size models, tie-breaking hashes and pair ids match the reference run with the same -K.
Output = @SQ header lines + alignment lines (no @PG: that line is the caller's command line).

    python tools/bm2_mem.py [-K bases] <idx_prefix> <in1.fq> [in2.fq] > out.sam

run(..., hits_of=f) lets the tests put a stand-in for the device stage (tests/helpers.py does, to check the harness on a CPU)."""
import argparse
import os

os.environ.setdefault("BM2_MALLOC_TUNE", "1")        # this program IS the host: the tail's threads run with glibc's trim / mmap thresholds set (sam_tail.cpp: the library touches them only on request)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))


def chunks(n_reads, lens, K, paired):
    """[(lo, hi)) read ranges: bseq_read takes reads (pairs) until the accumulated bases reach K."""
    lo, step = 0, 2 if paired else 1
    while lo < n_reads:
        hi, size = lo, 0
        while hi < n_reads and size < K:
            size += int(lens[hi:hi + step].sum())
            hi += step
        yield lo, hi
        lo = hi


def run(prefix, fq, K=10000000, out_path="-", threads=0, hits_of=None, device_tail=False, contexts=1):
    """hits_of(enc, off, ln) -> (alnregs ALNREG_DT, aln_off): the device stage (up to and including mem_sort_dedup_patch); None = a
    Context on GPU 0.  contexts = G > 1: every chunk is split at multiples of 512 reads over G contexts -- one per GPU while there
    are GPUs, further ones share a replica -- and paired ONCE (bm2_chunk_hits_sharded): same SAM for every G.
    device_tail: the mate-rescue and CIGAR alignments of the SAM tail run as device batches too (bm2_sam_pe_dev / bm2_sam_se_dev)."""
    import bm2
    paired = len(fq) == 2
    parts = [bm2.fastq_parse(open(f, "rb").read()) for f in fq]
    if paired:
        p, q = parts
        if len(p[2]) != len(q[2]):
            raise SystemExit("the two files hold different numbers of reads")
        seqs, names, quals = [], [], []
        for i in range(len(p[2])):
            for e in (p, q):
                seqs.append(e[0][e[1][i]:e[1][i] + e[2][i]]); names.append(e[3][i]); quals.append(e[5][i])
    else:
        e = parts[0]
        seqs = [e[0][e[1][i]:e[1][i] + e[2][i]] for i in range(len(e[2]))]
        names, quals = e[3], e[5]
    lens = np.array([len(s) for s in seqs], np.int64)
    opt, so = bm2.default_opt(), bm2.default_sam_opt(n_threads=threads)
    ctx = None
    if hits_of is None or device_tail:
        ctx = bm2.Context(0, prefix)
    extra = []
    tail_ctx = ctx
    if hits_of is None and contexts > 1:
        ndev = max(bm2.lib().bm2_device_count(), 1)
        idx = bm2.Index(prefix)
        per_dev = {0: ctx}
        ctxs = [ctx]
        for k in range(1, contexts):
            d = k % ndev
            c = bm2.Context(share=per_dev[d]) if d in per_dev else bm2.Context(d, idx)
            per_dev.setdefault(d, c); ctxs.append(c); extra.append(c)
        hits_of = lambda enc, off, ln: bm2.chunk_hits_sharded(ctxs, (enc, off, ln), opt)
        tail_ctx = ctxs                                      # the tail's rescue / CIGAR batches over the same contexts (bm2_sam_*_dev_multi)
    if hits_of is None:
        def hits_of(enc, off, ln):
            ctx.batch_upload(enc, off, ln); ctx.batch_run(opt); ctx.batch_finish(opt)
            return ctx.batch_download_alnregs()
    out = sys.stdout.buffer if out_path == "-" else open(out_path, "wb")
    out.write(bm2.sam_header(prefix))
    for lo, hi in chunks(len(seqs), lens, K, paired):
        enc = np.concatenate(seqs[lo:hi]) if hi > lo else np.zeros(0, np.uint8)
        ln = lens[lo:hi].astype(np.int32)
        off = np.concatenate([[0], np.cumsum(ln[:-1])]).astype(np.int64)
        aln, aln_off = hits_of(enc, off, ln)
        if paired:
            txt, _ = bm2.sam_pe(prefix, enc, off, ln, opt, aln, aln_off, names[lo:hi], quals[lo:hi], None, so, n_processed=lo, ctx=tail_ctx if device_tail else None)
        else:
            txt = bm2.sam_se(prefix, enc, off, ln, opt, aln, aln_off, names[lo:hi], quals[lo:hi], None, so, n_processed=lo, ctx=tail_ctx if device_tail else None)
        out.write(txt)
    if out is not sys.stdout.buffer:
        out.close()
    for c in reversed(extra):
        c.close()
    if ctx is not None:
        ctx.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-K", type=int, default=10000000)
    ap.add_argument("--threads", type=int, default=0, help="host threads of the SAM tail (0 = all)")
    ap.add_argument("--device-tail", action="store_true", help="rescue and CIGAR alignments of the SAM tail on the device as well")
    ap.add_argument("--contexts", type=int, default=1, help="split every chunk over this many contexts (GPUs first); one pairing per chunk")
    ap.add_argument("-o", default="-")
    ap.add_argument("prefix")
    ap.add_argument("fq", nargs="+")
    a = ap.parse_args(argv)
    run(a.prefix, a.fq, a.K, a.o, a.threads, device_tail=a.device_tail, contexts=a.contexts)


if __name__ == "__main__":
    main()
