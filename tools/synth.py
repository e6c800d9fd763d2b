#!/usr/bin/env python3
"""Seeded synthetic genomes and reads (test + bench inputs; there is no data on the box).

Genome: uniform-random ACGT contigs with planted repeat families (diverged copies of a
few master sequences, both strands), optional N gaps, optional ALT contigs (near copies
of a primary region, listed in <prefix>.alt).  Reads: Illumina-like 150 bp SE/PE
(substitutions, at most one short indel per read, a few pure-random and N-containing
reads), or ONT-like long reads (config 5 shape).  Everything is numpy-vectorised so
10^6..10^7 reads are generated in seconds; the same (seed, arguments) always yields the
same bytes.

This is input generation only -- no alignment logic lives here.
"""
import argparse
import os
import sys

import numpy as np

_ACGTN = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _revcomp_codes(a):
    r = a[::-1].copy()
    m = r < 4
    r[m] = 3 - r[m]
    return r


def make_genome(seed, contig_lens, n_repeat_families=6, repeat_len=(300, 6000),
                copies=(5, 40), divergence=(0.01, 0.12), n_gaps=2, gap_len=(50, 500),
                alt_contigs=0, alt_len=20000):
    """Return (names, [uint8 code arrays 0..4], alt_names)."""
    rng = np.random.default_rng(seed)
    contigs = [rng.integers(0, 4, size=int(l), dtype=np.uint8) for l in contig_lens]
    names = ["chr%d" % (i + 1) for i in range(len(contigs))]
    total = sum(len(c) for c in contigs)
    # planted repeat families
    for _ in range(n_repeat_families):
        rl = int(rng.integers(repeat_len[0], repeat_len[1] + 1))
        master = rng.integers(0, 4, size=rl, dtype=np.uint8)
        nc = int(rng.integers(copies[0], copies[1] + 1))
        div = rng.uniform(divergence[0], divergence[1])
        for _c in range(nc):
            ci = int(rng.integers(0, len(contigs)))
            c = contigs[ci]
            if len(c) <= rl + 2:
                continue
            pos = int(rng.integers(0, len(c) - rl))
            cp = master.copy()
            mut = rng.random(rl) < div
            cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
            if rng.random() < 0.5:
                cp = _revcomp_codes(cp)
            c[pos:pos + rl] = cp
    # N gaps
    for _ in range(n_gaps):
        ci = int(rng.integers(0, len(contigs)))
        c = contigs[ci]
        gl = int(rng.integers(gap_len[0], gap_len[1] + 1))
        if len(c) <= gl * 4:
            continue
        pos = int(rng.integers(gl, len(c) - 2 * gl))
        c[pos:pos + gl] = 4
    # ALT contigs: diverged copies of a region of a primary contig
    alt_names = []
    for i in range(alt_contigs):
        ci = int(rng.integers(0, len(contigs)))
        c = contigs[ci]
        al = min(alt_len, len(c) // 2)
        pos = int(rng.integers(0, len(c) - al))
        cp = c[pos:pos + al].copy()
        mut = (rng.random(al) < 0.01) & (cp < 4)
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
        contigs.append(cp)
        names.append("chr%d_alt%d" % (ci + 1, i + 1))
        alt_names.append(names[-1])
    assert total > 0
    return names, contigs, alt_names


def write_fasta(path, names, contigs, width=60):
    with open(path, "wb") as f:
        for n, c in zip(names, contigs):
            f.write(b">" + n.encode() + b"\n")
            s = _ACGTN[c]
            full = (len(s) // width) * width
            if full:
                block = np.empty((full // width, width + 1), dtype=np.uint8)
                block[:, :width] = s[:full].reshape(-1, width)
                block[:, width] = 10
                f.write(block.tobytes())
            if full < len(s):
                f.write(s[full:].tobytes() + b"\n")


def write_alt(path, alt_names):
    with open(path, "w") as f:
        for n in alt_names:
            f.write(n + "\n")


def _concat(contigs):
    offs = np.zeros(len(contigs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(c) for c in contigs])
    return np.concatenate(contigs), offs


def _sample_windows(rng, offs, n, span):
    """n start positions such that [start, start+span) lies inside one contig."""
    lens = np.diff(offs)
    usable = np.maximum(lens - span, 0).astype(np.float64)
    assert usable.sum() > 0, "contigs shorter than the requested span"
    ci = rng.choice(len(lens), size=n, p=usable / usable.sum())
    start = offs[ci] + (rng.random(n) * usable[ci]).astype(np.int64)
    return start


def _mutate(rng, reads, sub_rate, ramp=0.0):
    n, L = reads.shape
    p = np.full(L, sub_rate)
    if ramp > 0:
        p = p + ramp * (np.arange(L) / max(L - 1, 1)) ** 2
    m = (rng.random((n, L)) < p[None, :]) & (reads < 4)
    k = int(m.sum())
    reads[m] = (reads[m] + rng.integers(1, 4, size=k, dtype=np.uint8)) % 4
    return reads


def _extract(rng, genome, start, L, indel_frac, max_indel=6):
    """Rows of length L starting at `start`, a fraction with one insertion or deletion."""
    n = len(start)
    j = np.arange(L, dtype=np.int64)[None, :]
    has = rng.random(n) < indel_frac
    is_del = rng.random(n) < 0.5
    ilen = rng.integers(1, max_indel + 1, size=n)
    ipos = rng.integers(10, max(L - 10, 11), size=n)
    dshift = np.where(has & is_del, ilen, 0)[:, None]
    ishift = np.where(has & ~is_del, ilen, 0)[:, None]
    ip = ipos[:, None]
    idx = start[:, None] + j + np.where(j >= ip, dshift, 0) - np.where(j >= ip + ishift, ishift, 0)
    idx = np.minimum(idx, len(genome) - 1)
    reads = genome[idx]
    ins_mask = (j >= ip) & (j < ip + ishift)
    k = int(ins_mask.sum())
    if k:
        reads[ins_mask] = rng.integers(0, 4, size=k, dtype=np.uint8)
    return reads


def _revcomp_rows(reads):
    r = reads[:, ::-1].copy()
    m = r < 4
    r[m] = 3 - r[m]
    return r


def make_reads_se(seed, contigs, n, L=150, sub_rate=0.006, ramp=0.02, indel_frac=0.07,
                  random_frac=0.002, n_frac=0.001):
    rng = np.random.default_rng(seed)
    genome, offs = _concat(contigs)
    start = _sample_windows(rng, offs, n, L + 8)
    reads = _extract(rng, genome, start, L, indel_frac)
    rev = rng.random(n) < 0.5
    reads[rev] = _revcomp_rows(reads[rev])
    reads = _mutate(rng, reads, sub_rate, ramp)
    rnd = rng.random(n) < random_frac
    reads[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), L), dtype=np.uint8)
    hasn = np.nonzero(rng.random(n) < n_frac)[0]
    if len(hasn):
        reads[hasn, rng.integers(0, L, size=len(hasn))] = 4
    return reads


def make_reads_pe(seed, contigs, n_pairs, L=150, ins_mean=400, ins_sd=40, **kw):
    sub_rate = kw.get("sub_rate", 0.006)
    ramp = kw.get("ramp", 0.02)
    indel_frac = kw.get("indel_frac", 0.07)
    random_frac = kw.get("random_frac", 0.002)
    n_frac = kw.get("n_frac", 0.001)
    rng = np.random.default_rng(seed)
    # contigs: a list of code arrays, or (genome, offsets) already concatenated -- e.g. a memory map of the index's .0123 file, which
    # several generator processes then share through the page cache instead of holding a copy of the genome each
    genome, offs = contigs if isinstance(contigs, tuple) else _concat(contigs)
    isz = np.maximum(np.rint(rng.normal(ins_mean, ins_sd, size=n_pairs)).astype(np.int64), L + 20)
    span = int(isz.max()) + 16
    start = _sample_windows(rng, offs, n_pairs, span)
    r1 = _extract(rng, genome, start, L, indel_frac)
    r2 = _revcomp_rows(_extract(rng, genome, start + isz - L, L, indel_frac))
    flip = rng.random(n_pairs) < 0.5       # fragment from the reverse strand: swap roles
    r1f = np.where(flip[:, None], r2, r1)
    r2f = np.where(flip[:, None], r1, r2)
    r1f = _mutate(rng, r1f, sub_rate, ramp)
    r2f = _mutate(rng, r2f, sub_rate, ramp)
    rnd = rng.random(n_pairs) < random_frac
    k = int(rnd.sum())
    r1f[rnd] = rng.integers(0, 4, size=(k, L), dtype=np.uint8)
    r2f[rnd] = rng.integers(0, 4, size=(k, L), dtype=np.uint8)
    hasn = np.nonzero(rng.random(n_pairs) < n_frac)[0]
    if len(hasn):
        r1f[hasn, rng.integers(0, L, size=len(hasn))] = 4
    return r1f, r2f


def make_reads_long(seed, contigs, n, mean_len=10000, max_len=30000, err=0.10,
                    mix=(0.30, 0.35, 0.35)):
    """ONT-like reads: list of 1-D code arrays (ragged). mix = (sub, del, ins) shares of err."""
    rng = np.random.default_rng(seed)
    genome, offs = _concat(contigs)
    sigma = 0.45
    lens = np.minimum(rng.lognormal(np.log(mean_len) - sigma * sigma / 2, sigma, size=n), max_len)
    lens = np.maximum(lens.astype(np.int64), 500)
    out = []
    for i in range(n):
        Ls = int(lens[i])
        st = int(_sample_windows(rng, offs, 1, Ls + 8)[0])
        src = genome[st:st + Ls]
        u = rng.random(Ls)
        p_sub, p_del, p_ins = err * mix[0], err * mix[1], err * mix[2]
        keep = u >= p_del
        seq = src.copy()
        sub = (u >= p_del) & (u < p_del + p_sub) & (seq < 4)
        seq[sub] = (seq[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) % 4
        ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
        reps = np.where(keep, 1, 0) + np.where(ins & keep, 1, 0)
        seq2 = np.repeat(seq, reps)
        # positions that are the duplicated (inserted) copy get a random base
        ends = np.cumsum(reps)
        dup = ends[(reps == 2)] - 1
        seq2[dup] = rng.integers(0, 4, size=len(dup), dtype=np.uint8)
        seq2 = seq2[:max_len]
        if rng.random() < 0.5:
            seq2 = _revcomp_codes(seq2)
        out.append(seq2)
    return out


def write_fastq(path, reads, prefix="r", suffix="", qual=b"I"):
    """reads: 2-D uint8 code array (fixed length) or list of 1-D arrays."""
    with open(path, "wb") as f:
        if isinstance(reads, np.ndarray) and reads.ndim == 2:
            n, L = reads.shape
            seqs = _ACGTN[reads]
            q = qual * L
            chunk = []
            for i in range(n):
                chunk.append(b"@%s%d%s\n" % (prefix.encode(), i, suffix.encode()))
                chunk.append(seqs[i].tobytes())
                chunk.append(b"\n+\n" + q + b"\n")
                if len(chunk) >= 30000:
                    f.write(b"".join(chunk)); chunk = []
            f.write(b"".join(chunk))
        else:
            for i, r in enumerate(reads):
                f.write(b"@%s%d%s\n" % (prefix.encode(), i, suffix.encode()))
                f.write(_ACGTN[r].tobytes())
                f.write(b"\n+\n" + qual * len(r) + b"\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--out", required=True, help="output prefix")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--contigs", default="400000,250000,150000",
                    help="comma separated contig lengths")
    ap.add_argument("--alt", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--pe", action="store_true")
    ap.add_argument("--long", action="store_true")
    a = ap.parse_args(argv)
    lens = [int(x) for x in a.contigs.split(",")]
    names, contigs, alts = make_genome(a.seed, lens, alt_contigs=a.alt)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    write_fasta(a.out + ".fa", names, contigs)
    if alts:
        write_alt(a.out + ".fa.alt", alts)
    if a.long:
        write_fastq(a.out + ".fq", make_reads_long(a.seed + 1, contigs, a.reads))
    elif a.pe:
        r1, r2 = make_reads_pe(a.seed + 1, contigs, a.reads // 2, L=a.len)
        write_fastq(a.out + "_1.fq", r1, suffix="/1")
        write_fastq(a.out + "_2.fq", r2, suffix="/2")
    else:
        write_fastq(a.out + ".fq", make_reads_se(a.seed + 1, contigs, a.reads, L=a.len))
    return 0


if __name__ == "__main__":
    sys.exit(main())
