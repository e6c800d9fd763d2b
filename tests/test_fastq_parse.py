"""bm2_fastq_parse_mt (four-line fast path, AVX2 loops where the CPU has them, several threads) against bm2_fastq_parse (the sequential parser with kseq's
grammar) on records that exercise every per-base path: lengths around the 32-base vector width, lower case, N and other IUPAC letters, CRLF, quality
lines that begin with '@' or '+', names with "/1" and comments -- and inputs the fast path must hand to the sequential parser."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))


def _lib(emu_lib):
    import bm2
    if not os.path.exists(bm2.LIB_PATH):
        bm2.LIB_PATH = emu_lib                                   # (host functions only: any build of the library has them)
    return bm2


def _records(seed, n, crlf=False):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRYKMSWBDHVU.-*", np.uint8)
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 76, 150, 151, int(rng.integers(1, 400))]))
        seq = alpha[rng.integers(0, len(alpha), L)].tobytes()
        qual = bytes(rng.integers(33, 74, L).astype(np.uint8))
        if i % 7 == 0:
            qual = b"@" + qual[1:]                                # a quality line that looks like a header
        if i % 11 == 0:
            qual = b"+" + qual[1:]
        name = b"r%d" % i + (b"/1" if i % 3 == 0 else b"") + (b" cm:Z:x y" if i % 5 == 0 else b"")
        out.append(b"@" + name + nl + seq + nl + b"+" + nl + qual + nl)
    return b"".join(out)


def _same(bm2, text1, text2, threads):
    ch = bm2.FastqChunk(text1, text2, threads)
    try:
        a = bm2.fastq_parse(text1)
        if text2 is None:
            exp_enc, exp_len, names, comments, quals = a[0], a[2], a[3], a[4], a[5]
        else:
            b = bm2.fastq_parse(text2)
            n = min(len(a[2]), len(b[2]))
            encs, exp_len, names, comments, quals = [], [], [], [], []
            for i in range(n):
                for r in (a, b):
                    encs.append(r[0][r[1][i]:r[1][i] + r[2][i]]); exp_len.append(r[2][i]); names.append(r[3][i]); comments.append(r[4][i]); quals.append(r[5][i])
            exp_enc = np.concatenate(encs) if encs else np.zeros(0, np.uint8)
        assert ch.n_reads == len(exp_len)
        assert np.array_equal(ch.len, np.asarray(exp_len, np.int32)) and np.array_equal(ch.enc, exp_enc)
        for i in range(ch.n_reads):
            assert ch.f.name[i] == names[i] and ch.f.qual[i] == quals[i] and (ch.f.comment[i] or None) == (comments[i] or None), i
    finally:
        ch.close()


def test_fast_path_equals_the_sequential_parser(emu_lib):
    bm2 = _lib(emu_lib)
    for seed, crlf in ((1, False), (2, True)):
        t1, t2 = _records(seed, 3000, crlf), _records(seed + 10, 2900, crlf)
        for threads in (1, 3):
            _same(bm2, t1, None, threads)
            _same(bm2, t1, t2, threads)


def test_inputs_the_fast_path_hands_over(emu_lib):
    bm2 = _lib(emu_lib)
    good = _records(5, 200)
    # a sequence line with '>' / '+' / '@' in it (in the vector part and in the tail), wrapped sequence lines, FASTA: the sequential parser's grammar decides
    for bad in (b"@x\n" + b"A" * 40 + b">" + b"C" * 10 + b"\n+\n" + b"I" * 51 + b"\n",
                b"@x\n" + b"A" * 70 + b"@\n+\n" + b"I" * 71 + b"\n",
                b"@w\nACGT\nACGT\n+\nIIIIIIII\n",
                b">fa\nACGTNNAC\n"):
        _same(bm2, good + bad + good, None, 2)
