"""FASTQ text in, SAM text out, with the hot path on the device: bm2_fastq_parse -> bm2_seed_chain_extend (GPU) ->
bm2_batch_finish -> bm2_sam_pe, against the text the compiled reference prints for the same files.  This is the qualifier of
the headline metric ("SAM bit-exact vs ref") checked through every layer at once; each layer has its own tests
(test_pipeline_gpu.py: device stages; test_sam_tail.py: the host tail on oracle regs)."""
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ref_binary
from tools import synth
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu


def test_fastq_to_sam_paired_end_through_the_device(gpu_ctx_factory, tmp_path):
    exe = ref_binary()
    if exe is None:
        helpers.no_checker("oracle/_ref reference binary not present")
    names, ctg, alts = synth.make_genome(81, [300000, 150000, 60000], alt_contigs=1, alt_len=4000, n_repeat_families=8, repeat_len=(200, 2500),
                                         copies=(3, 30), divergence=(0.0, 0.06))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r1, r2 = synth.make_reads_pe(82, ctg, 3000, L=150, sub_rate=0.015, indel_frac=0.15, random_frac=0.01)
    rng = np.random.default_rng(5)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    for path, rr, suffix in ((f1, r1, b"/1"), (f2, r2, b"/2")):
        with open(path, "wb") as f:
            for i, r in enumerate(rr):
                q = bytes(rng.integers(40, 74, size=len(r), dtype=np.uint8))
                f.write(b"@pair%d" % i + suffix + b"\n" + bytes(b"ACGTN"[c] for c in r) + b"\n+\n" + q + b"\n")
    p = subprocess.run([exe, "mem", "-t", "1", fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@"))

    a = bm2.fastq_parse(open(f1, "rb").read())
    b = bm2.fastq_parse(open(f2, "rb").read())
    n = len(a[2])
    assert n == len(b[2]) == 3000
    seqs, names_, quals = [], [], []
    for i in range(n):                                           # interleave the two files: reads 2i, 2i+1 are a pair
        for e in (a, b):
            seqs.append(e[0][e[1][i]:e[1][i] + e[2][i]]); names_.append(e[3][i]); quals.append(e[5][i])
    enc = np.concatenate(seqs)
    ln = np.array([len(s) for s in seqs], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1])]).astype(np.int64)

    opt = bm2.default_opt()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, opt)                 # the device
    aln, aln_off = ctx.finish_regs((enc, off, ln), opt, regs, reg_off)           # the tail of mem_kernel2_core, on the device too
    ctx.batch_upload(enc, off, ln); ctx.batch_run(opt); ctx.batch_finish(opt)    # the same through the resident path
    aln2, aln_off2 = ctx.batch_download_alnregs()
    assert aln.tobytes() == aln2.tobytes() and (aln_off == aln_off2).all()
    got, pes = bm2.sam_pe(fa, enc, off, ln, opt, aln, aln_off, names_, quals)
    if ref != got:
        la, lb = ref.splitlines(), got.splitlines()
        for i, (x, y) in enumerate(zip(la, lb)):
            assert x == y, "line %d\n  ref : %s\n  ours: %s" % (i, x.decode()[:500], y.decode()[:500])
        assert len(la) == len(lb)
