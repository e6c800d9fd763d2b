"""Single-end SAM tail (bm2_sam_se = mem_mark_primary_se + mem_reg2sam + mem_gen_alt + mem_reg2aln + mem_aln2sam) against the
text the compiled reference prints for the same reads.  The regs come from the oracle (bit-identical to the device path,
tests/test_pipeline_gpu.py) through bm2_finish_regs, so the whole chain runs without a GPU."""
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ONT2D, oracle_finish_regs, ref_binary
from tools import oracle, refio, synth
import helpers  # noqa: E402


def _prg_to_regs(prg, n_reads):
    regs = np.zeros(len(prg), bm2.REG_DT)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        regs[f] = prg[f]
    reg_off = np.zeros(n_reads + 1, np.int64)
    np.add.at(reg_off, prg["read"] + 1, 1)
    return regs, np.cumsum(reg_off)


def _reference_sam(fa, fq, extra=()):
    p = subprocess.run([ref_binary(), "mem", "-t", "1"] + list(extra) + [fa, fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    return b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@"))


def _ours(fa, reads, names, quals, okw=None, sam_opt=None, comments=None, ctx=None):
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**(okw or {})))
    finally:
        ix.close()
    opt = bm2.default_opt(**(okw or {}))
    regs, reg_off = _prg_to_regs(exp["REGPRG"], len(ln))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
    return bm2.sam_se(fa, enc, off, ln, opt, aln, aln_off, names, quals, comments, sam_opt, ctx=ctx)


def _diff(a, b):
    la, lb = a.splitlines(), b.splitlines()
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d\n  ref : %s\n  ours: %s" % (i, x.decode()[:600], y.decode()[:600])
    return "line counts %d vs %d" % (len(la), len(lb))


def _case(tmp_path, seed, n_reads, L=150, **genome_kw):
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present (build it with `make -C oracle ref`)")
    kw = dict(alt_contigs=1, alt_len=4000, n_repeat_families=8, repeat_len=(200, 2500), copies=(3, 30), divergence=(0.0, 0.06))
    kw.update(genome_kw)
    names, ctg, alts = synth.make_genome(seed, [250000, 120000, 40000], **kw)
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    if alts:
        synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_se(seed + 1, ctg, n_reads, L=L, sub_rate=0.01, indel_frac=0.15, random_frac=0.01)
    return fa, reads


def _write_fastq(path, reads, quals):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@q%d\n" % i + b"ACGTN"[0:0] + bytes(b"ACGTN"[c] for c in r) + b"\n+\n" + quals[i] + b"\n")


def test_sam_se_matches_reference_text(tmp_path, monkeypatch):
    fa, reads = _case(tmp_path, 41, 4000)
    rng = np.random.default_rng(3)
    quals = [bytes(rng.integers(35, 74, size=len(r), dtype=np.uint8)) for r in reads]
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads, quals)
    ref = _reference_sam(fa, fq)
    got = _ours(fa, reads, ["q%d" % i for i in range(len(reads))], quals)
    assert ref == got, _diff(ref, got)
    assert ref.count(b"SA:Z:") > 0 and ref.count(b"XA:Z:") > 0 and ref.count(b"\t4\t*\t0\t0\t*") > 0      # the case has all record kinds
    monkeypatch.setenv("BM2_CIGAR_FLAT", "1")                   # the same with CIGAR generation as a session (dry pass, batch, real pass)
    assert _ours(fa, reads, ["q%d" % i for i in range(len(reads))], quals) == got
    cp, cu, cm = bm2.sam_cigar_stats()
    assert cu >= cp > 3000 and cm == 0, (cp, cu, cm)       # hits in the batch, lookups it served, numbered hits it lacked


def test_sam_se_options(tmp_path):
    # -a (all alignments), -Y (soft clips), -M (secondary flag for split hits), -T, -5, non-default scoring
    fa, reads = _case(tmp_path, 43, 1500, L=120)
    quals = [b"F" * len(r) for r in reads]
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads, quals)
    names = ["q%d" % i for i in range(len(reads))]
    for extra, okw, flag, T in ((["-a"], {}, 0x8, 30), (["-Y", "-M"], {}, 0x200 | 0x10, 30), (["-T", "50", "-5"], {}, 0x800 | 0x1000, 50),
                                (["-B", "3", "-O", "5,7", "-E", "2,1", "-L", "4,6"], dict(b=3, o_del=5, o_ins=7, e_del=2, e_ins=1, pen_clip5=4, pen_clip3=6), 0, 30)):
        ref = _reference_sam(fa, fq, extra)
        got = _ours(fa, reads, names, quals, okw, bm2.default_sam_opt(flag=flag, T=T))
        assert ref == got, "%s: %s" % (" ".join(extra), _diff(ref, got))


def test_sam_se_long_reads_ont2d(tmp_path):
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present")
    names, ctg, alts = synth.make_genome(47, [180000, 90000], alt_contigs=0, n_repeat_families=3, repeat_len=(300, 2000), copies=(3, 8),
                                         divergence=(0.0, 0.05))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_long(48, ctg, 60, mean_len=2500, max_len=6000, err=0.08)
    quals = [b"5" * len(r) for r in reads]
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads, quals)
    ref = _reference_sam(fa, fq, ["-x", "ont2d"])
    got = _ours(fa, reads, ["q%d" % i for i in range(len(reads))], quals, ONT2D, bm2.default_sam_opt(T=ONT2D.get("T", 30)))
    assert ref == got, _diff(ref, got)


# ---- paired-end ------------------------------------------------------------------------------------------------------
def _pe_case(tmp_path, seed, n_pairs, L=150, **rkw):
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present (build it with `make -C oracle ref`)")
    names, ctg, alts = synth.make_genome(seed, [300000, 150000, 60000], alt_contigs=1, alt_len=4000, n_repeat_families=8,
                                         repeat_len=(200, 2500), copies=(3, 30), divergence=(0.0, 0.06))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r1, r2 = synth.make_reads_pe(seed + 1, ctg, n_pairs, L=L, **rkw)
    return fa, r1, r2


def _pe_run(tmp_path, fa, r1, r2, extra, flag=0, okw=None, ctx=None, **skw):
    rng = np.random.default_rng(9)
    reads, quals, names = [], [], []
    for i in range(len(r1)):
        for r in (r1[i], r2[i]):
            reads.append(r); quals.append(bytes(rng.integers(40, 74, size=len(r), dtype=np.uint8))); names.append("p%d" % i)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    for path, sel in ((f1, 0), (f2, 1)):
        with open(path, "wb") as f:
            for i in range(sel, len(reads), 2):
                f.write(b"@" + names[i].encode() + b"\n" + bytes(b"ACGTN"[c] for c in reads[i]) + b"\n+\n" + quals[i] + b"\n")
    p = subprocess.run([ref_binary(), "mem", "-t", "1"] + list(extra) + [fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@"))
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**(okw or {})))
    finally:
        ix.close()
    opt = bm2.default_opt(**(okw or {}))
    regs, reg_off = _prg_to_regs(exp["REGPRG"], len(ln))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
    got, pes = bm2.sam_pe(fa, enc, off, ln, opt, aln, aln_off, names, quals, None, bm2.default_sam_opt(flag=flag, **skw), ctx=ctx)
    return ref, got, pes


def test_sam_pe_without_rescue_matches_reference_text(tmp_path):
    # -S: no mate rescue -> mem_pestat, mem_pair, mapping qualities and the paired record layout alone
    fa, r1, r2 = _pe_case(tmp_path, 51, 3000)
    ref, got, pes = _pe_run(tmp_path, fa, r1, r2, ["-S"], flag=0x20)
    assert not pes[1].failed and pes[1].low > 0
    assert ref == got, _diff(ref, got)


def test_sam_pe_matches_reference_text(tmp_path):
    # the default: mate rescue (local SW of the mate near the expected position), pairing, proper-pair flags, MC/TLEN
    fa, r1, r2 = _pe_case(tmp_path, 53, 3000)
    ref, got, pes = _pe_run(tmp_path, fa, r1, r2, [])
    assert ref == got, _diff(ref, got)


def test_sam_pe_rescue_batched_equals_inline(tmp_path, monkeypatch):
    # mate rescue planned up front and run as one batch (the default: the shape the device kernel needs) against the alignments made
    # inside the pair loop as mem_sam_pe makes them; the plan must cover what the pairs ask for
    fa, r1, r2 = _pe_case(tmp_path, 57, 2500, sub_rate=0.03, indel_frac=0.3, random_frac=0.05)
    ref, got, pes = _pe_run(tmp_path, fa, r1, r2, [])
    planned, used, missed = bm2.sam_rescue_stats()
    assert ref == got, _diff(ref, got)
    assert planned > 500 and used <= planned and missed <= planned // 100, (planned, used, missed)
    ref2, got2, pes2 = _pe_run(tmp_path, fa, r1, r2, [], rescue_inline=1)
    assert bm2.sam_rescue_stats() == (0, 0, 0)
    assert got2 == got
    # the batch as the flat arrays a device kernel takes (oriented mates in one buffer, targets as positions in ref_string)
    monkeypatch.setenv("BM2_RESCUE_FLAT", "1")
    ref3, got3, pes3 = _pe_run(tmp_path, fa, r1, r2, [])
    assert got3 == got and bm2.sam_rescue_stats() == (planned, used, missed)
    # CIGAR generation as a session: a dry pass records every alignment the flow could ask for, one batch, the real pass looks them up
    monkeypatch.setenv("BM2_CIGAR_FLAT", "1")
    ref4, got4, pes4 = _pe_run(tmp_path, fa, r1, r2, [])
    cp, cu, cm = bm2.sam_cigar_stats()
    assert got4 == got and cu >= cp > 2000 and cm == 0, (cp, cu, cm)


def test_sam_pe_text_does_not_depend_on_threads_or_batching(tmp_path, monkeypatch):
    # the worker pool, the in-order text commit and the scanned placement of the rescue batch: the same bytes with 1, 3 and 16 threads
    # (more than this machine has), with the alignments in place, as host batches and with the CIGAR session
    fa, r1, r2 = _pe_case(tmp_path, 59, 1800, sub_rate=0.03, indel_frac=0.3, random_frac=0.04)
    ref, base, pes = _pe_run(tmp_path, fa, r1, r2, [], n_threads=1)
    assert ref == base, _diff(ref, base)
    for env in ({}, {"BM2_RESCUE_FLAT": "1", "BM2_CIGAR_FLAT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for nt in (1, 3, 16):
            _, got, _ = _pe_run(tmp_path, fa, r1, r2, [], n_threads=nt)
            assert got == base, (env, nt)


def test_sam_pe_noisy_mates_and_options(tmp_path):
    # noisy, shorter reads: many mates seed badly or not at all, so the records depend on the rescue SW (score, sub-optimal
    # score, start found by the reverse pass); then the option paths: -a, -Y, -P (no pairing), -U / -m, non-default scoring
    fa, r1, r2 = _pe_case(tmp_path, 57, 1500, L=100, sub_rate=0.04, indel_frac=0.3, ins_mean=300, ins_sd=60, random_frac=0.02)
    for extra, flag, okw, skw in (([], 0, None, {}), (["-a"], 0x8, None, {}), (["-Y"], 0x200, None, {}), (["-P"], 0x4, None, {}),
                                  (["-U", "9", "-m", "2"], 0, None, dict(pen_unpaired=9, max_matesw=2)),
                                  # (-A scales the options that are not given: zdrop, clipping penalties, -T and -U, fastmap.cpp:547-561)
                                  (["-A", "2", "-B", "5", "-O", "7,8", "-E", "2,2"], 0,
                                   dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10), dict(T=60, pen_unpaired=34))):
        ref, got, pes = _pe_run(tmp_path, fa, r1, r2, extra, flag=flag, okw=okw, **skw)
        assert ref == got, "%s: %s" % (" ".join(extra), _diff(ref, got))


def test_fastq_text_to_sam_text(tmp_path):
    # the host I/O row end to end: FASTQ text (names with /1, comments, lower case, multi-line FASTA records mixed in)
    # -> bm2_fastq_parse -> regs -> bm2_sam_se == `bwa-mem2 mem -C` on the same file
    fa, reads = _case(tmp_path, 61, 600)
    rng = np.random.default_rng(4)
    fq = str(tmp_path / "r.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            s = bytes(b"ACGTN"[c] for c in r)
            if i % 7 == 3:
                s = s.lower()
            if i % 11 == 5:                                      # a FASTA record, sequence on two lines
                f.write(b">fa%d/2 XY:i:%d\n" % (i, i) + s[:70] + b"\n" + s[70:] + b"\n")
            else:
                q = bytes(rng.integers(35, 74, size=len(r), dtype=np.uint8))
                f.write(b"@rd%d/1 BC:Z:AC%dGT\tRX:Z:x\n" % (i, i) + s + b"\n+\n" + q + b"\n")
    ref = _reference_sam(fa, fq, ["-C"])
    enc, off, ln, names, comments, quals = bm2.fastq_parse(open(fq, "rb").read())
    assert len(ln) == len(reads) and names[0] == b"rd0" and comments[0] == b"BC:Z:AC0GT\tRX:Z:x"
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln)
    finally:
        ix.close()
    opt = bm2.default_opt()
    regs, reg_off = _prg_to_regs(exp["REGPRG"], len(ln))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
    got = bm2.sam_se(fa, enc, off, ln, opt, aln, aln_off, names, quals, comments)
    assert ref == got, _diff(ref, got)


def test_sam_pe_chunked_like_the_reference(tmp_path):
    # -K: the reference cuts the input into chunks; the insert-size model is re-estimated per chunk and the hash / pair ids keep
    # counting (n_processed).  Calling bm2_sam_pe once per chunk with n_processed must give the same text.
    fa, r1, r2 = _pe_case(tmp_path, 67, 1600, L=100, sub_rate=0.02, indel_frac=0.2, ins_mean=280, ins_sd=50)
    K = 60000
    reads, quals, names = [], [], []
    for i in range(len(r1)):
        for r in (r1[i], r2[i]):
            reads.append(r); quals.append(b"G" * len(r)); names.append("p%d" % i)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    for path, sel in ((f1, 0), (f2, 1)):
        with open(path, "wb") as f:
            for i in range(sel, len(reads), 2):
                f.write(b"@" + names[i].encode() + b"\n" + bytes(b"ACGTN"[c] for c in reads[i]) + b"\n+\n" + quals[i] + b"\n")
    p = subprocess.run([ref_binary(), "mem", "-t", "1", "-K", str(K), fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@"))
    opt = bm2.default_opt()
    ix = oracle.Index(fa)
    got, lo, n_chunks = b"", 0, 0
    try:
        while lo < len(reads):
            hi, size = lo, 0
            while hi < len(reads) and size < K:                  # bseq_read: whole pairs until the chunk holds >= K bases
                size += len(reads[hi]) + len(reads[hi + 1]); hi += 2
            enc, off, ln = refio.pack_reads(reads[lo:hi])
            exp = ix.run(enc, off, ln)
            regs, reg_off = _prg_to_regs(exp["REGPRG"], len(ln))
            aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
            txt, pes = bm2.sam_pe(fa, enc, off, ln, opt, aln, aln_off, names[lo:hi], quals[lo:hi], None, None, n_processed=lo)
            got += txt; lo = hi; n_chunks += 1
    finally:
        ix.close()
    assert n_chunks > 3
    assert ref == got, _diff(ref, got)


def test_sam_pe_given_insert_size_model(tmp_path):
    # -I mean,std: the caller fixes the FR model (fastmap.cpp:703-718); the other orientations are off
    fa, r1, r2 = _pe_case(tmp_path, 71, 1200, L=120, ins_mean=260, ins_sd=20, random_frac=0.1)
    reads, quals, names = [], [], []
    for i in range(len(r1)):
        for r in (r1[i], r2[i]):
            reads.append(r); quals.append(b"E" * len(r)); names.append("p%d" % i)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    for path, sel in ((f1, 0), (f2, 1)):
        with open(path, "wb") as f:
            for i in range(sel, len(reads), 2):
                f.write(b"@" + names[i].encode() + b"\n" + bytes(b"ACGTN"[c] for c in reads[i]) + b"\n+\n" + quals[i] + b"\n")
    p = subprocess.run([ref_binary(), "mem", "-t", "1", "-I", "260,20", fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@"))
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln)
    finally:
        ix.close()
    opt = bm2.default_opt()
    regs, reg_off = _prg_to_regs(exp["REGPRG"], len(ln))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
    pin = [bm2.PeStat(0, 0, 1, 0, 0., 0.) for _ in range(4)]
    pin[1] = bm2.PeStat(max(int(260. - 4. * 20. + .499), 1), int(260. + 4. * 20. + .499), 0, 0, 260., 20.)
    got, pes = bm2.sam_pe(fa, enc, off, ln, opt, aln, aln_off, names, quals, pes_in=pin)
    assert ref == got, _diff(ref, got)


def test_sam_se_reference_header_tags(tmp_path):
    # -V (XR:Z: = the contig's FASTA comment, tabs turned into spaces), -R (RG:Z:), -h (XA limits)
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present")
    names, ctg, alts = synth.make_genome(91, [200000, 90000], alt_contigs=1, alt_len=4000, n_repeat_families=5, repeat_len=(200, 2000),
                                         copies=(3, 20), divergence=(0.0, 0.05))
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        for i, (n, c) in enumerate(zip(names, ctg)):
            f.write(">%s assembly=test%d\tnote with tab\n" % (n, i) if i != 1 else ">%s\n" % n)
            s = "".join("ACGTN"[x] for x in c)
            for k in range(0, len(s), 80):
                f.write(s[k:k + 80] + "\n")
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_se(92, ctg, 1200, L=120, sub_rate=0.02, indel_frac=0.2)
    quals = [b"I" * len(r) for r in reads]
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads, quals)
    names_ = ["q%d" % i for i in range(len(reads))]
    for extra, so in ((["-V"], dict(flag=0x100)), (["-R", "@RG\\tID:grp1\\tSM:x"], dict(rg_id=b"grp1")), (["-h", "2,3"], dict(max_XA_hits=2, max_XA_hits_alt=3))):
        ref = _reference_sam(fa, fq, extra)
        got = _ours(fa, reads, names_, quals, None, bm2.default_sam_opt(**so))
        assert ref == got, "%s: %s" % (" ".join(extra), _diff(ref, got))
    assert b"XR:Z:assembly=test0 note with tab" in _reference_sam(fa, fq, ["-V"])
    # the header: @SQ lines (AH:* on the ALT contig), then the caller's @RG line; @PG (the command line) is the caller's
    p = subprocess.run([ref_binary(), "mem", "-t", "1", "-R", "@RG\\tID:grp1\\tSM:x", fa, fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    hdr = b"".join(l for l in p.stdout.splitlines(keepends=True) if l.startswith(b"@") and not l.startswith(b"@PG"))
    assert hdr == bm2.sam_header(fa, b"@RG\tID:grp1\tSM:x") and b"\tAH:*\n" in hdr


def test_sam_pe_chimeric_mates(tmp_path):
    # mates whose tail comes from another locus: split hits -> supplementary records, the no_pairing branch with a mate,
    # SA:Z: lists; with -M (secondary flag instead), -Y -a, -5
    fa, r1, r2 = _pe_case(tmp_path, 121, 1400)
    rng = np.random.default_rng(8)
    names_g, ctg, _ = synth.make_genome(121, [300000, 150000, 60000], alt_contigs=1, alt_len=4000, n_repeat_families=8, repeat_len=(200, 2500),
                                        copies=(3, 30), divergence=(0.0, 0.06))
    genome = np.concatenate(ctg)
    for i in range(0, len(r1), 7):
        p = int(rng.integers(0, len(genome) - 100)); k = int(rng.integers(50, 100))
        piece = genome[p:p + (150 - k)].copy(); piece[piece > 3] = 0
        (r1 if i % 2 else r2)[i][k:] = piece
    for extra, flag in (([], 0), (["-M"], 0x10), (["-Y", "-a"], 0x208), (["-5"], 0x1800)):
        ref, got, pes = _pe_run(tmp_path, fa, r1, r2, extra, flag=flag)
        assert ref == got, "%s: %s" % (" ".join(extra), _diff(ref, got))
        if not extra:
            assert sum(1 for l in ref.splitlines() if int(l.split(b"\t")[1]) & 0x800) > 50


def test_end_to_end_harness_with_the_oracle_backend(tmp_path):
    # tools/bm2_mem.py: FASTQ files -> chunks as `mem -K` cuts them -> SAM incl. @SQ header; the oracle stands in for the device
    # (the same harness with --backend gpu is what tests/test_end_to_end_gpu.py exercises piece by piece)
    fa, r1, r2 = _pe_case(tmp_path, 131, 900, L=100, ins_mean=280, ins_sd=40)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    for path, rr, sfx in ((f1, r1, b"/1"), (f2, r2, b"/2")):
        with open(path, "wb") as f:
            for i, r in enumerate(rr):
                f.write(b"@frag%d" % i + sfx + b"\n" + bytes(b"ACGTN"[c] for c in r) + b"\n+\n" + b"F" * len(r) + b"\n")
    K = 40000
    p = subprocess.run([ref_binary(), "mem", "-t", "1", "-K", str(K), fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@PG"))
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "out.sam")
    from helpers import oracle_regs_fn
    from tools import bm2_mem
    bm2_mem.run(fa, [f1, f2], K, out, hits_of=oracle_regs_fn(fa))
    got = open(out, "rb").read()
    assert ref == got, _diff(ref, got)
