"""The device kernels of the host tail (last in the suite on purpose: they were written after the round's GPU minutes were spent and ran
on the host emulator only).  Mate-rescue SW on the device (bm2_ksw_align2_dev, matesw.hip) against the host kernel that is pinned to the reference's
ksw_align2 (tests/test_ksw_align2.py): all seven result fields, byte and word lanes, with / without the start pass and the
minimum score, degenerate lengths, several scorings."""
import numpy as np
import pytest

import bm2
from test_ksw_align2 import KSW_XBYTE, KSW_XSTART, KSW_XSUBO, _pairs
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu


def _xtra(pairs, opt, rng):
    out = []
    for q, t in pairs:
        x = 19 * opt.a
        if rng.random() < 0.9:
            x |= KSW_XSUBO
        if rng.random() < 0.9:
            x |= KSW_XSTART
        if len(q) * opt.a < 250 and rng.random() < 0.9:
            x |= KSW_XBYTE
        out.append(x)
    return out


@pytest.mark.parametrize("kw", [{}, dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1), dict(b=1, o_del=1, o_ins=1)])
def test_device_ksw_align2_equals_host(gpu_ctx_factory, kw):
    ctx = gpu_ctx_factory()
    opt = bm2.default_opt(**kw)
    rng = np.random.default_rng(11)
    pairs = _pairs(23 + len(kw), 3000)
    for ql, tl in ((1, 1), (1, 40), (16, 16), (17, 300), (8, 5), (150, 0), (33, 1)):        # edges of the striping
        pairs.append((rng.integers(0, 4, ql).astype(np.uint8), rng.integers(0, 4, tl).astype(np.uint8)))
    xtra = _xtra(pairs, opt, rng)
    exp = bm2.ksw_align2(pairs, xtra, opt)
    got = bm2.ksw_align2(pairs, xtra, opt, ctx=ctx)
    bad = np.nonzero((exp != got).any(axis=1))[0]
    assert len(bad) == 0, "%d of %d differ; first: task %d (qlen %d, tlen %d, xtra %#x) host %s device %s" % (
        len(bad), len(pairs), bad[0], len(pairs[bad[0]][0]), len(pairs[bad[0]][1]), xtra[bad[0]], exp[bad[0]].tolist(), got[bad[0]].tolist())


def test_device_ksw_align2_paired_end_shape(gpu_ctx_factory):
    # the shape mate rescue produces: 150 bp mates against ~600 bp windows, most with the mate inside
    ctx = gpu_ctx_factory()
    opt = bm2.default_opt()
    rng = np.random.default_rng(12)
    pairs = []
    for i in range(20000):
        t = rng.integers(0, 4, int(rng.integers(400, 800)), dtype=np.uint8)
        if i % 4:
            s = int(rng.integers(0, len(t) - 150))
            q = t[s:s + 150].copy()
            m = rng.random(150) < 0.02
            q[m] = (q[m] + 1) % 4
        else:
            q = rng.integers(0, 4, 150, dtype=np.uint8)
        pairs.append((q, t))
    xtra = [KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 19] * len(pairs)
    exp = bm2.ksw_align2(pairs, xtra, opt)
    got = bm2.ksw_align2(pairs, xtra, opt, ctx=ctx)
    assert (exp == got).all()


def test_sam_pe_with_the_rescue_alignments_on_the_device(gpu_ctx_factory, tmp_path):
    # bm2_sam_pe_dev: the host plans the chunk's rescue alignments, the device runs them against its resident reference, the host
    # replays the pairs -- against the text of the compiled reference and against the all-host path
    import test_sam_tail as T
    fa, r1, r2 = T._pe_case(tmp_path, 61, 3000, sub_rate=0.02, indel_frac=0.2, random_frac=0.03)
    ref, got, pes = T._pe_run(tmp_path, fa, r1, r2, [])
    assert ref == got, T._diff(ref, got)
    host_stats = bm2.sam_rescue_stats()
    ctx = gpu_ctx_factory(fa)
    ref2, got2, pes2 = T._pe_run(tmp_path, fa, r1, r2, [], ctx=ctx)                 # rescue AND CIGAR batches on the device
    assert got2 == got, T._diff(got, got2)
    assert bm2.sam_rescue_stats() == host_stats and host_stats[0] > 500
    planned, used, missed = bm2.sam_cigar_stats()
    assert used >= planned > 3000 and missed == 0, (planned, used, missed)     # hits in the batch, lookups served, numbered hits it lacked


def test_sam_se_with_the_cigar_alignments_on_the_device(gpu_ctx_factory, tmp_path):
    # bm2_sam_se_dev: dry pass on the host, k_gen_cigar (with the retry loop of mem_reg2aln) for every hit the flow asks for, real pass -- against the text of
    # the compiled reference (all record kinds: supplementary, XA, unmapped)
    import test_sam_tail as T
    fa, reads = T._case(tmp_path, 41, 4000)
    rng = np.random.default_rng(3)
    quals = [bytes(rng.integers(35, 74, size=len(r), dtype=np.uint8)) for r in reads]
    fq = str(tmp_path / "r.fq")
    T._write_fastq(fq, reads, quals)
    ref = T._reference_sam(fa, fq)
    names = ["q%d" % i for i in range(len(reads))]
    got = T._ours(fa, reads, names, quals, ctx=gpu_ctx_factory(fa))
    assert ref == got, T._diff(ref, got)
    planned, used, missed = bm2.sam_cigar_stats()
    assert used >= planned > 3000 and missed == 0, (planned, used, missed)     # hits in the batch, lookups served, numbered hits it lacked


@pytest.mark.parametrize("kw", [{}, dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1)])
def test_device_gen_cigar_equals_host(gpu_ctx_factory, tmp_path, kw):
    # k_gen_cigar (cigar.hip) against the host code pinned to the reference's bwa_gen_cigar2 (tests/test_gen_cigar.py)
    import subprocess
    from helpers import ref_binary
    from test_gen_cigar import make_tasks
    from tools import synth
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present (it builds the index)")
    names, ctg, alts = synth.make_genome(17, [120000, 50000], alt_contigs=0, n_repeat_families=3, repeat_len=(200, 1500), copies=(3, 10),
                                         divergence=(0.0, 0.05), n_gaps=2, gap_len=(30, 200))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    opt = bm2.default_opt(**kw)
    tasks = make_tasks(ctg, 5 + len(kw), 5000)
    exp = bm2.gen_cigar(fa, opt, tasks)
    got = bm2.gen_cigar(fa, opt, tasks, ctx=gpu_ctx_factory(fa))
    bad = [i for i, (x, y) in enumerate(zip(exp, got)) if x != y]
    assert not bad, "%d of %d differ; first: task %d %s host %s device %s" % (len(bad), len(tasks), bad[0], tasks[bad[0]][1:], exp[bad[0]], got[bad[0]])


def test_cigar_retry_loop_beyond_the_first_band(gpu_ctx_factory, tmp_path, capfd):
    # mem_reg2aln's retry loop (bwamem.cpp:1748-1766) on the device: the RING kernel runs a task's first try only and hands a task that asks
    # for a wider band to the ROW kernel.  Real hits hardly ever ask; hits whose truesc is raised by hand do (the first band is inferred
    # from truesc, and the alignment's score stays below it): the device text must equal the host twin's (cigar_with_retries).
    import os
    import test_sam_tail as T
    from helpers import oracle_finish_regs
    from tools import oracle, refio
    fa, reads = T._case(tmp_path, 43, 1500)
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt())
    finally:
        ix.close()
    opt = bm2.default_opt()
    regs, reg_off = T._prg_to_regs(exp["REGPRG"], len(ln))
    aln, aln_off = oracle_finish_regs(fa, enc, off, ln, opt, regs, reg_off)
    aln = aln.copy()
    aln["truesc"] += 14                                             # every alignment now looks 14 short of its hit's score
    names = ["q%d" % i for i in range(len(reads))]
    quals = [b"I" * len(r) for r in reads]
    host = bm2.sam_se(fa, enc, off, ln, opt, aln, aln_off, names, quals, None, None)
    os.environ["BM2_TAIL_PROF"] = "1"
    try:
        dev = bm2.sam_se(fa, enc, off, ln, opt, aln, aln_off, names, quals, None, None, ctx=gpu_ctx_factory(fa))
    finally:
        del os.environ["BM2_TAIL_PROF"]
    assert host == dev, T._diff(host, dev)
    err = capfd.readouterr().err
    import re
    m = re.search(r"ring (\d+) \((\d+) of them deferred", err)
    assert m and int(m.group(2)) > 50, err[-600:]                  # the hand-over really happened


def test_s1_batch_resident_between_runs(gpu_ctx_factory):
    # bm2_bsw_upload / bm2_bsw_run / bm2_bsw_download (what `bench.py --workload bsw` times): the results of bm2_bsw, unchanged by a second
    # run over the resident batch, the kernel's own cell counter, and bench.py's synthetic extension tasks against the oracle
    # (last in the suite: the entry points are new)
    from helpers import pack_pairs, random_pairs
    from tools import oracle
    ctx = gpu_ctx_factory()
    opt = bm2.default_opt()
    prm = bm2.sw_params(opt, 5)
    pairs, ref, qer = pack_pairs(bm2, random_pairs(91, 2000))
    exp = ctx.bsw(pairs.copy(), ref, qer, 100, prm)
    ctx.bsw_upload(pairs, ref, qer)
    ms, cells = ctx.bsw_run(100, prm, count_cells=True)
    first = ctx.bsw_download()
    ms2, _ = ctx.bsw_run(100, prm)
    second = ctx.bsw_download()
    assert first.tobytes() == exp.tobytes() and second.tobytes() == exp.tobytes()
    assert cells > 0 and ms2 >= 0.0
    import bench
    l2, l1, h0, q, r, qo, ro = bench.make_extension_pairs(5, 1500)
    sp = np.zeros(len(l2), bm2.SEQPAIR_DT)
    sp["idr"], sp["idq"], sp["id"], sp["len1"], sp["len2"], sp["h0"] = ro[:-1], qo[:-1], np.arange(len(l2)), l1, l2, h0
    ctx.bsw_upload(sp, r, q)
    ctx.bsw_run(100, prm)
    got = ctx.bsw_download()
    oopt = oracle.default_opt()
    for i in range(len(l2)):
        e = oracle.ksw_extend(q[qo[i]:qo[i + 1]], r[ro[i]:ro[i + 1]], oopt, 100, 5, int(h0[i]))
        assert tuple(int(got[i][f]) for f in ("score", "qle", "tle", "gtle", "gscore", "max_off")) == e, i


# ---- the two tail kernels against the REFERENCE itself (oracle/_ref/refdump ksw / cigar: the reference's own ksw_align2 and bwa_gen_cigar2), not
# through the library's host twins

@pytest.mark.parametrize("args,kw", [([], {}), (["-A", "2", "-B", "5", "-O", "7,8", "-E", "2,1"], dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1))])
def test_device_ksw_align2_equals_the_reference(gpu_ctx_factory, tmp_path, args, kw):
    import subprocess
    from helpers import ref_binary
    dump = ref_binary("refdump")
    if dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    opt = bm2.default_opt(**kw)
    pairs = _pairs(31 + len(args), 2500)
    xtra = [KSW_XSUBO | KSW_XSTART | (19 * opt.a) | (KSW_XBYTE if len(q) * opt.a < 250 else 0) for q, t in pairs]
    pf, of = str(tmp_path / "pairs.txt"), str(tmp_path / "out.bin")
    with open(pf, "w") as f:
        for (q, t), x in zip(pairs, xtra):
            f.write("%d %s %s\n" % (x, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
    subprocess.check_call([dump] + args + ["ksw", pf, of], stderr=subprocess.DEVNULL)
    exp = np.fromfile(of, "<i4").reshape(-1, 7)
    got = bm2.ksw_align2(pairs, xtra, opt, ctx=gpu_ctx_factory())
    bad = np.nonzero((exp != got).any(axis=1))[0]
    assert len(bad) == 0, "%d of %d differ; first: pair %d reference %s device %s" % (len(bad), len(pairs), bad[0], exp[bad[0]].tolist(), got[bad[0]].tolist())


def test_device_gen_cigar_equals_the_reference(gpu_ctx_factory, tmp_path):
    import subprocess
    from helpers import ref_binary
    from test_gen_cigar import make_tasks
    from tools import synth
    exe, dump = ref_binary(), ref_binary("refdump")
    if exe is None or dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    names, ctg, alts = synth.make_genome(19, [150000, 60000], alt_contigs=0, n_repeat_families=3, repeat_len=(200, 1500), copies=(3, 10),
                                         divergence=(0.0, 0.05), n_gaps=2, gap_len=(30, 200))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tasks = make_tasks(ctg, 9, 3000)
    tf, of = str(tmp_path / "tasks.txt"), str(tmp_path / "out.bin")
    with open(tf, "w") as f:
        for q, rb, re_, w in tasks:
            f.write("%d %d %d %s\n" % (w, rb, re_, "".join("ACGTN"[c] for c in q)))
    subprocess.check_call([dump, "cigar", fa, tf, of], stderr=subprocess.DEVNULL)
    raw = open(of, "rb").read()
    exp, p = [], 0
    while p < len(raw):
        sc, nc, nm = np.frombuffer(raw, "<i4", 3, p); p += 12
        if nc < 0:
            exp.append((int(sc), int(nm), None, b"")); continue
        ops = [int(x) for x in np.frombuffer(raw, "<u4", nc, p)]; p += 4 * nc
        e = raw.index(b"\0", p); md = raw[p:e]; p += (e - p + 1 + 3) & ~3
        exp.append((int(sc), int(nm), ops, md))
    got = bm2.gen_cigar(fa, bm2.default_opt(), tasks, ctx=gpu_ctx_factory(fa))
    assert len(exp) == len(got) == len(tasks)
    for i, (x, y) in enumerate(zip(exp, got)):
        if x[2] is None:
            assert y[2] is None, "task %d: the reference returns NULL" % i
        else:
            assert x == y, "task %d (w %d, %d..%d): reference %s device %s" % (i, tasks[i][3], tasks[i][1], tasks[i][2], x, y)
