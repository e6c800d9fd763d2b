"""The legs of bench.py that are not the timed loop: synthetic genome + chunks, the compiled reference as checker and CPU baseline (oracle/_ref:
parity gates, `bwa-mem2 mem` timing), the drop-in binaries' timing, the FASTQ -> SAM pipeline of host threads, config 2's workload (bench_bsw) and the
side workloads as processes of their own.  bench.py holds the timed loop and the driver's line (tools/bench_line.py)."""
import json
import os
import queue
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

RANDOM_LINE_GLPS = 55.0        # measured: ~55 G independent 64-B lines/s delivered (tools/ubench/randline.hip)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIG5_READS = ["--reads", "20000"]                       # config 5's chunk inside the default line: what fits beside the index (9 GB of workspaces per 1000 reads; profiles/r05e_ont_scaling.log: 11.9 k reads/s at 10 000 reads, 16.6 k at 20 000)
ONT2D = dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1, pen_clip5=0, pen_clip3=0, min_seed_len=14, min_chain_weight=20,
             split_factor=10.0)      # `-x ont2d`, fastmap.cpp:812-826


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def ref_binary(kind="bwa-mem2"):
    flags = open("/proc/cpuinfo").read()
    for a in (["avx512bw"] if "avx512bw" in flags else []) + (["avx2"] if "avx2" in flags else []) + ["sse41"]:
        p = os.path.join(ROOT, "oracle", "_ref", "%s.%s" % (kind, a))
        if os.path.exists(p):
            return p, a
    return None, None


def contig_lengths(total_bp):
    """Human-like spread of contig sizes summing to total_bp (25 primary contigs)."""
    w = np.array([248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46,
                  50, 156, 57, 16], dtype=np.float64)
    l = np.maximum((w / w.sum() * total_bp).astype(np.int64), 2000)
    return [int(x) for x in l]


def prepare_genome(workdir, mbp, seed):
    from tools import synth
    pre = os.path.join(workdir, "genome_%dmbp_s%d.fa" % (mbp, seed))
    meta = pre + ".contigs.npz"
    memo = []

    def contigs():                                            # loaded on demand: 3 GB that a run with a cached chunk never touches
        if not memo:
            z = np.load(meta, allow_pickle=True)
            memo.append([z["c%d" % i] for i in range(int(z["n"]))])
        return memo[0]
    lens_fn = pre + ".contig_lens.npy"                        # (tools/gen_chunk.py samples reads from the mapped .0123 with these)
    if os.path.exists(pre + ".bwt.2bit.64") and os.path.exists(meta):
        if not os.path.exists(lens_fn):
            np.save(lens_fn, np.array([len(c) for c in contigs()], np.int64))
        return pre, contigs
    import bm2
    t = time.time()
    total = int(mbp * 1e6)
    names, ctg, alts = synth.make_genome(seed, contig_lengths(total), n_repeat_families=max(8, min(mbp, 512)),
                                         repeat_len=(300, 6000), copies=(5, 200), divergence=(0.01, 0.15),
                                         n_gaps=8, gap_len=(100, 5000), alt_contigs=3, alt_len=50000)
    synth.write_fasta(pre, names, ctg)
    synth.write_alt(pre + ".alt", alts)
    log("genome %d Mbp generated in %.1fs; building the index (bm2_index_build: same bytes as `bwa-mem2 index`, all host cores)..."
        % (mbp, time.time() - t))
    t = time.time()
    bm2.index_build(pre, None, 0)
    log("index built in %.1fs" % (time.time() - t))
    np.savez(meta, n=len(ctg), **{"c%d" % i: c for i, c in enumerate(ctg)})
    np.save(lens_fn, np.array([len(c) for c in ctg], np.int64))
    memo.append(ctg)
    return pre, contigs


def pe_chunk(workdir, contigs_fn, seed, n_reads, read_len, tag=""):
    """The timed chunk: n_reads/2 synthetic pairs, mates interleaved as bseq_read_orig delivers PE chunks.  Deterministic in (seed,
    n_reads, read_len), so the array is kept beside the index (the profiling passes of one box re-run this script several times)."""
    from tools import synth
    fn = os.path.join(workdir, "chunk_pe_s%d_n%d_l%d%s.npy" % (seed, n_reads, read_len, tag))
    if os.path.exists(fn):
        return np.load(fn)
    r1, r2 = synth.make_reads_pe(seed, contigs_fn(), n_reads // 2, L=read_len)
    seqs = np.empty((2 * len(r1), read_len), np.uint8)
    seqs[0::2] = r1; seqs[1::2] = r2
    try:
        tmp = "%s.%d.tmp.npy" % (fn, os.getpid())
        np.save(tmp, seqs); os.replace(tmp, fn)
    except OSError:
        pass
    return seqs


THREADS_SOURCE = [None]


def host_threads():
    """threads for the compiled reference: the CPUs this process can really use (cgroup quota), not the hardware threads it can see.  Which of
    the two counts was taken is recorded (THREADS_SOURCE -> cpu_baseline.threads_source): rounds 1-2 ran the baseline on min(cpu_count, 128)
    threads, so their baselines are not comparable with the quota-sized ones since round 3."""
    try:
        import bm2
        THREADS_SOURCE[0] = "cgroup CPU quota (bm2_host_cpus)"
        return max(1, min(bm2.host_cpus(), 128))
    except Exception:                                             # noqa
        THREADS_SOURCE[0] = "os.cpu_count() -- libbm2 could not be loaded: NOT the quota-sized count of the other runs"
        return min(os.cpu_count() or 1, 128)


def run_reference_mem(prefix, fq, extra=(), threads=None, out="/dev/null"):
    """`bwa-mem2 mem` of the compiled reference -> (stderr text, wall seconds) or (None, 0)."""
    exe, isa = ref_binary()
    if exe is None:
        return None, 0.0, None
    threads = threads or host_threads()
    t = time.time()
    p = subprocess.run([exe, "mem", "-t", str(threads), "-K", "100000000", "-o", out] + list(extra) + [prefix] + list(fq),
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        log("reference mem failed:", p.stderr[-500:])
        return None, 0.0, isa
    return p.stderr, time.time() - t, isa


def cpu_baseline(prefix, fq, n_reads_desc, extra=(), out="/dev/null"):
    """Time the compiled reference on a bounded sample of the same workload, all host cores (its SAM goes to `out`: the wide parity gate
    compares it with the library's text for the same reads)."""
    err, wall, isa = run_reference_mem(prefix, fq, extra, out=out)
    return cpu_baseline_from(err, wall, isa, n_reads_desc, extra)


def cpu_baseline_from(err, wall, isa, n_reads_desc, extra=()):
    """The baseline figures from the stderr of a finished `bwa-mem2 mem` run (its own per-chunk and per-kernel clocks)."""
    if err is None:
        return None
    threads = host_threads()
    n_proc, real = 0, 0.0
    for m in re.finditer(r"Processed (\d+) reads in [\d.]+ CPU sec, ([\d.]+) real sec", err):
        n_proc += int(m.group(1)); real += float(m.group(2))
    kern = re.search(r"Total kernel \(smem\+sal\+bsw\) time avg: ([\d.]+)", err)
    kern_s = float(kern.group(1)) if kern else None
    if n_proc == 0 or real <= 0:
        return None
    out = {"value": n_proc / real, "unit": "reads/s", "cores": threads, "threads_present": os.cpu_count(), "threads_source": THREADS_SOURCE[0], "kind": "reference",
           "sample": "%s, same index; bwa-mem2 v2.2.1 %s build, `mem -t %d %s` (%d = the CPUs this process may use: cgroup quota; the host shows %d hardware threads); whole `mem` chunk time "
                     "(seed+chain+extend+pairing+SAM) from its own 'Processed N reads' lines; wall %.1fs"
                     % (n_reads_desc, isa, threads, " ".join(extra), threads, os.cpu_count() or 0, wall)}
    if kern_s:
        out["hot_path_value"] = n_proc / kern_s
        out["hot_path_note"] = "reads / reference's own per-thread-average SMEM+SAL+BSW kernel time (the scope of the top-level `value`)"
    return out


def regs_records(regs, reg_off, lo, hi):
    """device regs of reads [lo, hi) in the record layout of refdump's REGPRG section"""
    from tools import refio
    a, b = int(reg_off[lo]), int(reg_off[hi])
    out = np.zeros(b - a, refio.REG_DT)
    out["read"] = np.repeat(np.arange(hi - lo), np.diff(reg_off[lo:hi + 1]))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        out[f] = regs[f][a:b]
    return out


def alnregs_records(aln, aln_off):
    from tools import refio
    out = np.zeros(len(aln), refio.REG_DT)
    out["read"] = np.repeat(np.arange(len(aln_off) - 1), np.diff(aln_off))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep"):
        out[f] = aln[f]
    return out


def sam_lines(text):
    return [l for l in text.split(b"\n") if l and not l.startswith(b"@")]


def parity_gate(ctx, bm2, prefix, workdir, seqs, regs, reg_off, opt, opt_args, paired, n_sample, tag, n_regs=None):
    """The reference on the first n_sample reads of the timed chunk, same index: regs (refdump) and SAM text (bwa-mem2 mem).  n_regs: the
    stage dumps (refdump: ONE host thread) cover only the first n_regs of them -- long reads cost it a second each -- while `bwa-mem2 mem`
    runs all n_sample on every CPU and every one of their SAM records is compared."""
    from tools import refio, synth
    refdump, isa = ref_binary("refdump")
    exe, _ = ref_binary()
    if refdump is None or exe is None:
        return {"reads": 0, "regs_equal": None, "fin_equal": None, "sam_equal": None, "note": "oracle/_ref is not built on this box"}
    n = n_sample
    nr = min(n_regs or n, n)
    res = {"reads": n, "sample": "first %d reads of the timed chunk (a 512-aligned prefix), reference %s build" % (n, isa) +
                                 ("" if nr == n else "; REGPRG / REGFIN dumps of the first %d, SAM records of all %d" % (nr, n)), "regs_reads": nr}
    # --- regs: REGPRG (device boundary) and REGFIN (after mem_sort_dedup_patch)
    t = time.time()
    rtxt = os.path.join(workdir, "parity_%s.txt" % tag)
    acgtn = np.frombuffer(b"ACGTN", np.uint8)
    with open(rtxt, "wb") as f:
        for s in seqs[:nr]:
            f.write(acgtn[s].tobytes() + b"\n")
    dump = os.path.join(workdir, "parity_%s.bin" % tag)
    p = subprocess.run([refdump] + list(opt_args) + [prefix, rtxt, dump], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        res.update(regs_equal=False, note="refdump failed: " + p.stderr[-300:])
        return res
    d = refio.read_dump(dump)
    got = regs_records(regs, reg_off, 0, nr)
    res["regs"] = int(len(got))
    res["regs_equal"] = bool(len(got) == len(d["REGPRG"]) and got.tobytes() == d["REGPRG"].tobytes())
    res["max_coord"] = int(d["REGPRG"]["re"].max()) if len(d["REGPRG"]) else 0
    res["regs_over_2p32"] = int((d["REGPRG"]["re"] >= (1 << 32)).sum())
    log("parity[%s]: refdump on %d reads in %.1fs: REGPRG equal = %s (%d regs, %d beyond 2^32)"
        % (tag, nr, time.time() - t, res["regs_equal"], len(got), res["regs_over_2p32"]))
    # --- the tail: a19 + pairing + SAM on the same prefix, from FASTQ text as the reference reads it
    t = time.time()
    if paired:
        f1, f2 = os.path.join(workdir, "parity_%s_1.fq" % tag), os.path.join(workdir, "parity_%s_2.fq" % tag)
        synth.write_fastq(f1, seqs[0:n:2], suffix="/1"); synth.write_fastq(f2, seqs[1:n:2], suffix="/2")
        fq = [f1, f2]
    else:
        f1 = os.path.join(workdir, "parity_%s.fq" % tag)
        synth.write_fastq(f1, seqs[:n])
        fq = [f1]
    ref_sam = os.path.join(workdir, "parity_%s.ref.sam" % tag)
    err, wall, isa_mem = run_reference_mem(prefix, fq, opt_args, out=ref_sam)
    if err is None:
        res.update(sam_equal=False, note="reference mem failed")
        return res
    res["_reference_run"] = (err, wall, isa_mem)                 # (popped by the caller: the run doubles as the CPU baseline where it is long enough)
    chunk = bm2.FastqChunk(open(fq[0], "rb").read(), open(fq[1], "rb").read() if paired else None, 0)
    try:
        sub_off = (reg_off[:n + 1] - reg_off[0]).astype(np.int64)
        sub_regs = regs[int(reg_off[0]):int(reg_off[n])]
        aln, aln_off = ctx.finish_regs(chunk, opt, sub_regs, sub_off)
        fin = alnregs_records(aln[:int(aln_off[nr])], aln_off[:nr + 1])
        res["fin_equal"] = bool(len(fin) == len(d["REGFIN"]) and fin.tobytes() == d["REGFIN"].tobytes())
        so = bm2.default_sam_opt(n_threads=0)
        txt = ctx.sam(chunk, opt, so, aln, aln_off, 0, paired).tobytes()
    finally:
        chunk.close()
    mine, ref = sam_lines(txt), sam_lines(open(ref_sam, "rb").read())
    res["sam_records"] = len(ref)
    res["sam_equal"] = bool(mine == ref)
    if not res["sam_equal"]:
        for i, (x, y) in enumerate(zip(mine, ref)):
            if x != y:
                res["first_sam_diff"] = {"line": i, "got": x[:300].decode("latin1"), "exp": y[:300].decode("latin1")}
                break
    log("parity[%s]: a19 equal = %s, SAM equal = %s (%d records; reference mem %.1fs, total %.1fs)"
        % (tag, res["fin_equal"], res["sam_equal"], len(ref), wall, time.time() - t))
    return res


def sam_gate(ctx, bm2, fq, ref_sam, opt, paired, tag):
    """The wide gate: the reads of the FASTQ files `fq` (one chunk) through the library -- parse, device pipeline incl. a19, tail -- against the
    SAM text the compiled reference wrote for the same files (`ref_sam`: the CPU-baseline run's output, or a single-end run of it)."""
    t = time.time()
    chunk = bm2.FastqChunk(open(fq[0], "rb").read(), open(fq[1], "rb").read() if paired else None, 0)
    try:
        ctx.batch_upload_chunk(chunk); ctx.batch_run(opt); ctx.batch_finish(opt)
        aln, aln_off = ctx.batch_download_alnregs()
        txt = ctx.sam(chunk, opt, bm2.default_sam_opt(n_threads=0), aln, aln_off, 0, paired).tobytes()
        n_reads = chunk.n_reads
    finally:
        chunk.close()
    mine, ref = sam_lines(txt), sam_lines(open(ref_sam, "rb").read())
    res = {"reads": n_reads, "sam_records": len(ref), "sam_equal": bool(mine == ref)}
    if not res["sam_equal"]:
        res["got_records"] = len(mine)
        for i, (x, y) in enumerate(zip(mine, ref)):
            if x != y:
                res["first_sam_diff"] = {"line": i, "got": x[:300].decode("latin1"), "exp": y[:300].decode("latin1")}
                break
    log("parity[%s]: %d reads through the library vs the reference's SAM: equal = %s (%d records, %.1fs)" % (tag, n_reads, res["sam_equal"], len(ref), time.time() - t))
    return res


def binding_leg(workdir, prefix, n_chunks=10, n_ref_chunks=2):
    """BASELINE config 3 through the CLI: `bwa-mem2.bm2 mem` (the reference's program -- reader, chunking, writer -- with libbm2 in place of
    mem_process_seqs) on ALL the end-to-end chunks' files (10 M reads), and the unmodified `bwa-mem2.<isa> mem` on the first n_ref_chunks of
    them (same -t / -K; the reference needs a minute for 10 M reads).  Walls from process start to exit (index load in both), the chunks' own
    'Processed N reads' lines (steady state: every chunk but the first, which attaches the library and uploads the replica), the reference's
    own I/O clocks, and the SAM of the reference's chunks against the same records of the binding's output (a prefix: same -K, same chunks)."""
    import hashlib
    exe, isa = ref_binary()
    bm2_exe = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2.%s" % isa)       # the binding built with the ISA of the reference binary it is timed beside
    if not os.path.exists(bm2_exe):
        bm2_exe = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2")
    f1 = [os.path.join(workdir, "e2e_%d_1.fq" % i) for i in range(n_chunks)]
    f2 = [os.path.join(workdir, "e2e_%d_2.fq" % i) for i in range(n_chunks)]
    have = [i for i in range(n_chunks) if os.path.exists(f1[i]) and os.path.exists(f2[i])]
    if exe is None or not os.path.exists(bm2_exe) or len(have) < 2 or have != list(range(len(have))):
        return {"skipped": "oracle/_ref/bwa-mem2.bm2 or the chunk files are not there"}
    n_chunks = len(have)
    n_ref_chunks = min(n_ref_chunks, n_chunks)
    threads = host_threads()
    files = {}
    for tag, k in (("all", n_chunks), ("ref", n_ref_chunks)):
        r1, r2 = os.path.join(workdir, "bind_%s_1.fq" % tag), os.path.join(workdir, "bind_%s_2.fq" % tag)
        for dst, srcs in ((r1, f1[:k]), (r2, f2[:k])):
            with open(dst, "wb") as o:
                for f in srcs:
                    o.write(open(f, "rb").read())
        files[tag] = (r1, r2)
    for f in f1[:n_chunks] + f2[:n_chunks]:
        os.remove(f)

    def run(binary, fq, out_sam, n_lines=None):
        t = time.time()
        p = subprocess.run([binary, "mem", "-t", str(threads), "-K", "150000000", "-o", out_sam, prefix, fq[0], fq[1]], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError("%s failed: %s" % (os.path.basename(binary), p.stderr[-400:]))
        wall = time.time() - t
        h = hashlib.md5()
        n = 0
        with open(out_sam, "rb") as f:
            for line in f:
                if line.startswith(b"@PG"):
                    continue
                if n_lines is not None and n >= n_lines and not line.startswith(b"@"):
                    break
                h.update(line); n += not line.startswith(b"@")
        chunks = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"Processed (\d+) reads in (?:[\d.]+ CPU sec, )?([\d.]+) real sec", p.stderr)]
        prof = {}
        for key, pat in (("reading_reads_s", r"Reading IO time \(reads\) avg: ([\d.]+)"), ("writing_sam_s", r"Writing IO time \(SAM\) avg: ([\d.]+)"),
                         ("index_read_s", r"Index read time avg: ([\d.]+)"), ("mem_process_seq_s", r"MEM_PROCESS_SEQ\(\)[^:]*: ([\d.]+)"), ("overall_s", r"Overall time \(sec\)[^:]*: ([\d.]+)")):
            m = re.search(pat, p.stderr)
            if m:
                prof[key] = float(m.group(1))
        return wall, h.hexdigest(), n, chunks, prof

    w_ref, md_ref, n_ref, ch_ref, prof_ref = run(exe, files["ref"], os.path.join(workdir, "bind_ref.sam"))
    w_bm2, md_all, n_all, ch_bm2, prof_bm2 = run(bm2_exe, files["all"], os.path.join(workdir, "bind_bm2.sam"))
    # the records of the reference's chunks are a prefix of the binding's output (same -K): the md5 of that prefix
    h = hashlib.md5(); n = 0
    with open(os.path.join(workdir, "bind_bm2.sam"), "rb") as f:
        for line in f:
            if line.startswith(b"@PG"):
                continue
            if not line.startswith(b"@"):
                if n >= n_ref:
                    break
                n += 1
            h.update(line)
    md_prefix = h.hexdigest()
    for fn in ("bind_ref.sam", "bind_bm2.sam"):
        try:
            os.remove(os.path.join(workdir, fn))
        except OSError:
            pass
    for r1, r2 in files.values():
        os.remove(r1); os.remove(r2)
    reads_all, reads_ref = sum(c[0] for c in ch_bm2), sum(c[0] for c in ch_ref)
    steady = ch_bm2[1:] if len(ch_bm2) > 1 else ch_bm2
    res = {"reads": reads_all, "chunks": len(ch_bm2), "threads": threads, "chunk_bases": 150000000, "bm2_wall_s": w_bm2, "reads_per_s_bm2_wall": reads_all / w_bm2 if w_bm2 > 0 else None,
           "bm2_chunk_real_s": [c[1] for c in ch_bm2],
           "reads_per_s_bm2_steady_chunks": sum(c[0] for c in steady) / sum(c[1] for c in steady) if steady and sum(c[1] for c in steady) > 0 else None,
           "bm2_profile": prof_bm2,
           "binding_binary": os.path.basename(bm2_exe), "reference": "bwa-mem2.%s mem" % isa, "reference_reads": reads_ref, "reference_wall_s": w_ref, "reference_chunk_real_s": [c[1] for c in ch_ref],
           "reads_per_s_reference_chunks": reads_ref / sum(c[1] for c in ch_ref) if ch_ref and sum(c[1] for c in ch_ref) > 0 else None, "reference_profile": prof_ref,
           "sam_records_compared": n_ref, "sam_records_bm2": n_all, "sam_equal": bool(md_ref == md_prefix and n == n_ref),
           "scope": "bm2: %d chunks from two FASTQ files to a SAM file, process start to exit (17 GB index load and the replica's upload included); per-chunk "
                    "times are mem_process_seqs' own (the reference's reader and writer run around it: bm2_profile); the reference ran the first %d "
                    "chunks, its records are compared with the same records of the binding's output" % (len(ch_bm2), n_ref_chunks)}
    log("binding: %d reads in %d chunks %.1f s (steady chunks %.2f M reads/s), reference %d reads %.1f s, SAM of the reference's chunks equal = %s"
        % (reads_all, len(ch_bm2), w_bm2, (res["reads_per_s_bm2_steady_chunks"] or 0) / 1e6, reads_ref, w_ref, res["sam_equal"]))
    return res


def s1_binding_leg(workdir, prefix):
    """BASELINE config 2 as it is worded -- seeding and chaining on the host, only the banded SW (seam S1) on the GPU, inside the reference's own
    program: `bwa-mem2.bm2s1 mem` (oracle/_ref, integration/bm2_bsw_binding.cpp) beside `bwa-mem2.<isa> mem` on the same single-end reads,
    same -t: chunk rates from their own 'Processed N reads' lines (index load excluded), wall from start to exit, SAM compared."""
    import hashlib
    exe, isa = ref_binary()
    s1 = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2s1.%s" % isa)
    if not os.path.exists(s1):
        s1 = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2s1")
    fq = os.path.join(workdir, "cpu_1.fq")
    if exe is None or not os.path.exists(s1) or not os.path.exists(fq) or not os.path.exists(prefix + ".bwt.2bit.64"):
        return {"skipped": "oracle/_ref/bwa-mem2.bm2s1, the index or the reads of the main run (cpu_1.fq) are not there"}
    threads = host_threads()
    # BASELINE config 2 names 1 M single-end reads: the CPU baseline's mates, both files, twice over (the records of a single-end run do not
    # depend on each other beyond the chunk's 512-read blocks)
    fq2 = os.path.join(workdir, "cpu_2.fq")
    fq_all = os.path.join(workdir, "s1_reads.fq")
    parts_in = [f for f in (fq, fq2) if os.path.exists(f)]
    with open(fq_all, "wb") as o:
        n_lines = 0
        for rep in range(2):
            for f in parts_in:
                b = open(f, "rb").read()
                o.write(b); n_lines += b.count(b"\n")
    fq = fq_all
    res = {"reads_file": "the CPU baseline's mates (%s) as single-end reads, twice over: %d reads" % (" + ".join(os.path.basename(f) for f in parts_in), n_lines // 4), "threads": threads}
    md = {}
    prof_pat = (("mem_process_seqs_s", r"MEM_PROCESS_SEQ\(\)[^:]*: ([\d.]+)"), ("kernels_s", r"Total kernel \(smem\+sal\+bsw\) time avg: ([\d.]+)"),
                ("smem_s", r"SMEM compute avg: ([\d.]+)"), ("sal_s", r"SAL compute avg: ([\d.]+)"), ("bsw_s", r"BSW time, avg: ([\d.]+)"), ("worker_sam_s", r"WORKER_SAM avg: ([\d.]+)"))
    # (Both programs with the same -t.  Measured in round 5, profiles/r05g_*: leaving two CPUs of the quota to the HIP runtime's threads -- the binding with
    #  -t 14 -- is WORSE, 7.0-7.2 s per chunk against 6.45 with -t 16: the fourteen threads' seeding takes what the two would have done.)
    t_s1 = threads
    res["threads_bm2s1"] = t_s1
    for tag, binary in (("reference", exe), ("bm2s1", s1)):
        out_sam = os.path.join(workdir, "s1_%s.sam" % tag)
        t = time.time()
        p = subprocess.run([binary, "mem", "-t", str(threads if tag == "reference" else t_s1), "-K", "100000000", "-o", out_sam, prefix, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        wall = time.time() - t
        if p.returncode != 0:
            return {"error": "%s failed: %s" % (os.path.basename(binary), p.stderr[-300:])}
        n_proc, real = 0, 0.0
        for m in re.finditer(r"Processed (\d+) reads in [\d.]+ CPU sec, ([\d.]+) real sec", p.stderr):
            n_proc += int(m.group(1)); real += float(m.group(2))
        h = hashlib.md5(); n = 0
        with open(out_sam, "rb") as f:
            for line in f:
                if not line.startswith(b"@PG"):
                    h.update(line); n += not line.startswith(b"@")
        os.remove(out_sam)
        md[tag] = (h.hexdigest(), n)
        prof = {}
        for key, pat in prof_pat:                                   # the program's own clocks (per-thread averages), printed at exit
            m = re.search(pat, p.stderr)
            if m:
                prof[key] = float(m.group(1))
        res[tag] = {"wall_s": wall, "reads": n_proc, "chunk_real_s": real, "reads_per_s_chunks": n_proc / real if real > 0 else None, "own_clocks": prof}
        m = re.search(r"\[bm2s1\] (\d+) SeqPairs in (\d+) device batches(?: from (\d+) calls)?", p.stderr)
        if m:
            res[tag]["seqpairs"], res[tag]["device_batches"] = int(m.group(1)), int(m.group(2))
            if m.group(3):
                res[tag]["calls"] = int(m.group(3))
        m = re.search(r"\[bm2s1\] per call ([\d.]+) ms \(all threads' calls: ([\d.]+) s\); per device batch: gather ([\d.]+) ms, bm2_bsw ([\d.]+) ms, scatter ([\d.]+) ms", p.stderr)
        if m:                                                       # the binding's own clocks: where a call's time goes (integration/bm2_bsw_binding.cpp)
            res[tag]["seam_clock"] = {"ms_per_call": float(m.group(1)), "calls_s_all_threads": float(m.group(2)), "gather_ms_per_batch": float(m.group(3)),
                                      "bm2_bsw_ms_per_batch": float(m.group(4)), "scatter_ms_per_batch": float(m.group(5))}
    try:
        os.remove(fq_all)
    except OSError:
        pass
    res["sam_equal"] = bool(md["reference"] == md["bm2s1"])
    res["sam_records"] = md["reference"][1]
    rp = res["reference"]["own_clocks"]
    if rp.get("bsw_s") and rp.get("mem_process_seqs_s"):
        share = rp["bsw_s"] / rp["mem_process_seqs_s"]
        res["seam_share"] = {"bsw_share_of_the_reference_chunk_time": share, "floor_of_bm2s1_over_reference": 1.0 - share,
                             "note": "seam S1 is this share of the reference's own chunk time (its per-thread clocks: the rest is SMEM + SAL on the host, chaining, "
                                     "SAM); a drop-in for S1 alone cannot take the chunk below (1 - share) of the reference's time however fast the kernel is"}
    log("S1 binding: reference %.1f s, bwa-mem2.bm2s1 %.1f s, SAM equal = %s" % (res["reference"]["wall_s"], res["bm2s1"]["wall_s"], res["sam_equal"]))
    return res


def end_to_end(ctx, bm2, texts, opt, paired, n_threads, n_tail=None, limit_s=None, n_dev=None, n_warm=None):
    """FASTQ text -> SAM text over the chunks `texts` = [(bytes1, bytes2 | None)] as a pipeline of host threads: the reader
    (bm2_fastq_parse_mt), n_dev device workers (H2D, seeding .. extension, mem_sort_dedup_patch, D2H; each with a context of its own on
    the shared index replica, chunk i on worker i % n_dev, so the copies and the latency-bound kernels of one chunk overlap the kernels
    of the next) and n_tail tail workers (pairing, rescue + CIGAR batches on the device through further contexts, SAM text; chunk i on
    worker i % n_tail).  The first n_warm chunks go through the SAME threads untimed (workspaces, the library's per-thread worker pools
    and buffers, the output buffers' pages); the pipeline drains, then the clock starts and all of `texts` follow.  A chunk's text is
    complete before it is counted; the tie-breaking hash of a read is seeded with its number in the input (n_before)."""
    hw = bm2.host_cpus()                                         # CPUs this process can really use (cgroup quota), not the hardware threads it sees
    n_tail = int(os.environ.get("BM2_E2E_TAILS", n_tail if n_tail is not None else (3 if hw >= 12 else 2)))
    n_dev = max(1, int(os.environ.get("BM2_E2E_DEVS", n_dev or 2)))
    if os.environ.get("BM2_E2E_LIMIT_S"):                        # (the host emulator needs minutes where the GPU needs milliseconds)
        limit_s = float(os.environ["BM2_E2E_LIMIT_S"])
    n_warm = min(len(texts), max(n_dev, n_tail) if n_warm is None else n_warm)
    work = list(texts[:n_warm]) + list(texts)                    # (warm-up chunks are the first timed ones again: another pass)
    tails = [bm2.Context(share=ctx) for _ in range(n_tail)]
    devs = [ctx] + [bm2.Context(share=ctx) for _ in range(n_dev - 1)]
    if os.environ.get("BM2_E2E_TAIL_PRIO", "1") != "0" and hasattr(tails[0], "set_stream_priority"):
        for c in tails:                                           # the tail's short batches go first: their chunk leaves the pipeline sooner
            c.set_stream_priority(1)
    # the stages' thread counts add up to the CPUs the process may use: beyond that the threads do not run in parallel, they get the process
    # throttled (measured on the MI355X box: 256 hardware threads visible, quota 16 -- profiles/r03c_cgroup.txt)
    # (the stages do not all compute at once -- a tail worker waits for its device batches, the reader for a free queue slot -- so the
    #  counts add up to somewhat more than the CPUs: the split below is the best of profiles/r03f's variants)
    n_parse = int(os.environ.get("BM2_E2E_PARSE_THREADS", max(1, min((3 * hw) // 8, 32))))
    # (threads per tail worker: a worker's threads sleep through its device batches -- rescue SW, CIGAR -- and while the writer copies, so the workers
    #  together hold 4/3 of the CPUs: with three workers on the 16-CPU box 5 threads each gave 10.9-11.0 M reads/s over 100 chunks, 6: 12.0-12.3, 7: 12.7,
    #  8: 12.5; four workers x 6: 12.3, two x 9: 10.5 -- profiles/r06j_*, r06k_*)
    so = bm2.default_sam_opt(n_threads=int(os.environ.get("BM2_E2E_TAIL_THREADS", n_threads or max((hw * 4 // 3) // n_tail, 1))))
    q_parsed, q_hits = [queue.Queue(maxsize=2) for _ in range(n_dev)], [queue.Queue(maxsize=1) for _ in range(n_tail)]
    busy, dyn_threads = [0], os.environ.get("BM2_E2E_DYN_THREADS", "1") != "0"
    last = [None] * n_tail
    free_pins = queue.Queue()
    for _ in range(n_dev + 2 * n_tail):
        free_pins.put([None])
    stage, err, lock, compute = {}, [], threading.Lock(), threading.Lock()
    done = [0] * len(work)
    devs_left, n_done = [n_dev], [0]
    warm_done, go = threading.Event(), threading.Event()
    if n_warm == 0:
        warm_done.set()

    def add(k, dt):
        with lock:
            stage[k] = stage.get(k, 0.0) + dt

    roles = {}                                                    # native thread id -> stage worker (for the per-thread CPU table)

    def name_thread(role, comm):
        roles[threading.get_native_id()] = role
        try:                                                      # (the library's pool workers are named after the thread they work for: host_pool.h)
            import ctypes
            ctypes.CDLL(None).prctl(15, comm.encode()[:15], 0, 0, 0)      # PR_SET_NAME
        except Exception:                                         # noqa
            pass

    def thread_cpu():
        """{tid: (comm, CPU seconds)} of every thread of this process (Linux: /proc/self/task/*/stat)"""
        out, tck = {}, os.sysconf("SC_CLK_TCK")
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    with open("/proc/self/task/%s/stat" % tid) as f:
                        st = f.read()
                    comm = st[st.index("(") + 1:st.rindex(")")]
                    fld = st[st.rindex(")") + 2:].split()
                    out[int(tid)] = (comm, (int(fld[11]) + int(fld[12])) / tck)
                except (OSError, ValueError, IndexError):
                    pass
        except OSError:
            pass
        return out

    def reader():
        name_thread("reader (parse calls)", "bm2-parse")
        try:
            n_before = 0
            for i, (t1, t2) in enumerate(work):
                if i == n_warm:
                    go.wait()                                     # the warm-up chunks have left the pipeline; the clock runs from here
                    n_before = 0
                # (a memory-bound scan.  The first chunk of a run has the host to itself: it is parsed on every CPU)
                t = time.perf_counter(); ch = bm2.FastqChunk(t1, t2, hw if i in (0, n_warm) else n_parse); add("parse", time.perf_counter() - t)
                q_parsed[i % n_dev].put((i, ch, n_before))
                n_before += ch.n_reads
        except Exception as e:                                    # noqa
            err.append(e)
        for q in q_parsed:
            q.put(None)

    def device(k):
        name_thread("device worker", "bm2-device")
        c = devs[k]
        try:
            while True:
                it = q_parsed[k].get()
                if it is None:
                    break
                i, ch, n_before = it
                t = time.perf_counter(); c.batch_upload_chunk(ch); add("h2d", time.perf_counter() - t)
                with compute:                                     # one chunk's seeding .. extension at a time: two of them side by side only slow each other
                    t = time.perf_counter(); c.batch_run(opt); add("device", time.perf_counter() - t)      # down (measured); the workers overlap copies with kernels
                    t = time.perf_counter(); c.batch_finish(opt); add("a19", time.perf_counter() - t)
                pin = free_pins.get()                             # a page-locked hit buffer from the pool (its last reader, a tail worker, has returned it)
                t = time.perf_counter()
                if pin[0] is None or len(pin[0].a) < 3 * ch.n_reads:
                    if pin[0] is not None:
                        pin[0].close()
                    pin[0] = bm2.Pinned(max(3 * ch.n_reads, 1 << 16), bm2.ALNREG_DT)
                aln, aln_off = c.batch_download_alnregs(out=pin[0].a)
                add("d2h", time.perf_counter() - t)
                q_hits[i % n_tail].put((i, ch, aln, aln_off, n_before, pin))
        except Exception as e:                                    # noqa
            err.append(e)
        with lock:
            devs_left[0] -= 1
            last = devs_left[0] == 0
        if last:
            for q in q_hits:
                q.put(None)

    def tail(k):
        name_thread("tail worker", "bm2-tail")
        buf = None
        try:
            while True:
                it = q_hits[k].get()
                if it is None:
                    break
                i, ch, aln, aln_off, n_before, pin = it
                if buf is None:
                    buf = np.empty(max(1 << 20, int(3 * (int(ch.f.n_bases) + 200 * ch.n_reads))), np.uint8)
                # threads of this call: the CPUs divided by the tail workers busy right now (all three in the steady state; the last chunks of a
                # run, with the other workers idle, take the whole host and leave the pipeline sooner)
                with lock:
                    busy[0] += 1
                    mine = max(so.n_threads, hw // busy[0]) if dyn_threads else so.n_threads
                t = time.perf_counter()
                try:
                    txt = tails[k].sam(ch, opt, bm2.default_sam_opt(n_threads=mine), aln, aln_off, n_before, paired, out=buf)
                finally:
                    with lock:
                        busy[0] -= 1
                add("tail", time.perf_counter() - t)
                done[i] = (len(txt), ch.n_reads, time.perf_counter())
                last[k] = (i, len(txt), n_before, buf)            # (this worker's latest chunk: its text stays in `buf` until the next one)
                aln = None
                free_pins.put(pin)
                ch.close()
                with lock:
                    n_done[0] += 1
                    if n_done[0] == n_warm:
                        warm_done.set()
        except Exception as e:                                    # noqa
            err.append(e)

    th = [threading.Thread(target=reader, daemon=True)] + [threading.Thread(target=device, args=(k,), daemon=True) for k in range(n_dev)] + \
         [threading.Thread(target=tail, args=(k,), daemon=True) for k in range(n_tail)]
    t_begin = time.perf_counter()

    def expired():
        return bool(limit_s) and time.perf_counter() - t_begin > limit_s

    for t in th:
        t.start()
    while not warm_done.wait(0.05):                               # (short waits: an error in one stage must not leave the others waiting on a queue)
        if err or expired():
            break
    if err:
        raise err[0]                                              # (the stages are daemon threads: whatever still waits on a queue goes with the process)
    if not warm_done.is_set():
        raise TimeoutError("end-to-end leg: warm-up not finished after %.0f s (stages so far: %s)" % (limit_s, {k: round(v, 1) for k, v in stage.items()}))
    with lock:
        stage.clear()
    t0 = time.perf_counter()
    cpu0 = time.process_time()                                   # CPU seconds of every thread of this process (the stages' workers and the library's pools)
    tc0 = thread_cpu()
    tc_seen, tc_next = dict(tc0), [time.perf_counter() + 0.5]     # (threads leave with their stage: the table keeps the last reading of every thread, taken twice a second)
    go.set()
    for t in th:
        while t.is_alive():
            t.join(0.05)
            if time.perf_counter() >= tc_next[0]:
                tc_seen.update(thread_cpu()); tc_next[0] = time.perf_counter() + 0.5
            if err or expired():
                break
        if err:
            break
        if t.is_alive():
            raise TimeoutError("end-to-end leg not finished after %.0f s (stages so far: %s)" % (limit_s, {k: round(v, 1) for k, v in stage.items()}))
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    tc_seen.update(thread_cpu())
    tc1 = tc_seen
    by_role = {}                                                  # CPU seconds per chunk by kind of thread: stage workers by role, the library's pools ("bm2-pool"), the rest by name
    for tid, (comm, sec) in tc1.items():
        d_sec = sec - tc0.get(tid, (comm, 0.0))[1]
        role = roles.get(tid) or ("main thread" if tid == os.getpid() else comm)
        r = by_role.setdefault(role, [0, 0.0]); r[0] += 1; r[1] += d_sec
    if err:
        raise err[0]
    for c in tails + devs[1:]:
        c.close()
    while not free_pins.empty():
        pin = free_pins.get()
        if pin[0] is not None:
            pin[0].close()
    # the text of the run's LAST chunk (still in its worker's buffer) against the same chunk put through ONE context, stage after stage, on
    # this thread: the pipeline (several contexts on one replica, worker threads, pinned buffers in rotation) must not change a byte
    check = None
    if os.environ.get("BM2_E2E_CHECK", "1") != "0" and any(x is not None for x in last):
        i, n_txt, n_before, buf = max((x for x in last if x is not None), key=lambda x: x[0])
        t1, t2 = work[i]
        ch = bm2.FastqChunk(t1, t2, 0)
        try:
            ctx.batch_upload_chunk(ch); ctx.batch_run(opt); ctx.batch_finish(opt)
            aln, aln_off = ctx.batch_download_alnregs()
            ref_txt = ctx.sam(ch, opt, bm2.default_sam_opt(n_threads=0), aln, aln_off, n_before, paired)
            check = {"chunk": int(i - n_warm), "bytes": int(n_txt), "equal_to_serial_run": bool(n_txt == len(ref_txt) and np.array_equal(buf[:n_txt], ref_txt))}
        finally:
            ch.close()
    timed = done[n_warm:]
    out_bytes = sum(d[0] for d in timed); n_reads = sum(d[1] for d in timed)
    nch = max(len(texts), 1)
    # the rate between the completion of the timed region's 4th chunk and its last one: what a long run converges to (`value` holds the
    # fill of the empty pipeline -- the first chunk's parse, copies, device stages and tail, ~0.45 s -- and its drain, spread over nch chunks)
    steady = None
    fin = sorted(d[2] for d in timed if isinstance(d, tuple) and len(d) > 2)
    if len(fin) >= 8:
        steady = {"chunks": len(fin) - 4, "ms_per_chunk": (fin[-1] - fin[3]) / (len(fin) - 4) * 1e3,
                  "reads_per_s": (n_reads / len(fin)) * (len(fin) - 4) / (fin[-1] - fin[3]) if fin[-1] > fin[3] else None}
    return {"value": n_reads / dt, "unit": "reads/s", "reads": n_reads, "chunks": len(texts), "distinct_chunks": len(set(id(t[0]) for t in texts)),
            "steady_state": steady, "warmup_chunks": n_warm, "wall_s": dt, "sam_bytes": out_bytes,
            "host_cpus": hw, "host_threads_visible": os.cpu_count(), "parse_threads": n_parse, "device_workers": n_dev, "tail_workers": n_tail, "threads_per_tail_worker": so.n_threads,
            "stage_ms_per_chunk": {k: v / nch * 1e3 for k, v in stage.items()}, "chunk_check": check,
            # what the host side costs: CPU seconds of the whole process per timed chunk, and the chunk time that alone would allow on this host's cores
            "host_cpu_s_per_chunk_by_thread_kind": {k: {"threads": v[0], "cpu_s_per_chunk": round(v[1] / nch, 4)} for k, v in sorted(by_role.items(), key=lambda kv: -kv[1][1]) if v[1] / nch >= 0.0005},
            "host_cpu_s_per_chunk": cpu_s / nch, "host_cpu_bound_ms_per_chunk": cpu_s / nch / max(hw, 1) * 1e3, "ms_per_chunk": dt / nch * 1e3,
            "scope": "FASTQ text in host memory -> bm2_fastq_parse_mt | H2D -> device pipeline incl. mem_sort_dedup_patch (a19) -> D2H | pairing / "
                     "mate rescue / CIGAR (device batches) / SAM text in host memory; one host thread per stage worker (%d device workers on contexts "
                     "sharing the index replica, %d tail workers), stages of consecutive chunks overlap; the warm-up chunks pass through the same "
                     "threads before the clock starts; `stage_ms_per_chunk` is the time a chunk spends in a stage on its worker; file I/O excluded"
                     % (n_dev, n_tail)}


def side_workload(a, name, extra, limit_s):
    """BASELINE configs 5 / 2 inside the default run: the same script with --workload <name> as a process of its own on the same index
    files (a failure or a hang there cannot take the main line with it) -> the fields of its JSON line that describe that workload."""
    full = os.path.join(a.workdir, "bench_full_%s.json" % name)
    try:
        os.remove(full)
    except OSError:
        pass
    cmd = [sys.executable, BENCH_PY, "--workload", name, "--gpus", "1", "--genome-mbp", str(a.genome_mbp), "--workdir", a.workdir, "--full-json", full,
           "--no-e2e", "--no-binding", "--no-side-workloads", "--budget-s", str(int(limit_s))] + [str(x) for x in extra]
    t = time.time()
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s + 30)
    except subprocess.TimeoutExpired:
        return {"error": "not finished after %.0f s" % (limit_s + 30)}
    for l in p.stderr.split("\n"):
        if l.startswith("[bench]"):
            log("[%s]" % name, l[8:])
    try:                                                         # the process's FULL record (its stdout line is the compact one)
        d = json.load(open(full))
    except (OSError, ValueError):
        line = [l for l in p.stdout.split("\n") if l.startswith("{")]
        if not line:
            return {"error": "exit code %d, no JSON line; stderr tail: %s" % (p.returncode, p.stderr[-300:])}
        d = json.loads(line[-1])
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "stage_ms_per_step", "dominant_stage", "work_per_read", "roofline", "extend_kernel",
            "chain_kernel", "parity", "cpu_baseline", "pairs_per_s", "s1_binding")
    out = {k: d[k] for k in keep if k in d}
    out["exit_code"] = p.returncode
    out["wall_s"] = time.time() - t
    return out


TASKS_PER_READ = 2.874          # extension tasks per 150 bp read of the pe150 workload (work_per_read.sw_tasks of its bench line)


def make_extension_pairs(seed, n, max_q=131, sub_rate=0.012):
    """n synthetic (query, target, h0) extension tasks shaped like those mem_chain2aln builds for 150 bp reads (bwamem.cpp:2229-2418): the
    query is what is left of the read beside a seed (1 .. 131 bases), the target the reference beside the seed's hit plus the room the
    band allows, h0 the seed's score; the query is the target's head with substitutions, one task in seven diverges half way (Z-drop).
    Vectorised: -> (len2[n], len1[n], h0[n], qer flat, ref flat) with pair i's bases at the prefix sums of the lengths."""
    rng = np.random.default_rng(seed)
    len2 = rng.integers(1, max_q + 1, size=n).astype(np.int32)
    len1 = (len2 + rng.integers(0, 45, size=n)).astype(np.int32)
    h0 = (150 - len2 - rng.integers(0, np.maximum(150 - len2 - 18, 1))).clip(19, 150).astype(np.int32)
    roff = np.concatenate([[0], np.cumsum(len1, dtype=np.int64)]); qoff = np.concatenate([[0], np.cumsum(len2, dtype=np.int64)])
    ref = rng.integers(0, 4, size=int(roff[-1]), dtype=np.uint8)
    src = np.arange(int(qoff[-1]), dtype=np.int64) - np.repeat(qoff[:-1], len2) + np.repeat(roff[:-1], len2)      # query base k of pair i = target base k
    qer = ref[src]
    flip = rng.random(len(qer)) < sub_rate
    qer[flip] = (qer[flip] + rng.integers(1, 4, size=int(flip.sum()), dtype=np.uint8)) & 3
    div = np.flatnonzero(rng.random(n) < 1.0 / 7)                 # divergent tails: random bases from the middle of the query on
    if len(div):
        pos = np.arange(int(qoff[-1]), dtype=np.int64) - np.repeat(qoff[:-1], len2)
        tail = np.zeros(n, bool); tail[div] = True
        m = np.repeat(tail, len2) & (pos >= np.repeat(len2 // 2, len2))
        qer[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
    return len2, len1, h0, qer, ref, qoff, roff


def bench_bsw(a, bm2, torch, dist_util, rank, world, local, emu, seed):
    """BASELINE.json config 2: the banded-SW kernel alone (S1, one SeqPair per wavefront), the batch resident in HBM."""
    from tools import oracle
    n = a.bsw_pairs
    len2, len1, h0, qer, ref, qoff, roff = make_extension_pairs(dist_util.shard_seed(seed, rank), n)
    pairs = np.zeros(n, bm2.SEQPAIR_DT)
    pairs["idr"], pairs["idq"], pairs["id"] = roff[:-1], qoff[:-1], np.arange(n)
    pairs["len1"], pairs["len2"], pairs["h0"] = len1, len2, h0
    opt = bm2.default_opt()
    w, end_bonus = 100, 5
    prm = bm2.sw_params(opt, end_bonus)
    ctx = bm2.Context(local)
    ctx.bsw_upload(pairs, ref, qer)
    _, cells = ctx.bsw_run(w, prm, count_cells=True)             # untimed: the cell counter costs an atomic per pair
    for _ in range(a.warmup):
        ctx.bsw_run(w, prm)
    torch.cuda.synchronize()
    dist_util.barrier(world)
    t0 = time.perf_counter()
    k_ms = 0.0
    for _ in range(a.steps):
        ms, _ = ctx.bsw_run(w, prm)                              # returns after the stream has drained
        k_ms += ms
    torch.cuda.synchronize()
    dist_util.barrier(world)
    dt = dist_util.max_over_ranks(time.perf_counter() - t0, world, "cpu" if emu else "cuda")
    got = ctx.bsw_download()
    ctx.close()
    if rank != 0:
        return 0
    steps = max(a.steps, 1)
    k_ms /= steps
    # parity + CPU baseline: the REFERENCE's own kernels (oracle/_ref/refdump.<isa> bswtime: BandedPairWiseSW::getScores8 / getScores16 /
    # scalarBandedSWAWrapper, pairs filed and ordered as the reference files and orders them, one BandedPairWiseSW object and one slice per
    # thread) on ALL pairs of the step, on the CPUs this process may use; every pair's six outputs are compared
    refdump, isa = ref_binary("refdump")
    threads = host_threads()
    cb, par = None, None
    fields = ("score", "qle", "tle", "gtle", "gscore", "max_off")
    if refdump is not None:
        fn_in, fn_out = os.path.join(a.workdir, "bsw_pairs_r%d.bin" % rank), os.path.join(a.workdir, "bsw_ref_r%d.bin" % rank)
        with open(fn_in, "wb") as f:
            np.array([n], np.int32).tofile(f); len2.astype(np.int32).tofile(f); len1.astype(np.int32).tofile(f); h0.astype(np.int32).tofile(f)
            qer.tofile(f); ref.tofile(f)
        best = None
        for _ in range(2):                                       # (the first run also pages the files in)
            p = subprocess.run([refdump, "bswtime", str(w), str(end_bonus), str(threads), fn_in, fn_out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if p.returncode != 0:
                log("refdump bswtime failed:", p.stderr[-300:])
                best = None
                break
            r = json.loads(p.stdout.strip().split("\n")[-1])
            if best is None or r["seconds"] < best["seconds"]:
                best = r
        if best is not None:
            exp = np.fromfile(fn_out, np.int32).reshape(-1, 8)
            g = np.stack([np.asarray(got[f], np.int64) for f in fields], axis=1)
            e = exp[:, :6].astype(np.int64)
            # gtle means something only while gscore > 0: the reference's vector kernels keep stepping a finished pair (tests/test_bsw_reference.py)
            cmp_cols = np.ones((len(e), 6), bool); cmp_cols[e[:, 4] <= 0, 3] = False
            bad_rows = np.flatnonzero(((g != e) & cmp_cols).any(axis=1))
            par = {"pairs": int(n), "sample": "ALL pairs of the step against the reference's getScores8 / getScores16 / scalarBandedSWAWrapper (%s build, refdump bswtime); six outputs, "
                                               "gtle where gscore > 0" % isa, "pairs_equal": len(bad_rows) == 0, "mismatches": int(len(bad_rows)),
                   "class8": best["class8"], "class16": best["class16"], "class32": best["class32"]}
            if len(bad_rows):
                i = int(bad_rows[0])
                par["first_diff"] = {"pair": i, "got": [int(x) for x in g[i]], "exp": [int(x) for x in e[i]], "len2": int(len2[i]), "len1": int(len1[i]), "h0": int(h0[i])}
            cb = {"value": n / best["seconds"] / TASKS_PER_READ, "unit": "reads/s", "cores": threads, "kind": "reference", "pairs_per_s": n / best["seconds"],
                  "gcups": cells / best["seconds"] / 1e9,
                  "sample": "the step's %d pairs through the reference's own BSW kernels (bwa-mem2 v2.2.1 %s build) on %d threads, each with its own BandedPairWiseSW object "
                            "and slice: %.3f s wall for the kernel calls (best of 2; slowest thread %.3f s)" % (n, isa, threads, best["seconds"], best["slowest_thread_s"])}
        for fn in (fn_in, fn_out):
            try:
                os.remove(fn)
            except OSError:
                pass
    if par is None:                                              # no compiled reference on this box: the oracle's restatement on a sample, one thread
        n_s = min(n, 3000)
        oopt = oracle.default_opt()
        t = time.time(); bad = 0
        for i in range(n_s):
            exp1 = oracle.ksw_extend(qer[qoff[i]:qoff[i + 1]], ref[roff[i]:roff[i + 1]], oopt, w, end_bonus, int(h0[i]))
            if tuple(int(got[i][f]) for f in fields) != exp1:
                bad += 1
        cpu_s = time.time() - t
        par = {"pairs": n_s, "sample": "the first %d pairs against the oracle's ksw_extend2 restatement (oracle/_ref is not built here)" % n_s, "pairs_equal": bad == 0, "mismatches": bad}
        cb = {"value": n_s / cpu_s / TASKS_PER_READ if cpu_s > 0 else None, "unit": "reads/s", "cores": 1, "kind": "port",
              "sample": "%d pairs through oracle/bm2_oracle.c (ora_ksw_extend_cls) from Python, one thread, %.1f s" % (n_s, cpu_s)}
    bad = par["mismatches"]
    algo_bytes = float(int(roff[-1]) + int(qoff[-1]) + 2 * n * pairs.dtype.itemsize)     # every base once, every SeqPair read and written
    out = {
        "metric": "aligned reads/s (150bp PE vs GRCh38) at 1/2/4/8 GPU; SAM bit-exact vs ref",
        "value": world * n / TASKS_PER_READ * a.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
        "data": "synthetic" if not emu else "synthetic; HOST EMULATOR RUN (not a measurement)",
        "config": {"workload": "config 2 shape: the banded-SW kernel alone (S1: bm2_bsw_upload / bm2_bsw_run: the pairs sorted on the device, one per LANE where query and scores fit the lane kernel, one per wavefront otherwise; band %d) on %d "
                               "synthetic extension tasks per GPU per step shaped like those of 150 bp reads, batch resident in HBM; `value` = tasks/s "
                               "divided by the %.3f tasks per read the pe150 workload measures" % (w, n, TASKS_PER_READ),
                   "pairs_per_gpu_per_step": n, "tasks_per_read": TASKS_PER_READ, "parallelism": "one batch per GPU over %d GPU(s), no collectives" % world},
        "pairs_per_s": world * n * a.steps / dt,
        "roofline": {"kernel": "k_bsw_lanes (+ k_bsw_list)", "bound": "hbm", "achieved": algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else 0.0, "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": k_ms, "launches_per_step": 1,
                     "note": "integer DP: the kernel is bound by VALU / LDS issue, not by HBM (each base is read once); `extend_kernel.gcups` is its rate"},
        "extend_kernel": {"kernel": "k_bsw_lanes (+ k_bsw_list)", "gcups": cells / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "avg_launch_ms": k_ms, "cells_per_launch": cells},
        "parity": par,
        "cpu_baseline": cb,
    }
    if world == 1 and not a.no_binding_s1:
        try:
            f1, f2 = os.path.join(a.workdir, "cpu_1.fq"), os.path.join(a.workdir, "cpu_2.fq")
            meta = os.path.join(a.workdir, "genome_%dmbp_s%d.fa.contigs.npz" % (a.genome_mbp, seed))
            if not os.path.exists(f1) and os.path.exists(meta):   # (a run of its own: the reads the main line's CPU baseline would have left)
                from tools import synth
                z = np.load(meta, allow_pickle=True)
                c1, c2 = synth.make_reads_pe(seed + 5, [z["c%d" % i] for i in range(int(z["n"]))], a.cpu_pairs, L=150)
                synth.write_fastq(f1, c1, suffix="/1"); synth.write_fastq(f2, c2, suffix="/2")
                del z, c1, c2
            out["s1_binding"] = s1_binding_leg(a.workdir, os.path.join(a.workdir, "genome_%dmbp_s%d.fa" % (a.genome_mbp, seed)))
            if out["s1_binding"].get("sam_equal") is False:
                bad += 1
        except Exception as e:                                                        # noqa
            out["s1_binding"] = {"error": str(e)}
    from tools import bench_line
    bench_line.emit(out, getattr(a, "full_json", None))
    return 0 if bad == 0 else 3


def run_side_workloads(a, early_s1, time_left):
    """BASELINE configs 5 and 2 as workloads of their own (each a process of its own on this GPU, with its parity gate, its kernels' figures and the
    compiled reference timed beside it): {"config5": line, "config2": line}."""
    res = {}
    for key, name, extra, need_s in (("config5", "ont2d", ["--steps", 5, "--warmup", 1, "--distinct-chunks", 5, "--parity-reads", 2048, "--parity-regs-reads", 200] + CONFIG5_READS, 480),
                                     ("config2", "bsw", ["--steps", 5, "--warmup", 2], 90)):
        if time_left() < need_s:
            res[key] = {"skipped": "time budget (%.0f s left)" % time_left()}
            continue
        s1_done = key == "config2" and isinstance(early_s1, dict) and "sam_equal" in early_s1
        try:
            res[key] = side_workload(a, name, extra + (["--no-binding-s1"] if s1_done else []), min(need_s * 2, time_left() - 20))
            if s1_done:                                          # (timed at the start of the run, the GPU to itself)
                res[key]["s1_binding"] = early_s1
        except Exception as e:                                                        # noqa
            res[key] = {"error": str(e)}
    return res
