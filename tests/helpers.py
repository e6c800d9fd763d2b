"""Shared test helpers: random extension-like task generators."""
import numpy as np


def random_pairs(seed, n, max_len=150, h0_max=150, n_frac=0.005, div=(0.02, 0.15), long_tail=True):
    """Extension-like (query, target, h0) triples: target = mutated copy of query + flank, like the pairs
    mem_chain2aln builds (bwamem.cpp:2229-2418)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ql = int(rng.integers(1, max_len + 1))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        d = rng.uniform(*div)
        t = []
        i = 0
        while i < ql:
            u = rng.random()
            if u < d * 0.7:
                t.append((q[i] + rng.integers(1, 4)) % 4); i += 1
            elif u < d * 0.85:
                i += int(rng.integers(1, 4))                       # deletion from target
            elif u < d:
                t.extend(rng.integers(0, 4, size=int(rng.integers(1, 4))).tolist())   # insertion in target
            else:
                t.append(q[i]); i += 1
        if rng.random() < 0.15:                                     # big gap
            p = int(rng.integers(0, len(t) + 1))
            t[p:p] = rng.integers(0, 4, size=int(rng.integers(5, 40))).tolist()
        if rng.random() < 0.2 and len(t) > 10:                      # divergence to trigger z-drop
            p = int(rng.integers(len(t) // 2, len(t)))
            t[p:] = rng.integers(0, 4, size=len(t) - p).tolist()
        if long_tail:
            t.extend(rng.integers(0, 4, size=int(rng.integers(0, max(2, ql)))).tolist())
        t = np.array(t if t else [0], dtype=np.uint8) % 4
        if rng.random() < n_frac * 20:
            q[int(rng.integers(0, ql))] = 4
        h0 = int(rng.integers(1, h0_max + 1))
        out.append((q, t, h0))
    return out


def pack_pairs(bm2, triples):
    """-> (pairs SEQPAIR_DT array, ref bytes, qer bytes) in the S1 layout (flat seqBuf arrays, bandedSWA.h:90-99)."""
    pairs = np.zeros(len(triples), bm2.SEQPAIR_DT)
    refs, qers = [], []
    ro = qo = 0
    for i, (q, t, h0) in enumerate(triples):
        pairs[i]["idr"], pairs[i]["idq"], pairs[i]["id"] = ro, qo, i
        pairs[i]["len1"], pairs[i]["len2"], pairs[i]["h0"] = len(t), len(q), h0
        refs.append(t); qers.append(q)
        ro += len(t); qo += len(q)
    return pairs, np.concatenate(refs), np.concatenate(qers)


# ---------------------------------------------------------------------------------------------------------------
import os
import subprocess


def gpu_box():
    """a HIP device is visible to this process (the GPU box; BM2_EMU_LIB runs of the gpu tests on the host emulator count as one)"""
    if os.environ.get("BM2_EMU_LIB"):
        return True
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:                                             # noqa
        return os.path.exists("/dev/kfd")


def no_checker(reason):
    """The compiled reference (oracle/_ref) is not there.  On a box WITHOUT a GPU the test has nothing to say and is skipped; on the GPU box a missing
    checker FAILS the test -- a parity suite that goes quiet when its checker did not travel is not green (hygiene)."""
    import pytest
    if gpu_box():
        pytest.fail("checker missing on a GPU box: " + reason)
    pytest.skip(reason)


def load_golden(golden_dir, name):
    """-> (index_prefix, enc, off, len, dump dict)"""
    from tools import refio
    pre = os.path.join(golden_dir, name + ".fa")
    _, seqs = refio.read_fastq(os.path.join(golden_dir, name + ".reads.txt"))
    enc, off, ln = refio.pack_reads(seqs)
    d = dict(np.load(os.path.join(golden_dir, name + ".dump.npz")))
    return pre, enc, off, ln, d


def ref_binary(kind="bwa-mem2"):
    """Path of the prebuilt reference binary matching this host's ISA (oracle/_ref travels to the GPU box), or None."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = open("/proc/cpuinfo").read()
    for a in (["avx512bw"] if "avx512bw" in flags else []) + (["avx2"] if "avx2" in flags else []) + ["sse41"]:
        p = os.path.join(root, "oracle", "_ref", "%s.%s" % (kind, a))
        if os.path.exists(p):
            return p
    return None


def build_index(fa_path):
    exe = ref_binary()
    if exe is None:
        return False
    subprocess.check_call([exe, "index", fa_path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return True


def chain_mask(a):
    """CHN0 records carry w/kept/first that mem_chain_seeds never initialises; compare only the defined fields."""
    b = a.copy()
    b["w"] = 0; b["kept"] = 0; b["first"] = 0
    return b


def gpu_stage_records(ctx, bm2, n_reads):
    """Re-assemble the device arrays of the last batch_run into the record layout of refdump (CHN1/SEED1/REGRAW)."""
    from tools import refio
    base = ctx.batch_fetch("read_base", "<i8")
    n_chain = ctx.batch_fetch("n_chain", "<i4")
    n_reg = ctx.batch_fetch("n_reg", "<i4")
    chn = ctx.batch_fetch("chn", bm2.DEVCHAIN_DT)
    seeds = ctx.batch_fetch("seeds", bm2.DEVSEED_DT)
    regs = ctx.batch_fetch("regs_raw", bm2.DEVREG_DT)
    C, S, R = [], [], []
    for r in range(n_reads):
        b = int(base[r])
        for j in range(int(n_chain[r])):
            c = chn[b + j]
            C.append((r, c["n"], c["rid"], c["is_alt"], c["pos"], c["frac_rep"], c["w"], c["kept"], c["first"]))
            for t in range(int(c["n"])):
                s = seeds[int(c["seed_off"]) + t]
                S.append((s["rbeg"], s["qbeg"], s["len"], s["score"], 0))
        for i in range(int(n_reg[r])):
            a = regs[b + i]
            R.append((r, 0, a["rb"], a["re"], a["qb"], a["qe"], a["rid"], a["score"], a["truesc"], 0, 0, 0, 0, a["w"],
                      a["seedcov"], 0, 0, a["seedlen0"], 0, 0, a["frac_rep"], 0))
    return (np.array(C, dtype=refio.CHAIN_DT), np.array(S, dtype=refio.SEED_DT), np.array(R, dtype=refio.REG_DT))


def regs_to_records(regs, reg_off):
    from tools import refio
    out = np.zeros(len(regs), refio.REG_DT)
    reads = np.repeat(np.arange(len(reg_off) - 1), np.diff(reg_off))
    out["read"] = reads
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        out[f] = regs[f]
    return out


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return "first difference at %d:\n  exp %s\n  got %s" % (i, a[i], b[i])
    return "lengths differ: %d vs %d" % (len(a), len(b))


ONT2D = dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1, pen_clip5=0, pen_clip3=0, min_seed_len=14, min_chain_weight=20,
             split_factor=10.0)      # `-x ont2d`, fastmap.cpp:812-826


def oracle_regs_fn(prefix):
    """Stand-in for the device stage, built on the CPU oracle: f(enc, off, ln) -> (alnregs, aln_off) as bm2_batch_finish +
    bm2_batch_download_alnregs return them.  For checking host-side code (the SAM tail, tools/bm2_mem.py) without a GPU."""
    from tools import oracle
    ix = oracle.Index(prefix)

    def f(enc, off, ln):
        return recs_to_alnregs(ix.run(enc, off, ln)["REGFIN"], len(ln))
    return f


def _oracle_opt(opt):
    """bm2.Opt -> oracle.OraOpt (the two structs list the same fields in the same order)."""
    import ctypes as C
    from tools import oracle
    assert C.sizeof(opt) == C.sizeof(oracle.OraOpt)
    return oracle.OraOpt.from_buffer_copy(bytes(opt))


def recs_to_alnregs(rec, n_reads):
    """refdump / oracle REG_DT records (REGFIN) -> (alnregs ALNREG_DT, aln_off) as the library hands them to the SAM tail."""
    import bm2
    aln = np.zeros(len(rec), bm2.ALNREG_DT)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep"):
        aln[f] = rec[f]
    off = np.zeros(n_reads + 1, np.int64)
    np.add.at(off, rec["read"] + 1, 1)
    return aln, np.cumsum(off)


def alnregs_to_recs(aln, aln_off):
    from tools import refio
    rec = np.zeros(len(aln), refio.REG_DT)
    rec["read"] = np.repeat(np.arange(len(aln_off) - 1), np.diff(aln_off))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep"):
        rec[f] = aln[f]
    return rec


def oracle_finish_regs(prefix, enc, off, ln, opt, regs, reg_off):
    """The tail of mem_kernel2_core (mem_sort_dedup_patch + ALT flag) by the CPU oracle, with the arguments and results of
    bm2.Context.finish_regs: lets the host-side SAM tail be tested without a GPU (the device stage itself is tested against the
    same REGFIN dumps in the emulator and GPU tests)."""
    from tools import oracle, refio
    rec = np.zeros(len(regs), refio.REG_DT)
    rec["read"] = np.repeat(np.arange(len(ln)), np.diff(reg_off))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        rec[f] = regs[f]
    ix = oracle.Index(prefix)
    try:
        fin = ix.finish_regs(enc, off, ln, rec, _oracle_opt(opt))
    finally:
        ix.close()
    return recs_to_alnregs(fin, len(ln))
