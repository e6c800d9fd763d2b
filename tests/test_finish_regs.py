"""The tail of mem_kernel2_core (mem_sort_dedup_patch + ALT flag, bwamem.cpp:1154-1169) against the reference's own REGFIN dumps.
Without a GPU: the CPU oracle's restatement (which the host-side SAM tests feed from) is pinned here; the device implementation
(finish.hip, bm2_finish_regs_dev / bm2_batch_finish) runs against the same dumps in tests/test_device_sources_on_host.py (emulator)
and, marked gpu, below."""
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ONT2D, alnregs_to_recs, first_diff, load_golden, oracle_finish_regs, ref_binary
from tools import refio, synth
import helpers  # noqa: E402

CASES = [("g60k", {}), ("g20k_l76", {}), ("g40k_ont", ONT2D)]


def _prg_to_regs(prg, n_reads):
    regs = np.zeros(len(prg), bm2.REG_DT)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        regs[f] = prg[f]
    reg_off = np.zeros(n_reads + 1, np.int64)
    np.add.at(reg_off, prg["read"] + 1, 1)
    return regs, np.cumsum(reg_off)


def split_hit_case(tmp_path):
    """long noisy reads with a junk stretch in the middle: the extension z-drops on both sides of it and mem_patch_reg re-joins the
    two hits through a global alignment -> (fa, enc, off, ln, refdump sections)"""
    exe = ref_binary("refdump")
    if exe is None or ref_binary() is None:
        helpers.no_checker("oracle/_ref not built")
    names, ctg, alts = synth.make_genome(91, [160000, 70000], alt_contigs=1, alt_len=4000, n_repeat_families=3,
                                         repeat_len=(300, 2000), copies=(3, 8), divergence=(0.0, 0.05))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_long(92, ctg, 50, mean_len=3000, max_len=8000, err=0.08)
    rng = np.random.default_rng(5)
    for r in reads[::2]:
        if len(r) > 1500:
            p = int(rng.integers(600, len(r) - 900))
            r[p:p + 300] = rng.integers(0, 4, size=300, dtype=np.uint8)
    rt = str(tmp_path / "reads.txt")
    with open(rt, "w") as f:
        for r in reads:
            f.write("".join("ACGTN"[c] for c in r) + "\n")
    dump = str(tmp_path / "d.bin")
    subprocess.check_call([exe, "-x", "ont2d", fa, rt, dump], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = refio.read_dump(dump)
    assert int((d["REGFIN"]["n_comp"] > 1).sum()) > 0, "the fixture should exercise mem_patch_reg"
    enc, off, ln = refio.pack_reads(reads)
    return fa, enc, off, ln, d


def _check(finish, pre, enc, off, ln, d, kw):
    regs, reg_off = _prg_to_regs(d["REGPRG"], len(ln))
    out, out_off = finish(pre, enc, off, ln, bm2.default_opt(**kw), regs, reg_off)
    got, exp = alnregs_to_recs(out, out_off), d["REGFIN"]
    assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), first_diff(exp, got)


@pytest.mark.parametrize("name,kw", CASES)
def test_oracle_finish_matches_reference_dump(golden_dir, name, kw):
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    _check(oracle_finish_regs, pre, enc, off, ln, d, kw)


def test_oracle_finish_merges_split_hits_like_the_reference(tmp_path):
    fa, enc, off, ln, d = split_hit_case(tmp_path)
    _check(oracle_finish_regs, fa, enc, off, ln, d, ONT2D)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", CASES)
def test_device_finish_matches_reference_dump(gpu_ctx_factory, golden_dir, name, kw):
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    ctx = gpu_ctx_factory(pre)
    _check(lambda p, e, o, l, opt, regs, ro: ctx.finish_regs((e, o, l), opt, regs, ro), pre, enc, off, ln, d, kw)


@pytest.mark.gpu
def test_device_finish_merges_split_hits_like_the_reference(gpu_ctx_factory, tmp_path):
    fa, enc, off, ln, d = split_hit_case(tmp_path)
    ctx = gpu_ctx_factory(fa)
    _check(lambda p, e, o, l, opt, regs, ro: ctx.finish_regs((e, o, l), opt, regs, ro), fa, enc, off, ln, d, ONT2D)
    # and the resident path: device regs -> bm2_batch_finish, against the oracle end to end
    opt = bm2.default_opt(**ONT2D)
    ctx.batch_upload(enc, off, ln); ctx.batch_run(opt); ctx.batch_finish(opt)
    aln, aln_off = ctx.batch_download_alnregs()
    got, exp = alnregs_to_recs(aln, aln_off), d["REGFIN"]
    assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), first_diff(exp, got)
