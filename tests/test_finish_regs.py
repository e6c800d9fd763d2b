"""Host tail of mem_kernel2_core (bm2_finish_regs = mem_sort_dedup_patch + ALT flag, bwamem.cpp:1154-1169) against the
reference's own REGFIN dumps.  Pure host code through the C ABI: no GPU needed."""
import os
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ONT2D, first_diff, load_golden, ref_binary
from tools import refio, synth


def _prg_to_regs(prg, n_reads):
    regs = np.zeros(len(prg), bm2.REG_DT)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        regs[f] = prg[f]
    reg_off = np.zeros(n_reads + 1, np.int64)
    np.add.at(reg_off, prg["read"] + 1, 1)
    return regs, np.cumsum(reg_off)


def _to_records(out, out_off):
    rec = np.zeros(len(out), refio.REG_DT)
    rec["read"] = np.repeat(np.arange(len(out_off) - 1), np.diff(out_off))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep"):
        rec[f] = out[f]
    return rec


@pytest.mark.parametrize("name,kw", [("g60k", {}), ("g20k_l76", {}), ("g40k_ont", ONT2D)])
def test_finish_regs_matches_reference_dump(golden_dir, name, kw):
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    regs, reg_off = _prg_to_regs(d["REGPRG"], len(ln))
    out, out_off = bm2.finish_regs(pre, enc, off, ln, bm2.default_opt(**kw), regs, reg_off)
    got, exp = _to_records(out, out_off), d["REGFIN"]
    assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), first_diff(exp, got)


def test_finish_regs_merges_split_hits_like_the_reference(tmp_path):
    # long noisy reads: z-drop splits alignments that mem_patch_reg then re-joins through a global alignment
    exe = ref_binary("refdump")
    if exe is None or ref_binary() is None:
        pytest.skip("oracle/_ref not built")
    names, ctg, alts = synth.make_genome(91, [160000, 70000], alt_contigs=1, alt_len=4000, n_repeat_families=3,
                                         repeat_len=(300, 2000), copies=(3, 8), divergence=(0.0, 0.05))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_long(92, ctg, 50, mean_len=3000, max_len=8000, err=0.08)
    rng = np.random.default_rng(5)
    for r in reads[::2]:                                  # a junk stretch in the middle: the extension z-drops on both sides of it
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
    enc, off, ln = refio.pack_reads(reads)
    regs, reg_off = _prg_to_regs(d["REGPRG"], len(ln))
    out, out_off = bm2.finish_regs(fa, enc, off, ln, bm2.default_opt(**ONT2D), regs, reg_off)
    got, exp = _to_records(out, out_off), d["REGFIN"]
    assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), first_diff(exp, got)
    n_merged = int((exp["n_comp"] > 1).sum())
    print("merged hits:", n_merged, "of", len(exp), "(from", len(d["REGPRG"]), "before the tail)")
    assert n_merged > 0, "the fixture should exercise mem_patch_reg"

